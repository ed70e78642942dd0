// K7: per-trajectory MLP programs -- the N-row (not N*M-row) networks around the filters:
// control / position / force-torque encoders, the hoisted halves of the join layers, the
// particle-filter and Kalman-filter weight models, the virtual sensors' trunk and heads.
//
// The reference evaluates them as dozens of tiny nn.Linear / ReLU / add launches per step
//   /root/reference/crossmodal/door_models/layers.py:11-40,66-95      (vector encoders)
//   /root/reference/crossmodal/door_models/crossmodal_pf.py:74-106    (PF weight model)
//   /root/reference/crossmodal/door_models/kf.py:81-126               (virtual sensor)
//   /root/reference/crossmodal/door_models/crossmodal_kf.py:134-167   (EKF weight model)
// Here a whole model is ONE launch: a short instruction list (LOAD / LINEAR / STORE) is
// interpreted by each wave for 8 rows at a time.  Vectors (<= 128 wide) live in per-wave LDS
// slots; a LINEAR streams the transposed weight rows from L2 once per 8 rows (one coalesced
// 256-B load per k feeds 8 FMAs per lane) and reads the inputs as LDS broadcasts.  Rows are
// few (N or T*N) and the work is ~50 kMAC per row: latency-, not throughput-critical -- the
// point is to keep library heuristics and ~50 launches per step off the hot path.
#include "mmf_common.h"

namespace {

constexpr int kRows = 8;     // rows per wave
constexpr int kMaxSlots = MMF_TRAJ_SLOTS;
constexpr int kMaxVec = 128; // max vector width
constexpr int kWaves = 4;    // waves per workgroup: at most 4 * 8 slots * 8 rows * 128 * 4 B = 128 KiB LDS.
// The launch sizes the slot file for what the program uses (n_slots, vector width 64 or 128): a
// 4-slot, 64-wide program takes 32 KiB per workgroup, so four workgroups (16 waves) share a CU and
// hide each other's L2 / LDS latency -- the kernel is a chain of dependent loads, not a FLOP problem.

struct IoPtrs {
  float* p[MMF_TRAJ_MAX_IO];
};

__device__ __forceinline__ float activate(float v, int act, float fparam) {
  switch (act) {
    case MMF_TRAJ_ACT_RELU: return fmaxf(v, 0.f);
    case MMF_TRAJ_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    case MMF_TRAJ_ACT_SQRT_SQ_PLUS: return sqrtf(__fmaf_rn(v, v, fparam));
    default: return v;
  }
}

__global__ __launch_bounds__(kWaves * MMF_WAVE) void traj_program_kernel(
    const MmfTrajInstr* __restrict__ prog, int n_instr, const float* __restrict__ weights, IoPtrs io, int R,
    int n_slots, int kVec) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* slots = lds + wave * (n_slots * kRows * kVec);  // [slot][row][kVec]
  const int wave_global = blockIdx.x * kWaves + wave, waves_total = gridDim.x * kWaves;

  for (int task = wave_global; task * kRows < R; task += waves_total) {
    const int row0 = task * kRows;
    const int nrows = min(kRows, R - row0);
    for (int ip = 0; ip < n_instr; ++ip) {
      const MmfTrajInstr I = prog[ip];
      if (I.op == MMF_TRAJ_LOAD) {
        const float* src = io.p[I.io];
        float* dst = slots + I.dst * (kRows * kVec);
#pragma unroll
        for (int r = 0; r < kRows; ++r) {
          const size_t g = static_cast<size_t>(row0 + min(r, nrows - 1)) * I.io_stride + I.io_off;
          if (lane < I.out_dim) dst[r * kVec + lane] = src[g + lane];
          if (lane + 64 < I.out_dim) dst[r * kVec + 64 + lane] = src[g + 64 + lane];
        }
      } else if (I.op == MMF_TRAJ_LINEAR) {
        const bool wide = I.out_dim > 64;
        const int out_pad = wide ? 128 : 64;
        float acc0[kRows], acc1[kRows];
        const float b0 = (I.b_off >= 0 && lane < I.out_dim) ? weights[I.b_off + lane] : 0.f;
        const float b1 = (I.b_off >= 0 && wide && lane + 64 < I.out_dim) ? weights[I.b_off + 64 + lane] : 0.f;
#pragma unroll
        for (int r = 0; r < kRows; ++r) { acc0[r] = b0; acc1[r] = b1; }
        const float* wT = weights + I.w_off + lane;
#pragma unroll 1
        for (int s = 0; s < 4; ++s) {
          if (I.src[s] < 0) break;
          const float* xs = slots + I.src[s] * (kRows * kVec) + I.src_off[s];
          const int dim = I.src_dim[s];
          int k = 0;
          // 16 k per step: all 16 (or 32) coalesced weight-row loads are issued before the first
          // FMA, so a 64-wide source costs 4 L2 round trips instead of 16 (rows are few: the
          // program is latency-, not throughput-bound)
          for (; k + 16 <= dim; k += 16) {
            float w0[16], w1[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              w0[e] = wT[(k + e) * out_pad];
              w1[e] = wide ? wT[(k + e) * out_pad + 64] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
              for (int r = 0; r < kRows; ++r) {
                const float4 x = *reinterpret_cast<const float4*>(xs + r * kVec + k + 4 * q);
                acc0[r] = __fmaf_rn(w0[4 * q + 0], x.x, acc0[r]); acc0[r] = __fmaf_rn(w0[4 * q + 1], x.y, acc0[r]);
                acc0[r] = __fmaf_rn(w0[4 * q + 2], x.z, acc0[r]); acc0[r] = __fmaf_rn(w0[4 * q + 3], x.w, acc0[r]);
                if (wide) {
                  acc1[r] = __fmaf_rn(w1[4 * q + 0], x.x, acc1[r]); acc1[r] = __fmaf_rn(w1[4 * q + 1], x.y, acc1[r]);
                  acc1[r] = __fmaf_rn(w1[4 * q + 2], x.z, acc1[r]); acc1[r] = __fmaf_rn(w1[4 * q + 3], x.w, acc1[r]);
                }
              }
          }
          for (; k + 4 <= dim; k += 4) {  // 4 k per step: b128 broadcast reads of the inputs
            float w0[4], w1[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              w0[e] = wT[(k + e) * out_pad];
              w1[e] = wide ? wT[(k + e) * out_pad + 64] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < kRows; ++r) {
              const float4 x = *reinterpret_cast<const float4*>(xs + r * kVec + k);
              // explicit fma: every row must round identically whatever its position in the
              // batch (left to the contraction pass, rows 6-7 of a wave came out differently)
              acc0[r] = __fmaf_rn(w0[0], x.x, acc0[r]); acc0[r] = __fmaf_rn(w0[1], x.y, acc0[r]);
              acc0[r] = __fmaf_rn(w0[2], x.z, acc0[r]); acc0[r] = __fmaf_rn(w0[3], x.w, acc0[r]);
              if (wide) {
                acc1[r] = __fmaf_rn(w1[0], x.x, acc1[r]); acc1[r] = __fmaf_rn(w1[1], x.y, acc1[r]);
                acc1[r] = __fmaf_rn(w1[2], x.z, acc1[r]); acc1[r] = __fmaf_rn(w1[3], x.w, acc1[r]);
              }
            }
          }
          for (; k < dim; ++k) {
            const float w0 = wT[k * out_pad];
            const float w1 = wide ? wT[k * out_pad + 64] : 0.f;
#pragma unroll
            for (int r = 0; r < kRows; ++r) {
              const float x = xs[r * kVec + k];
              acc0[r] = __fmaf_rn(w0, x, acc0[r]);
              if (wide) acc1[r] = __fmaf_rn(w1, x, acc1[r]);
            }
          }
          wT += dim * out_pad;
        }
        const float* res = I.res >= 0 ? slots + I.res * (kRows * kVec) : nullptr;
        float* dst = slots + I.dst * (kRows * kVec);
        // every source (and the residual) is read before anything is written: dst may alias them
        float v0[kRows], v1[kRows];
#pragma unroll
        for (int r = 0; r < kRows; ++r) {
          v0[r] = activate(__fadd_rn(acc0[r], res ? res[r * kVec + lane] : 0.f), I.act, I.fparam);
          v1[r] = wide ? activate(__fadd_rn(acc1[r], res ? res[r * kVec + 64 + lane] : 0.f), I.act, I.fparam) : 0.f;
        }
#pragma unroll
        for (int r = 0; r < kRows; ++r) {
          dst[r * kVec + lane] = v0[r];
          if (wide) dst[r * kVec + 64 + lane] = v1[r];
        }
      } else {  // STORE / STORE_DIAG
        float* out = io.p[I.io];
        const float* src = slots + I.src[0] * (kRows * kVec) + I.src_off[0];
        for (int r = 0; r < nrows; ++r) {
          const size_t g = static_cast<size_t>(row0 + r) * I.io_stride + I.io_off;
          if (I.op == MMF_TRAJ_STORE) {
            if (lane < I.out_dim) out[g + lane] = activate(src[r * kVec + lane], I.act, I.fparam);
            if (lane + 64 < I.out_dim) out[g + 64 + lane] = activate(src[r * kVec + 64 + lane], I.act, I.fparam);
          } else {  // (d x d) matrix with the vector on its diagonal; out_dim = d
            const int d = I.out_dim;
            if (lane < d * d) {
              const int i = lane / d, j = lane % d;
              out[g + lane] = (i == j) ? activate(src[r * kVec + i], I.act, I.fparam) : 0.f;
            }
          }
        }
      }
    }
  }
}

}  // namespace

extern "C" int mmf_traj_program(const MmfTrajInstr* prog, int n_instr, const float* weights,
                                float* const* io, int R, int n_slots, int vec_width, void* stream) {
  if (!prog || !weights || !io || n_instr < 1 || R < 0) return MMF_EINVAL;
  if (n_slots < 1 || n_slots > kMaxSlots || (vec_width != 64 && vec_width != kMaxVec)) return MMF_EINVAL;
  if (R == 0) return 0;
  IoPtrs p{};
  for (int i = 0; i < MMF_TRAJ_MAX_IO; ++i) p.p[i] = io[i];
  const size_t lds = static_cast<size_t>(kWaves) * n_slots * kRows * vec_width * sizeof(float);
  static_assert(static_cast<size_t>(kWaves) * kMaxSlots * kRows * kMaxVec * sizeof(float) <= 160 * 1024, "slots must fit LDS");
  auto k = traj_program_kernel;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
  if (e != hipSuccess) return static_cast<int>(e);
  const int tasks = (R + kRows - 1) / kRows;
  int grid = (tasks + kWaves - 1) / kWaves;
  const int per_cu = static_cast<int>((160 * 1024) / lds) < 8 ? static_cast<int>((160 * 1024) / lds) : 8;
  if (grid > 256 * per_cu) grid = 256 * per_cu;
  k<<<grid, kWaves * MMF_WAVE, lds, static_cast<hipStream_t>(stream)>>>(prog, n_instr, weights, p, R, n_slots, vec_width);
  MMF_CHECK_LAUNCH();
  return 0;
}
