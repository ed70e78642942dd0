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
// interpreted by each workgroup for 16 rows at a time.  Vectors (<= 128 wide) live in LDS
// slots; a LINEAR runs on v_mfma_f32_16x16x4_f32 (outputs x rows tiles, exact fp32, the same fma
// chains as a scalar evaluation) with its weights streamed from L2 as pre-arranged fragments.
// Rows are few (N or T*N) and the work is ~50 kMAC per row: the point is to keep library
// heuristics and ~50 launches per step off the hot path, and the step's latency short.
#include "mmf_common.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kRows = 16;    // rows per task: the N dimension of v_mfma_f32_16x16x4_f32
constexpr int kMaxSlots = MMF_TRAJ_SLOTS;
constexpr int kMaxVec = 128; // max vector width
constexpr int kPad = 4;      // LDS row stride = width + 4 floats: the 16 rows x 4 k of a B fragment hit 64 banks
constexpr int kWaves = 4;    // waves per workgroup = per task: one 16-output tile of a 64-wide layer each
// The launch sizes the slot file for what the program uses (n_slots, vector width 64 or 128): a 5-slot,
// 64-wide program takes 21 KiB per workgroup, so seven workgroups (28 waves) share a CU.

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

// LINEAR on the f32 matrix cores.  D (16 outputs x 16 rows) += A (16 outputs x 4 k) B (4 k x 16 rows), one
// v_mfma_f32_16x16x4_f32 per k-step and 16-output tile: lane (i = lane & 15, q = lane >> 4) supplies
// A = W[16 mt + i][4 s + q] and B = x[row i][4 s + q]; its accumulator registers r = 0..3 are outputs
// 16 mt + 4 q + r of row i.  The instruction is bit-for-bit a k-ordered fma chain (q = 0..3 inside a step,
// steps ascending), and the accumulators start at the bias: every row's outputs are the same fma chains in
// the same order as the VALU formulation this replaces (and as oracle/strict restates them), whatever the
// row's position in the batch.  Weights arrive pre-arranged per (source, group of 4 steps, tile) as one
// 16-byte fragment per lane (trajprog.py / mmf_traj_pack).
// Round 5: the four waves of a workgroup share ONE task of 16 rows -- wave w owns output tiles w (and w + 4 of a
// 128-wide layer) -- so a 64 x 64 layer is 16 MFMAs per wave behind ONE round of fragment loads (all groups of a
// source requested before the first MFMA) instead of 64 behind four dependent rounds: a program is a chain of L2
// round trips, and at the T*N = 512 rows of a training step there is one task per CU and nothing else to hide them.
// Which wave computes a tile does not change its chain: results are bit-identical to the one-wave form.
// What the NEXT instruction needs from memory, requested while the current one runs (an instruction's addresses depend on
// the program and the task only, never on data): without it every instruction starts with its own L2 / HBM round trip.
struct Pre {
  f32x4 a[4][2];   // LINEAR: the fragments of its first chunk (source 0, groups 0 .. 3) for this wave's tiles
  f32x4 bias[2];   // LINEAR: its bias (zeros without one)
  float io[4][2];  // LOAD / LOAD_ADD / MASK: rows wave, wave + 4, .. of the task, columns lane and lane + 64
};

__device__ __forceinline__ void prefetch(const MmfTrajInstr& I, const float* __restrict__ weights, const IoPtrs& io, int row0,
                                         int nrows, int lane, int wave, Pre& p) {
  if (I.op == MMF_TRAJ_LINEAR) {
    const int q = lane >> 4;
    const int tpw = I.out_dim > 64 ? 2 : 1, TT = 4 * tpw;
    const int groups = (I.src_dim[0] + 15) >> 4;
    const float* w = weights + I.w_off + lane * 4;
#pragma unroll
    for (int t = 0; t < 2; ++t)
      if (t < tpw) {
        const int mt = wave + 4 * t;
        p.bias[t] = I.b_off >= 0 ? *reinterpret_cast<const f32x4*>(weights + I.b_off + 16 * mt + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) {
          const int g = min(gg, groups - 1);  // a clamped duplicate is never multiplied
          p.a[gg][t] = *reinterpret_cast<const f32x4*>(w + (g * TT + mt) * 256);
        }
      }
  } else if (I.op == MMF_TRAJ_LOAD || I.op == MMF_TRAJ_LOAD_ADD || I.op == MMF_TRAJ_MASK) {
    const float* src = io.p[I.io];
#pragma unroll
    for (int jr = 0; jr < 4; ++jr) {
      const size_t g = static_cast<size_t>(row0 + min(wave + kWaves * jr, nrows - 1)) * I.io_stride + I.io_off;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int c = lane + 64 * h;
        p.io[jr][h] = c < I.out_dim ? src[g + c] : 0.f;
      }
    }
  }
}

template <int TPW>  // tiles per wave: 1 (out_dim <= 64) or 2
__device__ __forceinline__ void linear_mfma(const MmfTrajInstr& I, const float* __restrict__ weights, float* slots, int ld,
                                            int lane, int wave, const Pre& pre) {
  constexpr int TT = 4 * TPW;  // tiles of the layer
  const int i = lane & 15, q = lane >> 4;
  f32x4 acc[TPW], cur[4][TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    acc[t] = pre.bias[t];
#pragma unroll
    for (int gg = 0; gg < 4; ++gg) cur[gg][t] = pre.a[gg][t];
  }
  // chunks of four groups (64 inputs) over the sources in order; the next chunk's fragments are requested before this
  // chunk's MFMAs (the first came with `pre`)
  const float* w = weights + I.w_off + lane * 4;  // fragments of the current source
  // (selects, not run-time indices: the instruction stays in registers)
  auto sel = [](const int32_t (&v)[4], int k) { return k == 0 ? v[0] : (k == 1 ? v[1] : (k == 2 ? v[2] : v[3])); };
  int s = 0, g0 = 0;
  int dim = I.src_dim[0], groups = (dim + 15) >> 4;
#pragma unroll 1
  for (;;) {
    int ns = s, ng0 = g0 + 4;
    const float* nw = w;
    int ndim = dim, ngroups = groups;
    if (ng0 >= groups) {
      ns = s + 1;
      ng0 = 0;
      nw = w + groups * TT * 256;
      const bool more = ns < 4 && sel(I.src, ns) >= 0;
      if (more) { ndim = sel(I.src_dim, ns); ngroups = (ndim + 15) >> 4; }
      else ns = -1;
    }
    f32x4 nxt[4][TPW];
    if (ns >= 0) {
#pragma unroll
      for (int gg = 0; gg < 4; ++gg)
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
          const int g = min(ng0 + gg, ngroups - 1);
          nxt[gg][t] = *reinterpret_cast<const f32x4*>(nw + (g * TT + wave + 4 * t) * 256);
        }
    }
    const float* xs = slots + sel(I.src, s) * (kRows * ld) + sel(I.src_off, s) + i * ld + q;
    float b[4][4];
#pragma unroll
    for (int gg = 0; gg < 4; ++gg)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int k = 16 * (g0 + gg) + 4 * ks + q;
        b[gg][ks] = k < dim ? xs[16 * (g0 + gg) + 4 * ks] : 0.f;   // padded k: zero weight times a DEFINED zero
      }
#pragma unroll
    for (int gg = 0; gg < 4; ++gg)
      if (g0 + gg < groups) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int t = 0; t < TPW; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[gg][t][ks], b[gg][ks], acc[t], 0, 0, 0);
      }
    if (ns < 0) break;
#pragma unroll
    for (int gg = 0; gg < 4; ++gg)
#pragma unroll
      for (int t = 0; t < TPW; ++t) cur[gg][t] = nxt[gg][t];
    s = ns; g0 = ng0; w = nw; dim = ndim; groups = ngroups;
  }
  // epilogue: (+ residual) activation, written back as 16-byte pieces.  dst may alias a source: every wave has read
  // its sources before any wave writes
  __syncthreads();
  const float* res = I.res >= 0 ? slots + I.res * (kRows * ld) + i * ld + I.dst_off : nullptr;
  float* dst = slots + I.dst * (kRows * ld) + i * ld + I.dst_off;
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int mt = wave + 4 * t;
    f32x4 r = {0.f, 0.f, 0.f, 0.f};
    if (res) r = *reinterpret_cast<const f32x4*>(res + 16 * mt + 4 * q);
    f32x4 out;
#pragma unroll
    for (int e = 0; e < 4; ++e) out[e] = activate(__fadd_rn(acc[t][e], r[e]), I.act, I.fparam);
    *reinterpret_cast<f32x4*>(dst + 16 * mt + 4 * q) = out;
  }
}

__global__ __launch_bounds__(kWaves * MMF_WAVE) void traj_program_kernel(
    const MmfTrajInstr* __restrict__ prog, int n_instr, const float* __restrict__ weights, IoPtrs io, int R,
    int n_slots, int kVec) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ld = kVec + kPad;
  float* slots = lds;  // [slot][row][ld]: one task of 16 rows per workgroup

  for (int task = blockIdx.x; task * kRows < R; task += gridDim.x) {
    const int row0 = task * kRows;
    const int nrows = min(kRows, R - row0);
    MmfTrajInstr I = prog[0];
    Pre pre;
    prefetch(I, weights, io, row0, nrows, lane, wave, pre);
    for (int ip = 0; ip < n_instr; ++ip) {
      const MmfTrajInstr In = prog[ip + 1 < n_instr ? ip + 1 : ip];
      Pre nxt;
      if (ip + 1 < n_instr) prefetch(In, weights, io, row0, nrows, lane, wave, nxt);
      __builtin_amdgcn_sched_barrier(0);  // the requests stay ahead of this instruction's work
      if (I.op == MMF_TRAJ_LOAD || I.op == MMF_TRAJ_LOAD_ADD || I.op == MMF_TRAJ_MASK) {
        // LOAD: slot = act(io); LOAD_ADD: slot += io; MASK: slot = io > 0 ? slot : 0 (the backward of a ReLU, from the
        // stashed output).  Rows past the end repeat the last row: defined values that no STORE writes back.
        float* dst = slots + I.dst * (kRows * ld) + I.dst_off;
#pragma unroll
        for (int jr = 0; jr < 4; ++jr) {
          const int r = wave + kWaves * jr;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int c = lane + 64 * h;
            if (c < I.out_dim) {
              const float v = pre.io[jr][h];
              float* d = dst + r * ld + c;
              if (I.op == MMF_TRAJ_LOAD) *d = activate(v, I.act, I.fparam);
              else if (I.op == MMF_TRAJ_LOAD_ADD) *d = __fadd_rn(*d, v);
              else *d = v > 0.f ? *d : 0.f;
            }
          }
        }
      } else if (I.op == MMF_TRAJ_ADD || I.op == MMF_TRAJ_ZERO) {
        float* dst = slots + I.dst * (kRows * ld) + I.dst_off;
        const float* src = I.op == MMF_TRAJ_ADD ? slots + I.src[0] * (kRows * ld) + I.src_off[0] : nullptr;
#pragma unroll
        for (int r = wave; r < kRows; r += kWaves)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int c = lane + 64 * h;
            if (c < I.out_dim) dst[r * ld + c] = src ? __fadd_rn(dst[r * ld + c], src[r * ld + c]) : 0.f;
          }
      } else if (I.op == MMF_TRAJ_LINEAR) {
        if (I.out_dim > 64) linear_mfma<2>(I, weights, slots, ld, lane, wave, pre);
        else linear_mfma<1>(I, weights, slots, ld, lane, wave, pre);
      } else {  // STORE / STORE_DIAG
        float* out = io.p[I.io];
        const float* src = slots + I.src[0] * (kRows * ld) + I.src_off[0];
        for (int r = wave; r < nrows; r += kWaves) {
          const size_t g = static_cast<size_t>(row0 + r) * I.io_stride + I.io_off;
          if (I.op == MMF_TRAJ_STORE) {
            if (lane < I.out_dim) out[g + lane] = activate(src[r * ld + lane], I.act, I.fparam);
            if (lane + 64 < I.out_dim) out[g + 64 + lane] = activate(src[r * ld + 64 + lane], I.act, I.fparam);
          } else {  // (d x d) matrix with the vector on its diagonal; out_dim = d
            const int d = I.out_dim;
            if (lane < d * d) {
              const int i = lane / d, j = lane % d;
              out[g + lane] = (i == j) ? activate(src[r * ld + i], I.act, I.fparam) : 0.f;
            }
          }
        }
      }
      __syncthreads();  // the next instruction reads what any wave of this one wrote (and the next task reuses the slots)
      I = In;
      pre = nxt;
    }
  }
}

// Weight blob of a program from the nn.Module parameters where they lie: one workgroup per part.
__global__ __launch_bounds__(256) void traj_pack_kernel(const MmfTrajPackDesc* __restrict__ desc, float* __restrict__ blob) {
  const MmfTrajPackDesc d = desc[blockIdx.x];
  const float* src = reinterpret_cast<const float*>(d.src);
  float* dst = blob + d.dst_off;
  if (d.kind == MMF_TRAJ_PACK_BIAS) {
    for (int j = threadIdx.x; j < 128; j += blockDim.x) dst[j] = j < d.rows ? src[j] : 0.f;
    return;
  }
  const int MT = d.out_pad >> 4, groups = (d.dim + 15) >> 4;
  const int n = groups * MT * 256;
  for (int e = threadIdx.x; e < n; e += blockDim.x) {
    const int ks = e & 3, i = (e >> 2) & 15, q = (e >> 6) & 3, mt = (e >> 8) % MT, g = (e >> 8) / MT;
    const int o = 16 * mt + i, k = 16 * g + 4 * ks + q;
    float v = 0.f;
    if (k < d.dim && o < d.rows)
      v = d.kind == MMF_TRAJ_PACK_LAYER ? src[static_cast<size_t>(o) * d.ld + d.col0 + k]
                                        : src[static_cast<size_t>(k) * d.ld + d.col0 + o];  // transposed block
    dst[e] = v;
  }
}

}  // namespace

extern "C" int mmf_traj_pack(const MmfTrajPackDesc* desc, int n_desc, float* blob, void* stream) {
  if (!desc || !blob || n_desc < 1) return MMF_EINVAL;
  traj_pack_kernel<<<n_desc, 256, 0, static_cast<hipStream_t>(stream)>>>(desc, blob);
  MMF_CHECK_LAUNCH();
  return 0;
}

extern "C" int mmf_traj_program(const MmfTrajInstr* prog, int n_instr, const float* weights,
                                float* const* io, int R, int n_slots, int vec_width, void* stream) {
  if (!prog || !weights || !io || n_instr < 1 || R < 0) return MMF_EINVAL;
  if (n_slots < 1 || n_slots > kMaxSlots || (vec_width != 64 && vec_width != kMaxVec)) return MMF_EINVAL;
  if (R == 0) return 0;
  IoPtrs p{};
  for (int i = 0; i < MMF_TRAJ_MAX_IO; ++i) p.p[i] = io[i];
  const size_t lds = static_cast<size_t>(n_slots) * kRows * (vec_width + kPad) * sizeof(float);  // one task's slot file
  if (lds > 160 * 1024) return MMF_ETOOLARGE;
  auto k = traj_program_kernel;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
  if (e != hipSuccess) return static_cast<int>(e);
  const int tasks = (R + kRows - 1) / kRows;
  // a 5-slot, 64-wide program takes 21 KB per task: seven workgroups (28 waves) share a CU
  const int per_cu = static_cast<int>((160 * 1024) / lds) < 8 ? static_cast<int>((160 * 1024) / lds) : 8;
  int grid = tasks < 256 * per_cu ? tasks : 256 * per_cu;
  const int waves = kWaves;
  k<<<grid, waves * MMF_WAVE, lds, static_cast<hipStream_t>(stream)>>>(prog, n_instr, weights, p, R, n_slots, vec_width);
  MMF_CHECK_LAUNCH();
  return 0;
}
