// Shared device helpers for the gfx950 filter kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mmf.h"

#define MMF_WAVE 64

#define MMF_CHECK_LAUNCH()                          \
  do {                                              \
    hipError_t _e = hipGetLastError();              \
    if (_e != hipSuccess) return static_cast<int>(_e); \
  } while (0)

// the persistent small-problem step loop (pf_persistent.inc, compiled with particle_net.hip); reached through
// mmf_pf_forward_loop when MmfPfLoopArgs.persistent is set
int mmf_internal_pf_persistent(const MmfPfLoopArgs* args, void* stream);
int mmf_internal_ekf_persistent(const MmfEkfLoopArgs* args, void* stream);  /* ekf_persistent.inc */
#define MMF_INTERNAL_NOT_RESIDENT (-1000)  /* its grid would not be co-resident on this device: take the launch path */

// the exact-fp32 three-pass backward of the native training recursion over f16 recompute buffers (particle_net_train.inc)
int mmf_internal_train_forward_h(const float* packed, int n_res, int kind, const float* states, const float* traj_bias,
                                 void* stash_h, uint32_t* mask, float* out, int N, int M, int d, void* stream);
int mmf_internal_train_backward_h(const float* packed_t, const float* head_w, int n_res, int kind, const uint32_t* mask,
                                  const float* d_out, void* dz_h, float* dz_scale, float* d_states, int R, int d, void* stream);
int mmf_internal_weight_grads_h(const void* dz_h, const float* dz_scale, const void* stash_h, float* partial_w, float* partial_b,
                                int n_layers, int R, int n_splits, int accumulate, void* stream);
int mmf_internal_small_grads_h(const void* dz_first_h, const float* sc_first, const void* dz_join_h, const float* sc_join,
                               const void* h_last_h, const float* states, const float* d_out, float* p_first, float* p_head,
                               float* p_dout, float* p_traj, int N, int M, int d, int n_out, int n_slices, void* stream);

namespace mmf {

// ---- wave-wide reductions and scans on DPP (row / bank data paths of the VALU) instead of ds_bpermute:
// a __shfl_xor is an LDS-pipe instruction (~100+ cycles of latency each, 16 waves of K1 sharing the pipe);
// the DPP forms are plain VALU operands.  Tree of the reductions: xor 1, 2 (quad_perm), 4 (row_half_mirror
// on quad-uniform values), 8 (row_mirror), then row_bcast15 / row_bcast31 carry the row totals to row 3 --
// lane 63 holds ((r3 + r2) + (r1 + r0)) with every r_k an ascending-xor butterfly of its 16 lanes, i.e. the
// result of the butterfly with offsets 1, 2, 4, 8, 16, 32 (oracle/strict restates that tree); it is
// broadcast with v_readlane.
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E, kDppRowHalfMirror = 0x141, kDppRowMirror = 0x140;
constexpr int kDppRowBcast15 = 0x142, kDppRowBcast31 = 0x143;
constexpr int kDppRowShr1 = 0x111, kDppRowShr2 = 0x112, kDppRowShr4 = 0x114, kDppRowShr8 = 0x118;
constexpr int kDppWaveShr1 = 0x138;  // lane l reads lane l - 1 across the whole wave (lane 0: fill)

// lanes whose DPP source is masked off (row_mask) or out of the row (shifts) get `fill`
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_f32(float fill, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ unsigned dpp_u32(unsigned fill, unsigned v) {
  return static_cast<unsigned>(__builtin_amdgcn_update_dpp(static_cast<int>(fill), static_cast<int>(v), CTRL, ROW_MASK, 0xf, false));
}

// Copy N4 float4 from global memory to LDS with THREADS threads, BATCH loads in flight per thread.  The plain
// `for (i = tid; i < n; i += threads) dst[i] = src[i]` compiles to load -> wait -> store per iteration (hipcc does
// not unroll it): 15-19 dependent L2 round trips, ~6 us, before a workgroup's first MFMA -- half the duration of
// a per-particle-network launch at the reference's 32 x 300 size and ~4 % of one at 256 x 4096.  Every trip count
// is a compile-time constant, so the staging registers never become an indexed (scratch) array.
template <int N4, int THREADS, int BATCH = 8>
__device__ __forceinline__ void stage_to_lds(const float4* __restrict__ src, float4* __restrict__ dst, int tid) {
  constexpr int K = (N4 + THREADS - 1) / THREADS;
#pragma unroll
  for (int b = 0; b < K; b += BATCH) {
    float4 t[BATCH];
#pragma unroll
    for (int k = 0; k < BATCH; ++k)
      if (b + k < K) {
        const int i = tid + (b + k) * THREADS;
        if ((b + k + 1) * THREADS <= N4 || i < N4) t[k] = src[i];
      }
#pragma unroll
    for (int k = 0; k < BATCH; ++k)
      if (b + k < K) {
        const int i = tid + (b + k) * THREADS;
        if ((b + k + 1) * THREADS <= N4 || i < N4) dst[i] = t[k];
      }
  }
}

__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_f32<kDppXor1>(v, v));
  v = fmaxf(v, dpp_f32<kDppXor2>(v, v));
  v = fmaxf(v, dpp_f32<kDppRowHalfMirror>(v, v));
  v = fmaxf(v, dpp_f32<kDppRowMirror>(v, v));
  v = fmaxf(v, dpp_f32<kDppRowBcast15, 0xa>(v, v));
  v = fmaxf(v, dpp_f32<kDppRowBcast31, 0xc>(v, v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma clang fp contract(off)
  v = v + dpp_f32<kDppXor1>(0.f, v);
  v = v + dpp_f32<kDppXor2>(0.f, v);
  v = v + dpp_f32<kDppRowHalfMirror>(0.f, v);
  v = v + dpp_f32<kDppRowMirror>(0.f, v);
  v = v + dpp_f32<kDppRowBcast15, 0xa>(0.f, v);
  v = v + dpp_f32<kDppRowBcast31, 0xc>(0.f, v);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Inclusive scan across the 64 lanes of a wave (u32; Kogge-Stone inside each row of 16 with row_shr, then the
// row totals through row_bcast15 / row_bcast31).  The caller guarantees the wave total fits 32 bits.
__device__ __forceinline__ unsigned wave_inclusive_scan_u32(unsigned v) {
  v += dpp_u32<kDppRowShr1>(0u, v);
  v += dpp_u32<kDppRowShr2>(0u, v);
  v += dpp_u32<kDppRowShr4>(0u, v);
  v += dpp_u32<kDppRowShr8>(0u, v);
  v += dpp_u32<kDppRowBcast15, 0xa>(0u, v);
  v += dpp_u32<kDppRowBcast31, 0xc>(0u, v);
  return v;
}

// Inclusive prefix maximum across the 64 lanes (same data movement as the sum scan)
__device__ __forceinline__ unsigned wave_inclusive_scan_max_u32(unsigned v) {
  v = max(v, dpp_u32<kDppRowShr1>(0u, v));
  v = max(v, dpp_u32<kDppRowShr2>(0u, v));
  v = max(v, dpp_u32<kDppRowShr4>(0u, v));
  v = max(v, dpp_u32<kDppRowShr8>(0u, v));
  v = max(v, dpp_u32<kDppRowBcast15, 0xa>(0u, v));
  v = max(v, dpp_u32<kDppRowBcast31, 0xc>(0u, v));
  return v;
}

// u64 inclusive scan of per-lane values below 2^40: two u32 scans of the low 20 and the high 20 bits (each
// wave total below 2^26), recombined -- exact, no carries to propagate across lanes.
__device__ __forceinline__ unsigned long long wave_inclusive_scan(unsigned long long v, int /*lane*/) {
  const unsigned lo = wave_inclusive_scan_u32(static_cast<unsigned>(v) & 0xFFFFFu);
  const unsigned hi = wave_inclusive_scan_u32(static_cast<unsigned>(v >> 20));
  return (static_cast<unsigned long long>(hi) << 20) + lo;
}

// Deterministic fp32 exp for x <= 0: the exact operation sequence of
// oracle/resample.py::detexp (separate, un-fused multiplies and adds, Horner order), so the
// fixed-point weights -- and therefore the resampled indices -- match the oracle bit for bit.
__device__ __forceinline__ float detexp(float x) {
#pragma clang fp contract(off)
  const float LOG2E = 1.4426950408889634f;
  float t = x * LOG2E;
  t = fmaxf(t, -126.0f);  // fmaxf(NaN, c) = c; -inf -> -126
  float n = rintf(t);     // v_rndne_f32: round half to even
  float f = t - n;
  float p = 0.00015403530393381608f;
  p = p * f;
  p = p + 0.0013333558146428443f;
  p = p * f;
  p = p + 0.009618129107628477f;
  p = p * f;
  p = p + 0.05550410866482158f;
  p = p * f;
  p = p + 0.2402265069591007f;
  p = p * f;
  p = p + 0.6931471805599453f;
  p = p * f;
  p = p + 1.0f;
  float scale = __int_as_float((static_cast<int>(n) + 127) << 23);
  return p * scale;
}

}  // namespace mmf
