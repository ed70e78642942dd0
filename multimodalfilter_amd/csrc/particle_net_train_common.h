// K6 helpers shared by the step kernels (particle_net_train.inc) and the fused kernel (particle_net_fused.hip):
// fp32 / compact f16 row stores of a 32-particle tile, row / tile magnitudes, ReLU masks as bits.
#pragma once
#include "particle_net_tiles.h"

namespace {

__device__ __forceinline__ void stash_store(float* __restrict__ base, const Act<1>& a, int row, bool valid, int h) {
  if (!valid) return;
  float* p = base + static_cast<size_t>(row) * kUnits + 4 * h;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 v = {a.v[t][0][4 * g], a.v[t][0][4 * g + 1], a.v[t][0][4 * g + 2], a.v[t][0][4 * g + 3]};
      *reinterpret_cast<f32x4*>(p + 32 * t + 8 * g) = v;
    }
}

// COMPACT (round 4): the recompute buffers of the native training recursion in HALF the bytes -- they are what the
// recursion's HBM traffic consists of (written once, read once by the weight-gradient pass: ~10 KB per particle and
// network call in fp32 against 44 B of algorithmic input).  Activations are stored as f16 (post-ReLU values of a
// normalised network: 2^-11 relative), pre-activation gradients as f16 RELATIVE TO THEIR ROW'S LARGEST MAGNITUDE (one
// fp32 scale per row and layer: gradients of 1e-7 would underflow plain f16); the weight-gradient kernel multiplies
// the f16 values on the f16 MFMA (exact products, fp32 accumulation), the narrow reductions convert back to fp32.
__device__ __forceinline__ void stash_store_h(_Float16* __restrict__ base, const Act<1>& a, int row, bool valid, int h) {
  if (!valid) return;
  _Float16* p = base + static_cast<size_t>(row) * kUnits + 4 * h;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      using half4 = __attribute__((ext_vector_type(4))) _Float16;
      half4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = static_cast<_Float16>(fminf(a.v[t][0][4 * g + e], kF16SplitMax));
      *reinterpret_cast<half4*>(p + 32 * t + 8 * g) = v;
    }
}

// dz rows of one wave's 32-row tile -> f16 relative to the TILE's largest magnitude; `scale` (R) receives that
// magnitude for every row of the tile (tiles start at multiples of 32 rows: the weight-gradient kernel multiplies a
// tile's f16 product by ONE scale, the narrow reductions read it per row).  An element's error is
// max(2^-11 |v|, 2^-25 tile max): the sums over rows these buffers feed are dominated by the large rows.
// largest magnitude of this lane's ROW (lanes j and j + 32 hold the two halves of row j's 64 features) ..
__device__ __forceinline__ float row_absmax(const Act<1>& a) {
  float m = 0.f;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) m = fmaxf(m, fabsf(a.v[t][0][r]));
  return fmaxf(m, __shfl_xor(m, 32));
}
// .. and of the wave's 32-row tile (rows past R repeat row R - 1)
__device__ __forceinline__ float tile_absmax(float row_max) { return mmf::wave_max(row_max); }

__device__ __forceinline__ void dz_store_h(_Float16* __restrict__ base, float* __restrict__ scale, const Act<1>& a, int row,
                                           bool valid, int h, float m) {
  const float sc = (m > 0.f && m < 3.0e38f) ? m : 1.f;
  const float inv = 1.0f / sc;
  if (!valid) return;
  if (h == 0) scale[row] = sc;
  _Float16* p = base + static_cast<size_t>(row) * kUnits + 4 * h;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      using half4 = __attribute__((ext_vector_type(4))) _Float16;
      half4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = static_cast<_Float16>(a.v[t][0][4 * g + e] * inv);
      *reinterpret_cast<half4*>(p + 32 * t + 8 * g) = v;
    }
}

// ReLU masks as BITS: lane (j, h) owns 32 of its particle's 64 features (register r of tile t <-> bit 16 t + r),
// so the masks of one layer are two u32 per particle -- 8 B instead of the 256 B of the stashed activation
// the backward used to re-read only to test its sign.  Layout (NL + 1, R, 2) u32, word h of row `row`.
__device__ __forceinline__ void mask_store(unsigned* __restrict__ base, const Act<1>& a, int row, bool valid, int h) {
  unsigned w = 0u;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) w |= (a.v[t][0][r] > 0.f ? 1u : 0u) << (16 * t + r);
  if (valid) base[static_cast<size_t>(row) * 2 + h] = w;
}

// g *= [activation > 0]  (autograd's ReLU sub-gradient: 0 at 0)
__device__ __forceinline__ void mask_by_word(unsigned w, Act<1>& g) {
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) g.v[t][0][r] = ((w >> (16 * t + r)) & 1u) ? g.v[t][0][r] : 0.f;
}

__device__ __forceinline__ void mask_by_bits(const unsigned* __restrict__ base, Act<1>& g, int row, int h) {
  const unsigned w = base[static_cast<size_t>(row) * 2 + h];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) g.v[t][0][r] = ((w >> (16 * t + r)) & 1u) ? g.v[t][0][r] : 0.f;
}

__device__ __forceinline__ void zero_act(Act<1>& a) {
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) a.v[t][0][r] = 0.f;
}

}  // namespace
