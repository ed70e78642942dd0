// Tile machinery of the per-particle networks (K2 / K5 / K6): types, blob layout, the register-resident
// 64 x 64 layer chains on v_mfma_f32_32x32x2_f32 (exact fp32) and v_mfma_f32_32x32x16_f16 (f16x3).  Shared by
// particle_net.hip (inference kernels, K6 step kernels, the persistent loop) and particle_net_fused.hip (the fused
// training kernel); every definition is internal to the including translation unit.
#pragma once
#include <hip/hip_fp16.h>

#include <utility>

#include "mmf_common.h"

namespace {


using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using half8 = __attribute__((ext_vector_type(8))) _Float16;
using half2v = __attribute__((ext_vector_type(2))) _Float16;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

constexpr int kUnits = MMF_UNITS;
constexpr int kW0Cols = 8;                    // first layer padded to K = 8 (state dims, 1, zeros)
constexpr int kHeadRows = MMF_MAX_STATE_DIM + 1;
constexpr int kLayerFloats = kUnits * kUnits;
constexpr float kF16SplitMax = 65504.0f;      // hi = RTZ_f16(x) must stay finite and unsaturated

__host__ __device__ constexpr int num_layers(int n_res) { return 3 + 2 * n_res; }
__host__ __device__ constexpr int off_w0() { return 0; }
__host__ __device__ constexpr int off_layers() { return kUnits * kW0Cols; }
__host__ __device__ constexpr int off_bias(int n_res) { return off_layers() + num_layers(n_res) * kLayerFloats; }
__host__ __device__ constexpr int off_whead(int n_res) { return off_bias(n_res) + num_layers(n_res) * kUnits; }
__host__ __device__ constexpr int off_bhead(int n_res) { return off_whead(n_res) + kHeadRows * kUnits; }
__host__ __device__ constexpr int blob_floats(int n_res) { return off_bhead(n_res) + 8; }

// feature row held by accumulator register r of a 32-row tile, for lane half h
__host__ __device__ constexpr int rowmap(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// ------------------------------------------------------------------------------ packing
// f16x3 layers: feature index fed by element i of lane half h in k-step s (16 features per
// step): the rows accumulator registers 8(s&1) .. 8(s&1)+7 of input tile s>>1 hold
__host__ __device__ constexpr int kmap16(int s, int h, int i) {
  return 32 * (s >> 1) + 16 * (s & 1) + (i & 3) + 8 * (i >> 2) + 4 * h;
}

// ---- the dual-use weight image of the fused training kernel (MMF_PREC_F16X3_DUAL): a 64 x 64 layer as 64 rows of
// 256 B = [hi halves of the row | lo halves], 16-byte chunk `ch` of row `row` at byte 256 row + 16 (ch ^ swizzle(row)).
// Inside a row the columns are stored in 4-element runs ordered so that the 8 k-values one lane feeds a k-step of
// v_mfma_f32_32x32x16_f16 (kmap16) are ONE chunk: run index = 4 (c >> 4) + 2 ((c >> 2) & 1) + ((c >> 3) & 1).
// The swizzle makes the forward's row reads (ds_read_b128, lane = row) and the backward's transposed reads
// (ds_read_b64_tr_b16: 4 rows x 16 columns per 16 lanes) both bank-conflict free (checked by enumeration:
// scripts/debug/lds_bank_check.py).
__host__ __device__ constexpr int dual_swizzle(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
__host__ __device__ constexpr int dual_off(int row, int ch) { return 256 * row + 16 * (ch ^ dual_swizzle(row)); }

__device__ __forceinline__ unsigned short f16_bits_rz(float x) {
  return __half_as_ushort(__float2half_rn(x));
}

// ------------------------------------------------------------------------------ layer pieces
template <int CT>
struct Act {  // one 64-feature activation for 32*CT particles: [row tile][col tile]
  f32x16 v[2][CT];
};

// quad broadcast of lane (l & ~3) -- the primal column of a Jacobian group
__device__ __forceinline__ float quad_first(float x) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x00, 0xf, 0xf, true));
}

// The epilogue of a Jacobian column group, spelled out (explicit fused multiply-adds, no implicit contraction): the launch
// kernel and the persistent EKF loop (ekf_persistent.inc) must produce the same bits whatever surrounds these statements.
// x' = x + dir sigmoid(gate);  d x'_i / d x_c = d dir_i s + dir_i s (1 - s) d gate + [i == c]
__device__ __forceinline__ float jac_primal(float x, float dir, float sg) {
#pragma clang fp contract(off)
  return __builtin_fmaf(dir, sg, x);
}
__device__ __forceinline__ float jac_tangent(float ddir, float sg, float dir, float dgate, float delta) {
#pragma clang fp contract(off)
  const float u = (dir * (sg * (1.0f - sg))) * dgate;
  return __builtin_fmaf(ddir, sg, u) + delta;
}

template <int CT>
__device__ __forceinline__ void mfma_layer(const float* __restrict__ Wl, const Act<CT>& in,
                                           Act<CT>& acc, int lane) {
  // The blob in LDS is loop-invariant across tiles; without a compiler barrier LICM hoists
  // every layer's fragment reads out of the tile loop (hundreds of VGPRs -> scratch spills).
  asm volatile("" ::: "memory");
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int s4 = 0; s4 < 8; ++s4) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(Wl + ((t * 8 + s4) * 64 + lane) * 4);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int s = s4 * 4 + ks;
#pragma unroll
        for (int c = 0; c < CT; ++c)
          acc.v[t][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ks], in.v[s >> 4][c][s & 15],
                                                             acc.v[t][c], 0, 0, 0);
      }
    }
  }
}

// acc (+)= bias (LDS, natural order).  `scale` zeroes the bias on Jacobian tangent columns.
template <int CT, bool ADD>
__device__ __forceinline__ void add_bias(const float* __restrict__ bl, Act<CT>& acc, int h, float scale) {
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(bl + 32 * t + 8 * g + 4 * h);
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (ADD) acc.v[t][c][4 * g + e] += b[e] * scale;
          else acc.v[t][c][4 * g + e] = b[e] * scale;
        }
    }
}

// acc += bias two elements at a time (v_pk_add_f32): the f16x3 path's skip accumulators
template <int CT>
__device__ __forceinline__ void add_bias_packed(const float* __restrict__ bl, Act<CT>& acc, int h) {
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(bl + 32 * t + 8 * g + 4 * h);
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
          f32x2 v = {acc.v[t][c][4 * g + e], acc.v[t][c][4 * g + e + 1]};
          v += f32x2{b[e], b[e + 1]};
          acc.v[t][c][4 * g + e] = v[0];
          acc.v[t][c][4 * g + e + 1] = v[1];
        }
    }
}

// max(v, 0) as ONE instruction: on accumulator outputs hipcc adds a canonicalising v_max in
// front of fmaxf (2 ops per element).  On the raw bits, max((int)v, 0) is the same function
// for every non-NaN float (negative floats are negative ints; -0.0 -> +0.0) and is a single
// v_max_i32.  (Not inline asm: hipcc pads no MFMA->VALU wait states around asm operands.)
__device__ __forceinline__ float relu1(float v) {
  return __int_as_float(max(__float_as_int(v), 0));
}

// f16x3 layers: ReLU AND saturation at the largest f16 in ONE instruction (v_med3_f32; no canonicalisation is
// emitted in front of the builtin).  An activation beyond the f16 range then splits into hi = 65504 (0x7BFF,
// which the range tracking reports) and a finite residual instead of hi = +inf, lo = -inf, whose products are
// NaN: the failure mode of the mode is a raised flag over FINITE outputs.  v_med3_f32 returns min3 when an
// operand is NaN, i.e. it would swallow a NaN: the two ReLUs that see externally supplied numbers first (the
// first layer on the particle states, the one after the join layer on the per-trajectory term) therefore stay
// NaN-keeping (the first layer's inputs are tested directly, the ReLU after the join layer is relu_keepnan), so
// that a NaN / inf input reaches the next operand split and raises the flag there (0x7C00 / 0x7E00 >= 0x7BFF),
// and every later ReLU clamps whatever those produce.
__device__ __forceinline__ float relu_sat(float v) { return __builtin_amdgcn_fmed3f(v, 0.f, kF16SplitMax); }
// ReLU that keeps a NaN of EITHER sign (relu1 turns a negative NaN into 0): compare + select, used once per network
__device__ __forceinline__ float relu_keepnan(float v) { return v <= 0.f ? 0.f : v; }
// signed operands (dynamics trunk entry, Jacobian tangents): saturate both ways; a NaN becomes -65504 (flagged)
__device__ __forceinline__ float clamp_sat(float v) { return __builtin_amdgcn_fmed3f(v, -kF16SplitMax, kF16SplitMax); }

template <int CT, bool JAC, bool SAT = false>
__device__ __forceinline__ void relu(Act<CT>& a, bool primal) {
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = a.v[t][c][r];
        if (JAC) {
          // tangent columns follow the primal's mask (sub-gradient 0 at 0, as autograd)
          const float pv = quad_first(v);
          const float keep = pv > 0.f ? v : 0.f;
          a.v[t][c][r] = primal ? (SAT ? relu_sat(v) : relu1(v)) : (SAT ? clamp_sat(keep) : keep);
        } else {
          a.v[t][c][r] = SAT ? relu_sat(v) : relu1(v);
        }
      }
}

// saturate a signed activation in place (f16x3: the one operand split that follows no ReLU)
template <int CT>
__device__ __forceinline__ void saturate(Act<CT>& a) {
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) a.v[t][c][r] = clamp_sat(a.v[t][c][r]);
}

// y = relu(W2 relu(W1 x + b1) + b2 + x), in place in x, h as scratch (resblocks.Linear)
template <int CT, bool JAC>
__device__ __forceinline__ void res_block(const float* __restrict__ lds, int n_res, int l1,
                                          Act<CT>& x, Act<CT>& hbuf, int lane, bool primal) {
  const int h = lane >> 5;
  const float bs = (JAC && !primal) ? 0.f : 1.f;
  add_bias<CT, false>(lds + off_bias(n_res) + l1 * kUnits, hbuf, h, bs);
  mfma_layer<CT>(lds + off_layers() + l1 * kLayerFloats, x, hbuf, lane);
  relu<CT, JAC>(hbuf, primal);
  add_bias<CT, true>(lds + off_bias(n_res) + (l1 + 1) * kUnits, x, h, bs);
  mfma_layer<CT>(lds + off_layers() + (l1 + 1) * kLayerFloats, hbuf, x, lane);
  relu<CT, JAC>(x, primal);
}

// ------------------------------------------------------------------------------ f16x3 path
// Each fp32 operand is split exactly into two halves x = hi + lo + O(2^-22 x) (round-toward-
// zero, so the split never overflows to inf) and a product is evaluated as
// hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16 with fp32 accumulation: 3 MFMAs at 16x the
// f32-MFMA rate, ~22-bit operands (measured parity: tests/test_gpu_kernels.py).
template <int CT>
struct SplitAct {
  half8 hi[4][CT], lo[4][CT];  // [k-step][col tile]
};


// hi = RTZ_f16(x); lo = RTZ_f16(x - hi).  `neg_one` is -1.0f held in an SGPR the optimiser
// cannot see through, so that fma(float(hi), neg_one, x) selects v_fma_mix_f32 (f16 source read
// straight from the packed register, f32 arithmetic): 4 instructions per pair instead of
// cvt_pkrtz + 2 cvt_f32_f16 + pk_add + cvt_pkrtz.  (v_fma_mixlo/hi_f16 would fold the final
// conversion as well but issue at half rate on gfx950: scripts/ubench/valu_rate.hip.)
// x - hi is exact in fp32 (the residual has <= 13 significant bits).
using half2v = __attribute__((ext_vector_type(2))) _Float16;
using f32x2v = __attribute__((ext_vector_type(2))) float;
__device__ __forceinline__ void split_pair(float x0, float x1, float neg_one, unsigned& hi, unsigned& lo) {
  const half2v h = __builtin_convertvector(f32x2v{x0, x1}, half2v);  // v_cvt_pk_f16_f32: round to nearest even
  const float r0 = __builtin_fmaf(static_cast<float>(h[0]), neg_one, x0);
  const float r1 = __builtin_fmaf(static_cast<float>(h[1]), neg_one, x1);
  const half2v l = __builtin_convertvector(f32x2v{r0, r1}, half2v);
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, l);
}

// Range tracking on the packed hi halves: round-to-nearest maps every |x| >= 65504 to 0x7BFF or to
// +inf (0x7C00), both >= the threshold below, and for non-negative halves the i16 order is the f16 order, so ONE
// v_pk_max_i16 per pair keeps the running maximum (fmaxf on the fp32 values costs 3 ops per
// pair once canonicalisation is counted).  Negative halves compare below zero and are ignored,
// which is right after a ReLU; the one split that sees signed values (dynamics trunk entry, no
// ReLU after the join layer) passes SIGNED and masks the sign bits first.
using short2v = __attribute__((ext_vector_type(2))) short;
constexpr short kF16Saturated = 0x7BFF;  // also below +inf (0x7C00) and every NaN pattern

template <int CT, bool SIGNED = false>
__device__ __forceinline__ void split_act(const Act<CT>& x, SplitAct<CT>& o, float neg_one, short2v& amax) {
#pragma unroll
  for (int tp = 0; tp < 2; ++tp)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        u32x4 h, l;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          unsigned hh, ll;
          split_pair(x.v[tp][c][8 * u + 2 * p], x.v[tp][c][8 * u + 2 * p + 1], neg_one, hh, ll);
          h[p] = hh;
          l[p] = ll;
        }
        // a short tree per fragment, then one link of the running chain
        constexpr unsigned kMask = SIGNED ? 0x7fff7fffu : 0xffffffffu;
        const short2v m01 = __builtin_elementwise_max(__builtin_bit_cast(short2v, h[0] & kMask),
                                                      __builtin_bit_cast(short2v, h[1] & kMask));
        const short2v m23 = __builtin_elementwise_max(__builtin_bit_cast(short2v, h[2] & kMask),
                                                      __builtin_bit_cast(short2v, h[3] & kMask));
        amax = __builtin_elementwise_max(amax, __builtin_elementwise_max(m01, m23));
        o.hi[2 * tp + u][c] = __builtin_bit_cast(half8, h);
        o.lo[2 * tp + u][c] = __builtin_bit_cast(half8, l);
      }
  // Pin the running maximum here (no instruction is emitted): left alone, the compiler sinks
  // every v_pk_max_i16 to the flag test at the end of the tile and keeps the hi fragments of
  // all seven layers alive for it -- in scratch.
  unsigned pin = __builtin_bit_cast(unsigned, amax);
  asm volatile("" : "+v"(pin));
  amax = __builtin_bit_cast(short2v, pin);
}

template <int CT>
__device__ __forceinline__ void mfma_layer_f16(const float* __restrict__ Wl, const SplitAct<CT>& in,
                                               Act<CT>& acc, int lane) {
  asm volatile("" ::: "memory");  // see mfma_layer: keep LICM from hoisting the fragment reads
  const unsigned char* base = reinterpret_cast<const unsigned char*>(Wl) + lane * 16;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const half8 ahi = *reinterpret_cast<const half8*>(base + ((t * 4 + s) * 2 + 0) * 1024);
      const half8 alo = *reinterpret_cast<const half8*>(base + ((t * 4 + s) * 2 + 1) * 1024);
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        acc.v[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, in.hi[s][c], acc.v[t][c], 0, 0, 0);
        acc.v[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, in.lo[s][c], acc.v[t][c], 0, 0, 0);
        acc.v[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo, in.hi[s][c], acc.v[t][c], 0, 0, 0);
      }
    }
}

// JAC: the tile holds groups of {primal, tangents}: tangent columns take no bias, follow the primal's
// ReLU mask and are signed (their splits mask the sign bits for the range tracking).
template <int CT, bool SIGNED = false, bool JAC = false>
__device__ __forceinline__ void res_block_f16(const float* __restrict__ lds, int n_res, int l1,
                                              Act<CT>& x, Act<CT>& hbuf, SplitAct<CT>& sp, int lane,
                                              float neg_one, short2v& amax, bool primal = true) {
  const int h = lane >> 5;
  const float bs = (JAC && !primal) ? 0.f : 1.f;
  if constexpr (SIGNED) saturate<CT>(x);  // no ReLU in front of this split: saturate the signed values themselves
  split_act<CT, SIGNED || JAC>(x, sp, neg_one, amax);
  add_bias<CT, false>(lds + off_bias(n_res) + l1 * kUnits, hbuf, h, bs);
  mfma_layer_f16<CT>(lds + off_layers() + l1 * kLayerFloats, sp, hbuf, lane);
  relu<CT, JAC, true>(hbuf, primal);
  split_act<CT, JAC>(hbuf, sp, neg_one, amax);
  if constexpr (JAC) add_bias<CT, true>(lds + off_bias(n_res) + (l1 + 1) * kUnits, x, h, bs);
  else add_bias_packed<CT>(lds + off_bias(n_res) + (l1 + 1) * kUnits, x, h);
  mfma_layer_f16<CT>(lds + off_layers() + (l1 + 1) * kLayerFloats, sp, x, lane);
  relu<CT, JAC, true>(x, primal);
}

// ------------------------------------------------------------------ f16x3, pipelined halves
// The 64-particle tile is processed as two 32-particle halves whose layers are offset by half
// a layer: while the matrix pipe runs the 24 MFMAs of one half's layer, the wave issues the
// other half's ReLU / operand split / range tracking in their shadow.  An MFMA holds the
// SIMD's vector issue for 8 of its 32 cycles; up to six of these VALU instructions per MFMA
// are free when they are placed between independent MFMAs (scripts/ubench/mfma_fill.hip), and
// the split needs 4.7.  The order is pinned with sched_group_barrier; left to the scheduler
// (and in the unpipelined kernel, where a layer's VALU depends on its own MFMAs) the two kinds
// of work run back to back.
template <int... Is, class F>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

template <int C, bool SAT = true>
__device__ __forceinline__ void relu_half(Act<2>& a) {
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) a.v[t][C][r] = SAT ? relu_sat(a.v[t][C][r]) : relu_keepnan(a.v[t][C][r]);
}

template <int C>
__device__ __forceinline__ void saturate_half(Act<2>& a) {
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) a.v[t][C][r] = clamp_sat(a.v[t][C][r]);
}

template <int C, bool SIGNED>
__device__ __forceinline__ void split_half(const Act<2>& x, SplitAct<2>& o, float neg_one, short2v& amax) {
#pragma unroll
  for (int tp = 0; tp < 2; ++tp)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      u32x4 h, l;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        unsigned hh, ll;
        split_pair(x.v[tp][C][8 * u + 2 * p], x.v[tp][C][8 * u + 2 * p + 1], neg_one, hh, ll);
        h[p] = hh;
        l[p] = ll;
      }
      constexpr unsigned kMask = SIGNED ? 0x7fff7fffu : 0xffffffffu;
      const short2v m01 = __builtin_elementwise_max(__builtin_bit_cast(short2v, h[0] & kMask),
                                                    __builtin_bit_cast(short2v, h[1] & kMask));
      const short2v m23 = __builtin_elementwise_max(__builtin_bit_cast(short2v, h[2] & kMask),
                                                    __builtin_bit_cast(short2v, h[3] & kMask));
      amax = __builtin_elementwise_max(amax, __builtin_elementwise_max(m01, m23));
      o.hi[2 * tp + u][C] = __builtin_bit_cast(half8, h);
      o.lo[2 * tp + u][C] = __builtin_bit_cast(half8, l);
    }
  unsigned pin = __builtin_bit_cast(unsigned, amax);  // see split_act
  asm volatile("" : "+v"(pin));
  amax = __builtin_bit_cast(short2v, pin);
}

template <int C, bool ADD>
__device__ __forceinline__ void bias_half(const float* __restrict__ bl, Act<2>& acc, int h) {
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(bl + 32 * t + 8 * g + 4 * h);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (ADD) acc.v[t][C][4 * g + e] += b[e];
        else acc.v[t][C][4 * g + e] = b[e];
      }
    }
}

// The A fragments (weights) of fragment group g + 1 are read from LDS while group g's MFMAs
// run; the last group of a stage reads the first group of the NEXT stage (`next`), so no stage
// opens with an exposed LDS round trip.
struct FragPair {
  half8 hi, lo;
};
__device__ __forceinline__ FragPair load_frag(const float* __restrict__ Wl, int lane, int g) {
  const unsigned char* base = reinterpret_cast<const unsigned char*>(Wl) + lane * 16 + g * 2048;
  FragPair f;
  f.hi = *reinterpret_cast<const half8*>(base);
  f.lo = *reinterpret_cast<const half8*>(base + 1024);
  return f;
}

template <int C>
__device__ __forceinline__ void mfma_half(const float* __restrict__ Wl, const float* __restrict__ next,
                                          FragPair& cur, const SplitAct<2>& in, Act<2>& acc, int lane) {
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int g = t * 4 + s;
      const FragPair nxt = g < 7 ? load_frag(Wl, lane, g + 1) : load_frag(next, lane, 0);
      acc.v[t][C] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.hi, in.hi[s][C], acc.v[t][C], 0, 0, 0);
      acc.v[t][C] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.hi, in.lo[s][C], acc.v[t][C], 0, 0, 0);
      acc.v[t][C] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.lo, in.hi[s][C], acc.v[t][C], 0, 0, 0);
      cur = nxt;
    }
}

// One region's issue order: 8 fragment groups of {2 LDS reads, 3 x (1 MFMA, VPM VALU)}.
template <int VPM>
__device__ __forceinline__ void pin_mfma_valu_interleave() {
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // DS read
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);  // VALU
    }
  }
}

enum Kind { kDynamics = 0, kMeasure = 1, kJacobian = 2 };

}  // namespace
