// K2 / K5: the per-particle networks (dynamics, measurement, forward-mode Jacobian) as
// register-resident chains of v_mfma_f32_32x32x2_f32 tiles.
//
// Replaces the reference's op-per-layer evaluation over R = N*M rows:
//   /root/reference/crossmodal/door_models/dynamics.py:102-134 (push_models/dynamics.py:34-64)
//   /root/reference/crossmodal/door_models/pf.py:63-107        (push_models/pf.py:65-109)
//   /root/reference/crossmodal/base_models/crossmodal_pf.py:106-139 (modality logsumexp)
// and torchfilter's default autograd DynamicsModel.jacobian.
//
// Mapping.  A wave owns 32*CT particles; a particle is a COLUMN (lane & 31) of every MFMA
// tile, the 64 hidden features are the rows.  For Y = W X (W: 64x64, X: 64 x particles)
//   A operand = W fragment  : lane (i, h) holds W[32t + i][kmap(s, h)]      (from LDS)
//   B operand = X           : lane (j, h) holds X[kmap(s, h)][j]            (a register)
//   C/D       = Y tile t    : lane (j, h), reg r holds Y[32t + (r&3) + 8(r>>2) + 4h][j]
// Choosing kmap(s, h) = 32(s>>4) + ((s&15)&3) + 8((s&15)>>2) + 4h makes register (s & 15) of
// output tile (s >> 4) of one layer *be* the B operand of k-step s of the next layer: the whole
// network runs without moving an activation between lanes, LDS or HBM.  Weights are packed in
// that fragment order once (mmf_pack_particle_net) and a launch keeps one network's blob
// (<= 149.5 KiB) in LDS; each lane fetches four k-steps of A with one ds_read_b128.
//
// Roofline: 2 FLOP/MAC * {37,312 dynamics | 28,928 measurement} MAC per particle against
// ~32 B of HBM traffic => compute bound on the f32 MFMA peak (157.3 TFLOP/s); DESIGN.md.
#include <hip/hip_fp16.h>
#include <stdlib.h>

#include <cmath>
#include <cstdio>
#include <vector>

#include <utility>

#include "mmf_common.h"
#include "../../include/mmf_detmath.h"
#include "../../include/mmf_philox.h"

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

__device__ __forceinline__ unsigned short f16_bits_rz(float x) {
  return __half_as_ushort(__float2half_rn(x));
}

__global__ void pack_particle_net_kernel(MmfParticleNetDesc d, float* __restrict__ out, int precision) {
  const int total = blob_floats(d.n_res);
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < total; q += gridDim.x * blockDim.x) {
    float v = 0.f;
    if (q < off_layers()) {
      const int row = q / kW0Cols, k = q % kW0Cols;
      if (k < d.d_in) v = d.w_in[row * d.d_in + k];
      else if (k == d.d_in) v = d.b_in[row];
    } else if (q < off_bias(d.n_res)) {
      const int rel = q - off_layers();
      const int l = rel / kLayerFloats, e = rel % kLayerFloats;
      const float* W;
      int stride = kUnits, coff = 0;
      if (l < 2) W = d.w_enc[l];
      else if (l == 2) { W = d.w_join; stride = d.join_in; coff = d.join_state_off; }
      else W = d.w_res[l - 3];
      if (precision == MMF_PREC_F32) {
        const int ks = e & 3, lane = (e >> 2) & 63, s4 = (e >> 8) & 7, t = e >> 11;
        const int s = 4 * s4 + ks, h = lane >> 5, i = lane & 31;
        const int k = 32 * (s >> 4) + rowmap(s & 15, h);
        v = W[(32 * t + i) * stride + coff + k];
      } else {
        // two halves per float slot; layout [t][s][hi|lo][lane][8]
        unsigned short hb[2];
        for (int z = 0; z < 2; ++z) {
          const int he = 2 * e + z;
          const int i = he & 7, lane = (he >> 3) & 63, part = (he >> 9) & 1, s = (he >> 10) & 3, t = he >> 12;
          const float w = W[(32 * t + (lane & 31)) * stride + coff + kmap16(s, lane >> 5, i)];
          const __half hi = __float2half_rn(w);
          hb[z] = part ? f16_bits_rz(w - __half2float(hi)) : __half_as_ushort(hi);
        }
        v = __uint_as_float(static_cast<unsigned>(hb[0]) | (static_cast<unsigned>(hb[1]) << 16));
      }
    } else if (q < off_whead(d.n_res)) {
      const int rel = q - off_bias(d.n_res);
      const int l = rel / kUnits, row = rel % kUnits;
      if (l < 2) v = d.b_enc[l][row];
      else if (l > 2) v = d.b_res[l - 3][row];  // join bias travels in traj_bias
    } else if (q < off_bhead(d.n_res)) {
      const int rel = q - off_whead(d.n_res);
      const int o = rel / kUnits, k = rel % kUnits;
      if (o < d.n_out) v = d.w_head[o * kUnits + k];
    } else {
      const int o = q - off_bhead(d.n_res);
      if (o < d.n_out) v = d.b_head[o];
    }
    out[q] = v;
  }
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
#ifdef MMF_EXP_NO_SPLIT_LO  // scripts/k2_experiments.sh: what do the 3 residual instructions cost? (wrong results)
  const half2v l = h;
#else
  const float r0 = __builtin_fmaf(static_cast<float>(h[0]), neg_one, x0);
  const float r1 = __builtin_fmaf(static_cast<float>(h[1]), neg_one, x1);
  const half2v l = __builtin_convertvector(f32x2v{r0, r1}, half2v);
#endif
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
#ifndef MMF_EXP_NO_RANGE  // scripts/k2_experiments.sh: what does range tracking cost?
      constexpr unsigned kMask = SIGNED ? 0x7fff7fffu : 0xffffffffu;
      const short2v m01 = __builtin_elementwise_max(__builtin_bit_cast(short2v, h[0] & kMask),
                                                    __builtin_bit_cast(short2v, h[1] & kMask));
      const short2v m23 = __builtin_elementwise_max(__builtin_bit_cast(short2v, h[2] & kMask),
                                                    __builtin_bit_cast(short2v, h[3] & kMask));
      amax = __builtin_elementwise_max(amax, __builtin_elementwise_max(m01, m23));
#endif
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
#ifndef MMF_EXP_TWO_PRODUCTS  // scripts/k2_experiments.sh: what does the third MFMA cost?
      acc.v[t][C] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.lo, in.hi[s][C], acc.v[t][C], 0, 0, 0);
#endif
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

// ---- the 64x64 layers of one tile of 32*CT particles (f16x3), software-pipelined on ROW tiles.
// res_block_f16 / mfma_layer_f16 (the unpipelined path) issue, per fragment group, two LDS reads, wait for them,
// 3*CT dependent MFMAs -- and the ReLU / operand split of a layer's output only after all of its MFMAs: ~2,000 cycles
// per layer for a lone wave at CT = 1 (768 of them MFMA), which IS a small filter's step time (pf_persistent.inc).
// The column-half pipeline (PIPE) overlaps the two kinds of work perfectly but reads every weight fragment TWICE
// (once per half, half a layer apart); this one keeps the column tiles in lock step -- one fragment read serves all
// CT column tiles -- and pipelines on the output ROW tiles instead.
// Same operations, same order PER ACCUMULATOR (k-steps ascending, hi*hi, hi*lo, lo*hi), so the same bits -- only
// the interleaving changes:
//  * output rows 0..31 (tile 0) feed k-steps 0, 1 of the next layer, rows 32..63 (tile 1) k-steps 2, 3: the
//    fragment groups run (t0,s0) (t0,s1) (t1,s0) (t1,s1) | (t0,s2) (t0,s3) | (t1,s2) (t1,s3), so that tile 1 of the
//    previous layer is post-processed (activation + split + range tracking: 72 VALU) under the first four groups,
//    the next layer's accumulators are initialised under the next two, and tile 0 of this layer is post-processed
//    under the last two (and a little after them);
//  * weight fragments are requested TWO groups ahead, across layer boundaries;
//  * the order is pinned with sched_group_barrier (as particle_net_kernel's PIPE variant does for column halves).
constexpr int kGroupT[8] = {0, 0, 1, 1, 0, 0, 1, 1};
constexpr int kGroupS[8] = {0, 1, 0, 1, 2, 3, 2, 3};

enum SmallAct { kActReluSat = 0, kActReluKeepNan = 1, kActSaturate = 2, kActRelu1 = 3 };

template <int CT, int T, int ACT>
__device__ __forceinline__ void act_tile(Act<CT>& a) {
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v = a.v[T][c][r];
      a.v[T][c][r] = ACT == kActReluSat ? relu_sat(v) : ACT == kActReluKeepNan ? relu_keepnan(v) : ACT == kActSaturate ? clamp_sat(v) : relu1(v);
    }
}

// rows of tile T -> k-steps 2T, 2T + 1 of the next layer's operand (split_act's arithmetic, one row tile)
template <int CT, int T, bool SIGNED>
__device__ __forceinline__ void split_tile(const Act<CT>& x, SplitAct<CT>& o, float neg_one, short2v& amax) {
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      u32x4 hh, ll;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        unsigned a, b;
        split_pair(x.v[T][c][8 * u + 2 * p], x.v[T][c][8 * u + 2 * p + 1], neg_one, a, b);
        hh[p] = a;
        ll[p] = b;
      }
      constexpr unsigned kMask = SIGNED ? 0x7fff7fffu : 0xffffffffu;
      const short2v m01 = __builtin_elementwise_max(__builtin_bit_cast(short2v, hh[0] & kMask), __builtin_bit_cast(short2v, hh[1] & kMask));
      const short2v m23 = __builtin_elementwise_max(__builtin_bit_cast(short2v, hh[2] & kMask), __builtin_bit_cast(short2v, hh[3] & kMask));
      amax = __builtin_elementwise_max(amax, __builtin_elementwise_max(m01, m23));
      o.hi[2 * T + u][c] = __builtin_bit_cast(half8, hh);
      o.lo[2 * T + u][c] = __builtin_bit_cast(half8, ll);
    }
  unsigned pin = __builtin_bit_cast(unsigned, amax);  // see split_act
  asm volatile("" : "+v"(pin));
  amax = __builtin_bit_cast(short2v, pin);
}

template <int CT, int G>
__device__ __forceinline__ void mfma_group(const FragPair& f, const SplitAct<CT>& in, Act<CT>& acc) {
  constexpr int t = kGroupT[G], s = kGroupS[G];
#pragma unroll
  for (int c = 0; c < CT; ++c) {  // ONE fragment pair serves every column tile
    acc.v[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.hi, in.hi[s][c], acc.v[t][c], 0, 0, 0);
    acc.v[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.hi, in.lo[s][c], acc.v[t][c], 0, 0, 0);
    acc.v[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.lo, in.hi[s][c], acc.v[t][c], 0, 0, 0);
  }
}

template <int CT, int GROUPS, int VPM>
__device__ __forceinline__ void pin_small_region() {
#pragma unroll
  for (int g = 0; g < GROUPS; ++g) {
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // the fragment pair of group + 2
#pragma unroll
    for (int m = 0; m < 3 * CT; ++m) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);  // VALU in its shadow
    }
  }
}

// On entry X holds the first layer's pre-activation (d -> 64, bias included); on exit H holds the trunk's output,
// activated (what the head reads).  `join_init(acc)`: writes the per-trajectory term of the join layer.
template <int CT, int NRES, int KIND, class JoinInit>
__device__ __forceinline__ void rowpipe_net_f16(const float* __restrict__ lds, Act<CT>& X, Act<CT>& H, SplitAct<CT>& SP,
                                                JoinInit&& join_init, int lane, float neg_one, short2v& amax) {
  constexpr int NL = 3 + 2 * NRES;
  const int h = lane >> 5;
  const float* layers = lds + off_layers();
  // fragments in flight: two groups ahead for a lone 32-particle tile (a group is 3 MFMAs = 96 cycles, an LDS read
  // ~130), one group ahead with two column tiles (6 MFMAs per group; and the registers are needed elsewhere)
  constexpr int AHEAD = CT == 1 ? 2 : 1;
  FragPair fr[AHEAD + 1];
  // which accumulator layer l writes, and what its OUTPUT goes through on its way into layer l + 1
  auto out_is_h = [](int l) constexpr { return l == 0 || l == 2 || (l > 2 && (l - 3) % 2 == 1); };
  asm volatile("" ::: "memory");  // keep the LDS fragment reads inside the caller's tile loop (see mfma_layer)
  fr[0] = load_frag(layers, lane, kGroupT[0] * 4 + kGroupS[0]);
  if constexpr (AHEAD == 2) fr[1] = load_frag(layers, lane, kGroupT[1] * 4 + kGroupS[1]);
  // prologue: the first layer's ReLU (keeps a NaN state visible: see relu_sat), both tiles split, H = b_0
  act_tile<CT, 0, kActRelu1>(X);
  act_tile<CT, 1, kActRelu1>(X);
  split_tile<CT, 0, false>(X, SP, neg_one, amax);
  split_tile<CT, 1, false>(X, SP, neg_one, amax);
  add_bias<CT, false>(lds + off_bias(NRES), H, h, 1.f);
  __builtin_amdgcn_sched_barrier(0);

  static_for<NL>([&](auto layer) {
    constexpr int l = decltype(layer)::value;
    Act<CT>& out = out_is_h(l) ? H : X;
    // activation of layer l's output on its way into layer l + 1 (the final layer: the ReLU in front of the head)
    constexpr bool to_join_out = (l + 1 == 3);  // the value that leaves the join layer
    constexpr int act_next = (to_join_out && KIND == kMeasure) ? kActReluKeepNan : (to_join_out ? kActSaturate : kActReluSat);
    constexpr bool signed_next = to_join_out && KIND != kMeasure;
    // ... and of layer l - 1's output into this layer (its tile 1 is still pending when this layer starts)
    constexpr bool from_join = (l == 3);
    constexpr int act_in = (from_join && KIND == kMeasure) ? kActReluKeepNan : (from_join ? kActSaturate : kActReluSat);
    constexpr bool signed_in = from_join && KIND != kMeasure;
    auto step = [&](auto gi) {
      constexpr int G = decltype(gi)::value;
      constexpr int i = 8 * l + G;           // global fragment-group counter
      if constexpr (i + AHEAD < 8 * NL) {
        constexpr int l2 = (i + AHEAD) / 8, g2 = (i + AHEAD) % 8;
        fr[(i + AHEAD) % (AHEAD + 1)] = load_frag(layers + l2 * kLayerFloats, lane, kGroupT[g2] * 4 + kGroupS[g2]);
      }
      mfma_group<CT, G>(fr[i % (AHEAD + 1)], SP, out);
    };
    using I = std::integral_constant<int, 0>;
    // ---- region 1: groups 0..3 (k-steps 0, 1) || tile 1 of the previous layer -> k-steps 2, 3
    step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{});
    step(std::integral_constant<int, 2>{}); step(std::integral_constant<int, 3>{});
    if constexpr (l > 0) {
      Act<CT>& prev = out_is_h(l - 1) ? H : X;
      act_tile<CT, 1, act_in>(prev);
      split_tile<CT, 1, signed_in>(prev, SP, neg_one, amax);
    }
    pin_small_region<CT, 4, 6>();
    __builtin_amdgcn_sched_barrier(0);
    // ---- region 2: groups 4, 5 (tile 0 completes) || the next layer's accumulator: bias, skip + bias, or the
    // per-trajectory term (join)
    step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
    if constexpr (l + 1 < NL) {
      constexpr int m = l + 1;
      Act<CT>& nxt = out_is_h(m) ? H : X;
      if constexpr (m == 2) {
        join_init(nxt);
      } else if constexpr (m == 1 || (m > 3 && (m - 3) % 2 == 1)) {
        add_bias_packed<CT>(lds + off_bias(NRES) + m * kUnits, nxt, h);   // the block's skip + bias
      } else {
        add_bias<CT, false>(lds + off_bias(NRES) + m * kUnits, nxt, h, 1.f);
      }
    }
    pin_small_region<CT, 2, 6>();
    __builtin_amdgcn_sched_barrier(0);
    // ---- region 3: groups 6, 7 (tile 1 completes) || tile 0 of this layer -> k-steps 0, 1 of the next
    step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{});
    act_tile<CT, 0, act_next>(out);
    if constexpr (l + 1 < NL) split_tile<CT, 0, signed_next>(out, SP, neg_one, amax);
    pin_small_region<CT, 2, 6>();
    __builtin_amdgcn_sched_barrier(0);
    (void)sizeof(I);
  });
  act_tile<CT, 1, kActReluSat>(H);  // tile 1 of the last layer (the last layer always writes H)
}



struct NetArgs {
  const float* packed;
  const float* states_in;   // (R, D)            [jacobian: (N, D)]
  const float* traj_bias;   // (N, 64)
  const float* noise;       // (R, D) or null
  const float* scale_tril;  // (D, D) or null
  const float* mod_logw;    // (N * stride) or null
  float* states_out;        // (R, D)
  float* loglik;            // (R)
  float* jac;               // (N, D, D)
  int R;                    // rows (columns of the tiles); jacobian: 4 * N
  int M;                    // particles per trajectory (row -> trajectory = row / M)
  int logw_stride;
  int combine;
  int* range_flag;          // f16x3: set to 1 when an activation left the f16-split range
  unsigned long long noise_seed;  // noise_mode 2: counter-based noise (mmf_philox.h), no tensor
  unsigned noise_step, noise_traj0;
  int noise_mode;           // 0: `noise` tensor or none, 2: philox
};

// blockIdx.y selects one of up to MMF_LOOP_MAX_MEAS independent problems of the same shape (the
// sub-filters of a fused EKF evaluate their Jacobians in one launch).  `seq` > 1 instead runs that many
// problems ONE AFTER THE OTHER in every workgroup -- the modalities of a crossmodal particle filter, whose
// second network combines its log-likelihood with the first's (logsumexp in the epilogue): the tile -> wave
// mapping is the same in every pass, so a lane re-reads what it wrote itself, and the launch boundary
// between the two networks (a drained chip, a launch gap) disappears.
struct NetArgsMulti {
  NetArgs a[MMF_LOOP_MAX_MEAS];
  int seq = 1;  // problems a workgroup runs back to back (blockIdx.y covers the rest): see mmf_pf_measure_seq
};

template <int D, int NRES, int KIND, int CT, int PREC, int WPS, bool PIPE = false, bool ROWPIPE = false>
__global__ __launch_bounds__(WPS * 256, WPS) void particle_net_kernel(NetArgsMulti multi) {
#pragma unroll 1
  for (int pass = 0; pass < multi.seq; ++pass) {
  const NetArgs a = multi.a[blockIdx.y * multi.seq + pass];
  if (pass) {
    __threadfence();   // this lane's log-likelihoods of the previous pass, before it reads them back
    __syncthreads();   // every wave is done with the previous network's weights in LDS
  }
  static_assert(!PIPE || (CT == 2 && PREC == MMF_PREC_F16X3 && KIND != kJacobian), "pipelined halves: f16x3, 64-particle tiles");
  static_assert(!ROWPIPE || (!PIPE && PREC == MMF_PREC_F16X3 && KIND != kJacobian), "row-tile pipeline: f16x3");
  constexpr int kThreads = WPS * 256;           // WPS waves per SIMD, one workgroup per CU (LDS)
  constexpr int kWavesPerBlock = kThreads / MMF_WAVE;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr bool JAC = KIND == kJacobian;
  constexpr bool F16 = PREC == MMF_PREC_F16X3;
  constexpr int NOUT = (KIND == kMeasure) ? 1 : D + 1;
  constexpr int TILE = 32 * CT;
  static_assert(!JAC || D <= 3, "jacobian groups are 4 columns: primal + up to 3 tangents");

  // stage this network's fragment-ordered weights in LDS once per workgroup
  {
    const float4* src = reinterpret_cast<const float4*>(a.packed);
    float4* dst = reinterpret_cast<float4*>(lds);
    mmf::stage_to_lds<blob_floats(NRES) / 4, kThreads>(src, dst, threadIdx.x);
  }
  __syncthreads();

  // -1.0f in an SGPR, opaque to the optimiser (see split_pair); the asm emits no instruction
  float neg_one = -1.0f;
  asm volatile("" : "+s"(neg_one));

  const int lane = threadIdx.x & 63;
  const int j = lane & 31, h = lane >> 5;
  const int wave_global = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int waves_total = gridDim.x * kWavesPerBlock;
  const int ntiles = (a.R + TILE - 1) / TILE;

  // The inputs of a tile's first layer (the particle states: up to (D+2)/2 floats per lane and
  // column) are requested one tile ahead: a tile opens with a dependent HBM access otherwise
  // (~1-2 k cycles with nothing else to issue), 8 times per wave at the headline size.
  constexpr int KS0 = (D + 2) / 2;
  auto first_layer_inputs = [&](int tile, float (&b)[KS0][CT]) {
    const int base = tile * TILE;
#pragma unroll
    for (int s = 0; s < KS0; ++s) {
      const int comp = 2 * s + h;
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        int row = base + 32 * c + j;
        row = row < a.R ? row : a.R - 1;
        float v;
        if (JAC) {
          const int role = row & 3;
          if (role == 0) v = comp < D ? a.states_in[(row >> 2) * D + comp] : (comp == D ? 1.f : 0.f);
          else v = (comp == role - 1) ? 1.f : 0.f;  // tangent e_{role-1}; role > D: zero column
        } else {
          v = comp < D ? a.states_in[static_cast<size_t>(row) * D + comp] : (comp == D ? 1.f : 0.f);
        }
        b[s][c] = v;
      }
    }
  };
  float bnext[KS0][CT];
  if (wave_global < ntiles) first_layer_inputs(wave_global, bnext);

  for (int tile = wave_global; tile < ntiles; tile += waves_total) {
    const int base = tile * TILE;
    // column -> row / trajectory bookkeeping for the CT columns this lane feeds
    int col_row[CT], col_traj[CT];
    bool col_primal[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      int row = base + 32 * c + j;
      row = row < a.R ? row : a.R - 1;
      col_row[c] = row;
      col_traj[c] = JAC ? (row >> 2) : (row / a.M);
      col_primal[c] = JAC ? ((row & 3) == 0) : true;
    }
    const bool primal = col_primal[0];  // same for every c: 32 is a multiple of 4

    // ---- encoder layer 0: relu(W0 [x; 1]) as (kW0Cols / 2) k-steps
    Act<CT> X, H;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) X.v[t][c][r] = 0.f;
    float bcur[KS0][CT];
#pragma unroll
    for (int s = 0; s < KS0; ++s)
#pragma unroll
      for (int c = 0; c < CT; ++c) bcur[s][c] = bnext[s][c];
    if (tile + waves_total < ntiles) first_layer_inputs(tile + waves_total, bnext);
#pragma unroll
    for (int s = 0; s < KS0; ++s) {
      const int comp = 2 * s + h;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const float w = lds[off_w0() + (32 * t + j) * kW0Cols + comp];
#pragma unroll
        for (int c = 0; c < CT; ++c)
          X.v[t][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(w, bcur[s][c], X.v[t][c], 0, 0, 0);
      }
    }
    if constexpr (!ROWPIPE) relu<CT, JAC>(X, primal);  // (the row-tile pipeline's prologue applies it)

    SplitAct<F16 ? CT : 0> SP;
    short2v amax = {0, 0};  // f16x3: largest hi halves handed to the MFMAs in this tile
    if constexpr (F16) {
      // a NaN / inf particle state: the first layer's exact-f32 MFMAs turn it into NaNs of either sign, and a
      // negative NaN would pass the ReLU as 0 -- report the input itself (a handful of compares per tile)
      bool bad = false;
#pragma unroll
      for (int s = 0; s < KS0; ++s)
#pragma unroll
        for (int c = 0; c < CT; ++c) bad |= !(fabsf(bcur[s][c]) <= 3.0e38f);
      if (bad) amax = short2v{0x7fff, 0x7fff};
    }
    if constexpr (PIPE) {
      constexpr int NL = 3 + 2 * NRES;  // 64x64 layers: encoder block, join, NRES trunk blocks
      FragPair frag;                    // weight fragments of the next MFMA group, in flight
      // V(C, l): everything half C needs before layer l's MFMAs -- the ReLU that ends layer
      // l - 1, the operand split of layer l's input, the initial value of its accumulator.
      // Layers 0, 2, 3, 5, .. read X-or-H alternately: even position in a block reads the block
      // input, odd position reads the hidden activation and accumulates onto the skip.
      auto vstage = [&](auto half, auto layer) {
        constexpr int C = decltype(half)::value;
        constexpr int l = decltype(layer)::value;
        const float* bl = lds + off_bias(NRES) + l * kUnits;
        if constexpr (l == 0) {            // X (ReLU'd by the encoder's first layer) -> H = b + W X
          split_half<C, false>(X, SP, neg_one, amax);
          bias_half<C, false>(bl, H, h);
        } else if constexpr (l == 1) {     // X += b + W relu(H)
          relu_half<C>(H);
          split_half<C, false>(H, SP, neg_one, amax);
          bias_half<C, true>(bl, X, h);
        } else if constexpr (l == 2) {     // join: H = traj_bias + W relu(X)
          relu_half<C>(X);
          split_half<C, false>(X, SP, neg_one, amax);
          const float* tb = a.traj_bias + static_cast<size_t>(col_traj[C]) * kUnits + 4 * h;
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const f32x4 b = *reinterpret_cast<const f32x4*>(tb + 32 * t + 8 * g);
#pragma unroll
              for (int e = 0; e < 4; ++e) H.v[t][C][4 * g + e] = b[e];
            }
        } else if constexpr ((l - 3) % 2 == 0) {  // trunk block, first layer: X = b + W H
          constexpr bool kSigned = (l == 3 && KIND != kMeasure);  // no ReLU after the join (dynamics)
          // l == 3 consumes the per-trajectory term: its ReLU keeps a NaN / inf for the split to report (relu_sat)
          if constexpr (kSigned) saturate_half<C>(H);
          else relu_half<C, (l != 3)>(H);
          split_half<C, kSigned>(H, SP, neg_one, amax);
          bias_half<C, false>(bl, X, h);
        } else {                                  // trunk block, second layer: H += b + W relu(X)
          relu_half<C>(X);
          split_half<C, false>(X, SP, neg_one, amax);
          bias_half<C, true>(bl, H, h);
        }
      };
      auto mstage = [&](auto half, auto layer) {
        constexpr int C = decltype(half)::value;
        constexpr int l = decltype(layer)::value;
        constexpr bool to_h = (l == 0 || l == 2 || (l > 2 && (l - 3) % 2 == 1));
        // the stage after (C, l) is (1, l) for C == 0 and (0, l + 1) for C == 1
        constexpr int ln = (C == 0) ? l : (l + 1 < NL ? l + 1 : l);
        mfma_half<C>(lds + off_layers() + l * kLayerFloats, lds + off_layers() + ln * kLayerFloats, frag, SP,
                     to_h ? H : X, lane);
      };
      using I0 = std::integral_constant<int, 0>;
      using I1 = std::integral_constant<int, 1>;
      asm volatile("" ::: "memory");  // keep the LDS fragment reads inside the tile loop (see mfma_layer)
      frag = load_frag(lds + off_layers(), lane, 0);
      vstage(I0{}, I0{});
      __builtin_amdgcn_sched_barrier(0);
      static_for<NL>([&](auto layer) {
        constexpr int l = decltype(layer)::value;
        mstage(I0{}, layer);
        vstage(I1{}, layer);
        pin_mfma_valu_interleave<5>();
        __builtin_amdgcn_sched_barrier(0);
        mstage(I1{}, layer);
        if constexpr (l + 1 < NL) vstage(I0{}, std::integral_constant<int, l + 1>{});
        else relu_half<0>(H);
        pin_mfma_valu_interleave<5>();
        __builtin_amdgcn_sched_barrier(0);
      });
      relu_half<1>(H);
    } else if constexpr (ROWPIPE) {
      // column tiles in lock step (one LDS read per weight fragment for all of them), pipelined on output ROW tiles
      rowpipe_net_f16<CT, NRES, KIND>(lds, X, H, SP, [&](Act<CT>& acc) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int c = 0; c < CT; ++c) {
            const float* tb = a.traj_bias + static_cast<size_t>(col_traj[c]) * kUnits + 32 * t + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const f32x4 b = *reinterpret_cast<const f32x4*>(tb + 8 * g);
#pragma unroll
              for (int e = 0; e < 4; ++e) acc.v[t][c][4 * g + e] = b[e];
            }
          }
      }, lane, neg_one, amax);
    } else {
    // ---- encoder residual block (layers 0, 1)
    if constexpr (F16) res_block_f16<CT, false, JAC>(lds, NRES, 0, X, H, SP, lane, neg_one, amax, primal);
    else res_block<CT, JAC>(lds, NRES, 0, X, H, lane, primal);

    // ---- join layer (2): per-trajectory hoisted half arrives as the accumulator init
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const float* tb = a.traj_bias + static_cast<size_t>(col_traj[c]) * kUnits + 32 * t + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 b = *reinterpret_cast<const f32x4*>(tb + 8 * g);
#pragma unroll
          for (int e = 0; e < 4; ++e) H.v[t][c][4 * g + e] = (JAC && !primal) ? 0.f : b[e];
        }
      }
    if constexpr (F16) {
      split_act<CT, JAC>(X, SP, neg_one, amax);
      mfma_layer_f16<CT>(lds + off_layers() + 2 * kLayerFloats, SP, H, lane);
    } else {
      mfma_layer<CT>(lds + off_layers() + 2 * kLayerFloats, X, H, lane);
    }
    if (KIND == kMeasure) {
      if constexpr (F16) {  // consumes the per-trajectory term: keep a NaN of either sign for the next split to report
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) H.v[t][c][r] = relu_keepnan(H.v[t][c][r]);
      } else {
        relu<CT, JAC>(H, primal);
      }
    }

    // ---- residual trunk: activations now live in H, X is scratch
#pragma unroll
    for (int i = 0; i < NRES; ++i) {
      // without a ReLU after the join layer (dynamics) the trunk's first split sees signed values
      if constexpr (F16) {
        if (i == 0 && KIND != kMeasure) res_block_f16<CT, true, JAC>(lds, NRES, 3, H, X, SP, lane, neg_one, amax, primal);
        else res_block_f16<CT, false, JAC>(lds, NRES, 3 + 2 * i, H, X, SP, lane, neg_one, amax, primal);
      }
      else res_block<CT, JAC>(lds, NRES, 3 + 2 * i, H, X, lane, primal);
    }

    }
    if constexpr (F16) {
      // an operand beyond the f16 range cannot be split exactly: inner activations saturate at 65504 (relu_sat,
      // hi = 0x7BFF), a non-finite state or per-trajectory term arrives as hi = inf / NaN (>= 0x7C00) and is
      // clamped by the next relu_sat -- either way the tile's outputs stay finite and the flag says they are
      // invalid (engine.check_range raises: per forward_loop, and per step for a bare forward())
      if (a.range_flag != nullptr && (amax[0] >= kF16Saturated || amax[1] >= kF16Saturated))
        atomicOr(a.range_flag, 1);
    }

    // ---- head (64 -> NOUT) on the VALU: each lane holds 32 of the 64 features of its columns
    float out[NOUT][CT];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
      float part[CT];
#pragma unroll
      for (int c = 0; c < CT; ++c) part[c] = 0.f;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 w = *reinterpret_cast<const f32x4*>(
              lds + off_whead(NRES) + o * kUnits + 32 * t + 8 * g + 4 * h);
#pragma unroll
          for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) part[c] = __builtin_fmaf(w[e], H.v[t][c][4 * g + e], part[c]);  // explicit: the strict mode's chain
        }
#pragma unroll
      for (int c = 0; c < CT; ++c) out[o][c] = part[c] + __shfl_xor(part[c], 32);
    }

    // ---- epilogue: lane l finalises column l of the tile (CT == 2) or column j (CT == 1, h == 0)
    float mine[NOUT];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
      if (CT == 2) {
        // a plain `h ? out[o][1] : out[o][0]` is canonicalised into a dynamically indexed array,
        // i.e. a round trip through scratch (12-48 B/lane); pin both values in VGPRs first
        float lo_col = out[o][0], hi_col = out[o][CT - 1];
        asm volatile("" : "+v"(lo_col), "+v"(hi_col));
        mine[o] = h ? hi_col : lo_col;
      } else {
        mine[o] = out[o][0];
      }
    }
    const int my_row = base + (CT == 2 ? lane : j);
    const bool active = my_row < a.R && (CT == 2 || h == 0);
    const float* bh = lds + off_bhead(NRES);

    if (KIND == kMeasure) {
      if (active) {
        const int traj = my_row / a.M;
        float ll = mine[0] + bh[0];
        if (a.mod_logw) ll += a.mod_logw[static_cast<size_t>(traj) * a.logw_stride];
        if (a.combine) {
          const float prev = a.loglik[my_row];
          if constexpr (PREC == MMF_PREC_F32) {
            // exact-fp32 mode = the bit-reproducible mode: shared deterministic exp / log (mmf_detmath.h)
            ll = mmf_det_logaddexp(prev, ll);
          } else {
            const float m = fmaxf(prev, ll);
            ll = (m == -INFINITY) ? m : m + logf(expf(prev - m) + expf(ll - m));
          }
        }
        a.loglik[my_row] = ll;
      }
    } else if (KIND == kDynamics) {
      if (active) {
        const float gate = mine[D] + bh[D];
        float sg;
        if constexpr (PREC == MMF_PREC_F32) sg = mmf_det_sigmoid(gate);
        else sg = 1.0f / (1.0f + expf(-gate));
        float xo[D], eps[D];
        const bool noisy = a.noise != nullptr || a.noise_mode == 2;
        if (a.noise_mode == 2) {
          // counter-based noise: a pure function of (seed, step, trajectory, particle) -- nothing is read
          float z[4];
          const unsigned traj = static_cast<unsigned>(my_row / a.M);
          mmf_philox_normal4(a.noise_seed, a.noise_step, a.noise_traj0 + traj, static_cast<unsigned>(my_row) - traj * a.M, z);
#pragma unroll
          for (int i = 0; i < D; ++i) eps[i] = z[i];
        }
#pragma unroll
        for (int i = 0; i < D; ++i) {
          xo[i] = a.states_in[static_cast<size_t>(my_row) * D + i];
          if (a.noise_mode != 2) eps[i] = a.noise ? a.noise[static_cast<size_t>(my_row) * D + i] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < D; ++i) {
          float v = __builtin_fmaf(mine[i] + bh[i], sg, xo[i]);
          if (noisy) {
#pragma unroll
            for (int k = 0; k < D; ++k) v = __builtin_fmaf(a.scale_tril[i * D + k], eps[k], v);
          }
          a.states_out[static_cast<size_t>(my_row) * D + i] = v;
        }
      }
    } else {  // jacobian: primal column -> x', tangent column c -> d x' / d x_c
      const int role = my_row & 3;
      const int traj = my_row >> 2;
      float dirp[D], gatep;
#pragma unroll
      for (int i = 0; i < D; ++i) dirp[i] = quad_first(mine[i] + bh[i]);
      gatep = quad_first(mine[D] + bh[D]);
      const float sg = 1.0f / (1.0f + expf(-gatep));
      if (active) {
        if (role == 0) {
#pragma unroll
          for (int i = 0; i < D; ++i)
            a.states_out[traj * D + i] = a.states_in[traj * D + i] + dirp[i] * sg;
        } else if (role <= D) {
          const float dgate = mine[D];  // tangent columns carry no bias
#pragma unroll
          for (int i = 0; i < D; ++i) {
            const float dv = mine[i] * sg + dirp[i] * (sg * (1.0f - sg)) * dgate + ((i == role - 1) ? 1.f : 0.f);
            a.jac[(static_cast<size_t>(traj) * D + i) * D + (role - 1)] = dv;
          }
        }
      }
    }
  }
  }  // pass
}

template <int D, int NRES, int KIND, int PREC, int CT, int WPS, bool PIPE = false, bool ROWPIPE = false>
int launch_variant(const NetArgsMulti& m, int count, hipStream_t s) {
  const NetArgs& a = m.a[0];
  const size_t lds = static_cast<size_t>(blob_floats(NRES)) * sizeof(float);
  constexpr int waves = WPS * 4, tile = 32 * CT;
  const int ntiles = (a.R + tile - 1) / tile;
  int grid = (ntiles + waves - 1) / waves;
  if (grid > 256) grid = 256;
  if (grid < 1) grid = 1;
  auto k = particle_net_kernel<D, NRES, KIND, CT, PREC, WPS, PIPE, ROWPIPE>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
  if (e != hipSuccess) return static_cast<int>(e);
  k<<<dim3(grid, count), WPS * 256, lds, s>>>(m);
  MMF_CHECK_LAUNCH();
  return 0;
}

template <int D, int NRES, int KIND, int PREC>
int launch_ct(const NetArgsMulti& m, int count, hipStream_t s) {
  const NetArgs& a = m.a[0];
  // small problems: 32-particle tiles spread over more waves; large: 64-particle tiles
  const bool big = a.R >= 256 * 8 * 64;
  static const int variant = [] { const char* v = getenv("MMF_K2_VARIANT"); return v ? atoi(v) : 0; }();
  if constexpr (PREC == MMF_PREC_F16X3 && KIND != kJacobian) {
    if (big && variant == 1) return launch_variant<D, NRES, KIND, PREC, 1, 3>(m, count, s);  // 32-particle tiles, 3 waves/SIMD
    if (big && variant == 2) return launch_variant<D, NRES, KIND, PREC, 1, 2>(m, count, s);
    if (big && variant == 3) return launch_variant<D, NRES, KIND, PREC, 2, 2, false>(m, count, s);  // unpipelined
    if (big && variant == 4) return launch_variant<D, NRES, KIND, PREC, 2, 2, false, true>(m, count, s);  // row-tile pipeline: half the LDS fragment reads
    if (!big && variant == 5) return launch_variant<D, NRES, KIND, PREC, 1, 2, false, true>(m, count, s);  // small problems, row-tile pipeline
    if (big) return launch_variant<D, NRES, KIND, PREC, 2, 2, true>(m, count, s);
  }
  if (big) return launch_variant<D, NRES, KIND, PREC, 2, 2>(m, count, s);
  return launch_variant<D, NRES, KIND, PREC, 1, 2>(m, count, s);
}

template <int KIND>
int launch_multi(const NetArgsMulti& m, int count, int d, int n_res, int precision, hipStream_t s);

template <int KIND>
int launch(const NetArgs& a, int d, int n_res, int precision, hipStream_t s) {
  NetArgsMulti m{};
  m.a[0] = a;
  return launch_multi<KIND>(m, 1, d, n_res, precision, s);
}

template <int KIND>
int launch_multi(const NetArgsMulti& m, int count, int d, int n_res, int precision, hipStream_t s) {
#define MMF_CASE(D, NR)                                                              \
  if (d == D && n_res == NR) {                                                       \
    if (precision == MMF_PREC_F32) return launch_ct<D, NR, KIND, MMF_PREC_F32>(m, count, s); \
    if (precision == MMF_PREC_F16X3) return launch_ct<D, NR, KIND, MMF_PREC_F16X3>(m, count, s); \
    return MMF_EINVAL;                                                               \
  }
  if (KIND == kJacobian || KIND == kDynamics) {
    MMF_CASE(2, 3) MMF_CASE(3, 3)
    return MMF_EINVAL;
  }
  MMF_CASE(2, 2) MMF_CASE(3, 2)
#undef MMF_CASE
  return MMF_EINVAL;
}

}  // namespace

extern "C" size_t mmf_particle_net_floats(int n_res) {
  if (n_res < 0 || n_res > MMF_MAX_RES) return 0;
  return static_cast<size_t>(blob_floats(n_res));
}

extern "C" int mmf_pack_particle_net(const MmfParticleNetDesc* d, float* packed, int precision,
                                     void* stream) {
  if (!d || !packed) return MMF_EINVAL;
  if (precision != MMF_PREC_F32 && precision != MMF_PREC_F16X3) return MMF_EINVAL;
  if (d->d_in < 1 || d->d_in > MMF_MAX_STATE_DIM || d->n_res < 0 || d->n_res > MMF_MAX_RES) return MMF_EINVAL;
  if (d->n_out < 1 || d->n_out > kHeadRows) return MMF_EINVAL;
  if (d->join_state_off < 0 || d->join_state_off + kUnits > d->join_in) return MMF_EINVAL;
  if (!d->w_in || !d->b_in || !d->w_join || !d->w_head || !d->b_head) return MMF_EINVAL;
  for (int i = 0; i < 2; ++i)
    if (!d->w_enc[i] || !d->b_enc[i]) return MMF_EINVAL;
  for (int i = 0; i < 2 * d->n_res; ++i)
    if (!d->w_res[i] || !d->b_res[i]) return MMF_EINVAL;
  const int total = blob_floats(d->n_res);
  pack_particle_net_kernel<<<(total + 255) / 256, 256, 0, static_cast<hipStream_t>(stream)>>>(*d, packed, precision);
  MMF_CHECK_LAUNCH();
  return 0;
}

extern "C" int mmf_pf_dynamics(const float* packed, int n_res, int precision, const float* states_in,
                               const float* traj_bias, const float* noise, const float* scale_tril,
                               float* states_out, int* range_flag, int N, int M, int d, void* stream) {
  if (!packed || !states_in || !traj_bias || !states_out || (noise && !scale_tril)) return MMF_EINVAL;
  if (N < 0 || M < 1) return MMF_EINVAL;
  if (static_cast<long long>(N) * M > 0x7fffffffLL / 8) return MMF_ETOOLARGE;
  if (N == 0) return 0;
  NetArgs a{};
  a.packed = packed; a.states_in = states_in; a.traj_bias = traj_bias; a.noise = noise;
  a.scale_tril = scale_tril; a.states_out = states_out; a.R = N * M; a.M = M;
  a.range_flag = range_flag;
  return launch<kDynamics>(a, d, n_res, precision, static_cast<hipStream_t>(stream));
}

extern "C" int mmf_pf_dynamics_philox(const float* packed, int n_res, int precision, const float* states_in,
                                      const float* traj_bias, unsigned long long seed, unsigned step, unsigned traj0,
                                      const float* scale_tril, float* states_out, int* range_flag, int N, int M, int d,
                                      void* stream) {
  if (!packed || !states_in || !traj_bias || !states_out || !scale_tril) return MMF_EINVAL;
  if (N < 0 || M < 1) return MMF_EINVAL;
  if (static_cast<long long>(N) * M > 0x7fffffffLL / 8) return MMF_ETOOLARGE;
  if (N == 0) return 0;
  NetArgs a{};
  a.packed = packed; a.states_in = states_in; a.traj_bias = traj_bias; a.noise = nullptr;
  a.scale_tril = scale_tril; a.states_out = states_out; a.R = N * M; a.M = M;
  a.range_flag = range_flag;
  a.noise_seed = seed; a.noise_step = step; a.noise_traj0 = traj0; a.noise_mode = 2;
  return launch<kDynamics>(a, d, n_res, precision, static_cast<hipStream_t>(stream));
}

namespace {
__global__ void philox_normals_kernel(unsigned long long seed, unsigned step, unsigned traj0, int M, int d, size_t R,
                                      float* __restrict__ out) {
  const size_t r = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (r >= R) return;
  float z[4];
  const unsigned traj = static_cast<unsigned>(r / M);
  mmf_philox_normal4(seed, step, traj0 + traj, static_cast<unsigned>(r - static_cast<size_t>(traj) * M), z);
  for (int i = 0; i < d; ++i) out[r * d + i] = z[i];
}
__global__ void philox_uniforms_kernel(unsigned long long seed, unsigned step0, unsigned traj0, int T, int N,
                                       float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= T * N) return;
  out[i] = mmf_philox_uniform(seed, step0 + static_cast<unsigned>(i / N), traj0 + static_cast<unsigned>(i % N));
}
}  // namespace

extern "C" int mmf_philox_normals(unsigned long long seed, unsigned step, unsigned traj0, float* out, int N, int M,
                                  int d, void* stream) {
  if (!out || N < 0 || M < 1 || d < 1 || d > 4) return MMF_EINVAL;
  const size_t R = static_cast<size_t>(N) * M;
  if (R == 0) return 0;
  philox_normals_kernel<<<static_cast<unsigned>((R + 255) / 256), 256, 0, static_cast<hipStream_t>(stream)>>>(
      seed, step, traj0, M, d, R, out);
  MMF_CHECK_LAUNCH();
  return 0;
}

extern "C" int mmf_philox_uniforms(unsigned long long seed, unsigned step0, unsigned traj0, float* out, int T, int N,
                                   void* stream) {
  if (!out || T < 0 || N < 0) return MMF_EINVAL;
  if (T * N == 0) return 0;
  philox_uniforms_kernel<<<(T * N + 255) / 256, 256, 0, static_cast<hipStream_t>(stream)>>>(seed, step0, traj0, T, N, out);
  MMF_CHECK_LAUNCH();
  return 0;
}

extern "C" int mmf_pf_measure(const float* packed, int n_res, int precision, const float* states,
                              const float* traj_bias, const float* modality_logw, int logw_stride,
                              float* loglik, int combine, int* range_flag, int N, int M, int d,
                              void* stream) {
  if (!packed || !states || !traj_bias || !loglik) return MMF_EINVAL;
  if (N < 0 || M < 1) return MMF_EINVAL;
  if (static_cast<long long>(N) * M > 0x7fffffffLL / 8) return MMF_ETOOLARGE;
  if (N == 0) return 0;
  NetArgs a{};
  a.packed = packed; a.states_in = states; a.traj_bias = traj_bias; a.mod_logw = modality_logw;
  a.logw_stride = logw_stride; a.loglik = loglik; a.combine = combine; a.R = N * M; a.M = M;
  a.range_flag = range_flag;
  return launch<kMeasure>(a, d, n_res, precision, static_cast<hipStream_t>(stream));
}

extern "C" int mmf_pf_measure_seq(const float* const* packed, int n_res, int precision, const float* states,
                                  const float* const* traj_bias, const float* const* modality_logw, int logw_stride,
                                  float* loglik, int* range_flag, int K, int N, int M, int d, void* stream) {
  if (!packed || !states || !traj_bias || !loglik) return MMF_EINVAL;
  if (K < 1 || K > MMF_LOOP_MAX_MEAS || N < 0 || M < 1) return MMF_EINVAL;
  if (static_cast<long long>(N) * M > 0x7fffffffLL / 8) return MMF_ETOOLARGE;
  if (N == 0) return 0;
  NetArgsMulti m{};
  for (int k = 0; k < K; ++k) {
    if (!packed[k] || !traj_bias[k]) return MMF_EINVAL;
    NetArgs& a = m.a[k];
    a.packed = packed[k]; a.states_in = states; a.traj_bias = traj_bias[k];
    a.mod_logw = modality_logw ? modality_logw[k] : nullptr;
    a.logw_stride = logw_stride; a.loglik = loglik; a.combine = k > 0; a.R = N * M; a.M = M;
    a.range_flag = range_flag;
  }
  m.seq = K;
  return launch_multi<kMeasure>(m, 1, d, n_res, precision, static_cast<hipStream_t>(stream));
}

extern "C" int mmf_dynamics_jacobian(const float* packed, int n_res, int precision, const float* states_in,
                                     const float* traj_bias, float* states_out, float* jac,
                                     int* range_flag, int N, int d, void* stream) {
  if (!packed || !states_in || !traj_bias || !states_out || !jac) return MMF_EINVAL;
  if (N < 0) return MMF_EINVAL;
  if (N > 0x7fffffff / 32) return MMF_ETOOLARGE;
  if (N == 0) return 0;
  NetArgs a{};
  a.packed = packed; a.states_in = states_in; a.traj_bias = traj_bias;
  a.states_out = states_out; a.jac = jac; a.R = 4 * N; a.M = 4;
  a.range_flag = range_flag;
  return launch<kJacobian>(a, d, n_res, precision, static_cast<hipStream_t>(stream));
}

extern "C" int mmf_dynamics_jacobian_multi(const float* const* packed, int n_res, int precision,
                                           const float* states_in, const float* const* traj_bias,
                                           float* states_out, float* jac, int* range_flag, int K, int N,
                                           int d, void* stream) {
  if (!packed || !states_in || !traj_bias || !states_out || !jac) return MMF_EINVAL;
  if (K < 1 || K > MMF_LOOP_MAX_MEAS || N < 0) return MMF_EINVAL;
  if (N > 0x7fffffff / 32) return MMF_ETOOLARGE;
  if (N == 0) return 0;
  NetArgsMulti m{};
  for (int k = 0; k < K; ++k) {
    if (!packed[k] || !traj_bias[k]) return MMF_EINVAL;
    NetArgs& a = m.a[k];
    a.packed = packed[k]; a.traj_bias = traj_bias[k];
    a.states_in = states_in + static_cast<size_t>(k) * N * d;
    a.states_out = states_out + static_cast<size_t>(k) * N * d;
    a.jac = jac + static_cast<size_t>(k) * N * d * d;
    a.R = 4 * N; a.M = 4;
    a.range_flag = range_flag;
  }
  return launch_multi<kJacobian>(m, K, d, n_res, precision, static_cast<hipStream_t>(stream));
}

#include "particle_net_train.inc"
#include "pf_persistent.inc"
