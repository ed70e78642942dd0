// K6, fused (round 5): recompute + backward data path + weight gradients of one per-particle network in ONE kernel.
// Its own translation unit; shares the tile machinery (particle_net_tiles.h) and the K6 row stores
// (particle_net_train_common.h) with particle_net.hip.
//
// Replaces, for the native training recursion (pf_train_loop.hip), the three passes of round 4 --
// particle_net_train_fwd_kernel (recompute, writes every layer input), particle_net_train_bwd_kernel (writes every
// pre-activation gradient) and weight_grad_h_kernel (re-reads both) -- whose f16 buffers were ~2.1 KB of HBM traffic
// per particle and network call against 44 B of algorithmic input, 60 % of config 5's training step
// (/root/reference/crossmodal/train_helpers.py:124-162 -> torchfilter.train.train_filter over
// door_models/dynamics.py:102-134 and door_models/pf.py:63-107).  Here a layer's input a_l and gradient dz_l never
// leave the chip:
//
//  * ONE copy of the weights in LDS serves both directions: the MMF_PREC_F16X3_DUAL image (dual_off, particle_net.hip)
//    is read by rows for the forward product (ds_read_b128) and through ds_read_b64_tr_b16 for the transposed
//    product of the backward; 7 layers x 16 KB + 32 KB of exchange slots fit the 160 KB of a CU.
//  * the recompute runs the forward pass's own f16x3 arithmetic; the hi halves of every layer's operand split ARE the
//    f16 copy of a_l the weight gradient needs (16 VGPRs per layer), and a ReLU's mask is "that half is non-zero".
//  * dW_l = sum_p dz_l[p] a_l[p]^T contracts over PARTICLES, which are the lane index of every tile of the chain, so
//    both operands need a transpose: each wave parks its 32-particle tile of (dz_l, a_l) as f16 in an LDS slot
//    ([particle][feature], 8-byte units XOR-swizzled), and reads all four waves' slots back transposed
//    (ds_read_b64_tr_b16) as the A / B operands of v_mfma_f32_32x32x16_f16 -- f16 x f16 products are exact in the fp32
//    accumulator.  Wave w owns quadrant (w >> 1, w & 1) of every layer's 64 x 64 gradient: 16 accumulator registers per
//    layer, resident across all tiles of the launch; one read-modify-write of the workgroup's partial at the end
//    (pw (NL, slots, 64, 64): no atomics, fixed summation order).
//  * gradients have no fixed range, f16 does: dz is exchanged relative to a RUNNING exponent per layer (the largest row
//    magnitude any group of the launch has shown so far, brought to [2^14, 2^15)); a group with a larger one first
//    rescales the layer's accumulators by the exact power of two (rare), smaller ones are stored relative to it (their
//    error is 2^-25 of the largest, as in fixed point).  The backward already scales every row by the power of two
//    c_p that brings it to [2^7, 2^8) before its own operand split, so the exchanged value is that split's hi half
//    times 2^(7 - (E - e_row)) -- one v_pk_mul_f16 per pair -- and the MFMAs accumulate straight into the resident
//    accumulators (no per-tile partial product, no fp32 fix-up); the stored partial is accW 2^(E - 141).
//
// The dynamics network (9 layers = 144 KB) does not leave room for the exchange slots: it runs as three launches --
// encoder forward (first layer + block 0: writes its 64 outputs per particle, fp32), trunk (join + 3 blocks + head +
// the sigmoid-gate epilogue, forward and backward; writes the gradient w.r.t. the encoder output) and encoder
// backward -- 1 KB of traffic per particle instead of 5.8.
//
// What still travels through HBM in the round-4 compact format: dz of the first layer and of the join layer and the
// head's input (3 x 128 B per particle), for the narrow reductions of small_grads_h_kernel.

#include "particle_net_train_common.h"

namespace {

typedef short v4i16 __attribute__((__vector_size__(8)));
typedef short v8i16 __attribute__((__vector_size__(16)));
using LdsV4Ptr = __attribute__((address_space(3))) v4i16*;

// two ds_read_b64_tr_b16 -> one 8-element MFMA operand (elements 0..3 from `p0`, 4..7 from `p1`)
__device__ __forceinline__ half8 lds_tr_pair(const unsigned char* p0, const unsigned char* p1) {
  const v4i16 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LdsV4Ptr)(p0));
  const v4i16 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LdsV4Ptr)(p1));
  const v8i16 c = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(half8, c);
}

// exchange image [32 particles][64 features] f16 (128-B rows): 8-byte unit u of particle p at 128 p + 8 (u ^ swizzle(p))
__host__ __device__ constexpr int xchg_swizzle(int p) { return (p & 5) | ((p & 2) << 2) | ((p & 8) >> 2); }

enum FusedPart { kFull = 0, kTrunk = 1, kEnc = 2, kEncFwd = 3 };

template <int NRES, int PART>
struct FusedLayout {
  static constexpr int NL = num_layers(NRES);
  static constexpr int L0 = PART == kTrunk ? 2 : 0;
  static constexpr int L1 = (PART == kEnc || PART == kEncFwd) ? 2 : NL;
  static constexpr int NLAY = L1 - L0;
  static constexpr int kImgBytes = NLAY * kLayerFloats * 4;
  // small section: W0 (64 x 8) | biases (NL x 64) | head weights (4 x 64) | head bias (8)
  static constexpr int kTailFloats = blob_floats(NRES) - off_bias(NRES);
  static constexpr int kSmallFloats = off_layers() + kTailFloats;
  static constexpr int kSmallOff = kImgBytes;
  static constexpr int kXchgOff = kSmallOff + ((kSmallFloats * 4 + 255) / 256) * 256;
  static constexpr int kSlotBytes = 8192;  // dz image (4 KB) | a image (4 KB)
  static constexpr int kScaleOff = kXchgOff + (PART == kEncFwd ? 0 : 4 * kSlotBytes);
  static constexpr int kBytes = kScaleOff + 64;
};

struct FusedArgs {
  const float* blob;       // MMF_PREC_F16X3_DUAL
  const float* states;     // (R, D): parts with the first layer
  const float* traj_bias;  // (N, 64): parts with the join layer
  const float* act_in;     // kTrunk: (R, 64) fp32 encoder output
  float* act_out;          // kEncFwd
  const float* d_out;      // measurement: (R) dL / d log-likelihood
  const float* g_next;     // dynamics trunk: (R, D) dL / d x'
  float* d_raw;            // dynamics trunk out: (R, D + 1) dL / d (dir, gate)
  const float* g_in;       // kEnc: (R, 64) fp32 dL / d encoder output
  float* g_out;            // kTrunk out
  float* d_states;         // (R, D): through the network only
  _Float16* dz_first_h;    // compact slots for small_grads_h_kernel
  float* sc_first;
  _Float16* dz_join_h;
  float* sc_join;
  _Float16* h_last_h;
  float* pw;               // (NL, slots, 64, 64), accumulated in place: slot blockIdx.x
  float* pb;               // (NL, slots, 64)
  int R, M, slots;
};

__device__ __forceinline__ void split_pair_plain(float x0, float x1, float neg_one, unsigned& hi, unsigned& lo) {
  const half2v hh = __builtin_convertvector(f32x2v{x0, x1}, half2v);
  const float r0 = __builtin_fmaf(static_cast<float>(hh[0]), neg_one, x0);
  const float r1 = __builtin_fmaf(static_cast<float>(hh[1]), neg_one, x1);
  const half2v ll = __builtin_convertvector(f32x2v{r0, r1}, half2v);
  hi = __builtin_bit_cast(unsigned, hh);
  lo = __builtin_bit_cast(unsigned, ll);
}

// split_act without the range tracking (the forward pass of the step checked these very values)
__device__ __forceinline__ void split_act_nr(const Act<1>& x, SplitAct<1>& o, float neg_one) {
#pragma unroll
  for (int tp = 0; tp < 2; ++tp)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      u32x4 hh, ll;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        unsigned a, b;
        split_pair_plain(x.v[tp][0][8 * u + 2 * p], x.v[tp][0][8 * u + 2 * p + 1], neg_one, a, b);
        hh[p] = a;
        ll[p] = b;
      }
      o.hi[2 * tp + u][0] = __builtin_bit_cast(half8, hh);
      o.lo[2 * tp + u][0] = __builtin_bit_cast(half8, ll);
    }
}

// g *= [a > 0] with a given as the f16 hi halves of the operand split that consumed it (element (t, r) <-> half
// (r & 7) of fragment 2 t + (r >> 3)); an activation below 2^-25 counts as 0
__device__ __forceinline__ void mask_by_halves(const half8 (&s)[4], Act<1>& g) {
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const unsigned short bits = __builtin_bit_cast(v8i16, s[2 * t + (r >> 3)])[r & 7];
      g.v[t][0][r] = (bits & 0x7fffu) ? g.v[t][0][r] : 0.f;
    }
}
__device__ __forceinline__ void mask_by_act(const Act<1>& a, Act<1>& g) {
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) g.v[t][0][r] = a.v[t][0][r] > 0.f ? g.v[t][0][r] : 0.f;
}

__device__ __forceinline__ void load_rows_f32(const float* __restrict__ base, Act<1>& a, int row, int h, float scale) {
  const float* p = base + static_cast<size_t>(row) * kUnits + 4 * h;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(p + 32 * t + 8 * g);
#pragma unroll
      for (int e = 0; e < 4; ++e) a.v[t][0][4 * g + e] = v[e] * scale;
    }
}

template <int D, int NRES, int KIND, int PART>
__global__ __launch_bounds__(256, 1) void particle_net_train_fused_kernel(FusedArgs a) {
  using LY = FusedLayout<NRES, PART>;
  constexpr int NL = LY::NL, L0 = LY::L0, NLAY = LY::NLAY;
  constexpr bool FIRST = PART != kTrunk, HEAD = PART == kFull || PART == kTrunk, BWD = PART != kEncFwd;
  constexpr bool JOIN = HEAD;
  constexpr int NOUT = (KIND == kMeasure) ? 1 : D + 1;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  unsigned char* ldsb = reinterpret_cast<unsigned char*>(lds);
  float* small = lds + LY::kSmallOff / 4;
  const float* sW0 = small;
  const float* sBias = small + off_layers();
  const float* sWhead = sBias + NL * kUnits;
  const float* sBhead = sWhead + kHeadRows * kUnits;
  float* tmax = lds + LY::kScaleOff / 4;  // the four waves' tile maxima of the layer being exchanged
  {
    const float4* src = reinterpret_cast<const float4*>(a.blob + off_layers() + L0 * kLayerFloats);
    mmf::stage_to_lds<NLAY * kLayerFloats / 4, 256, 4>(src, reinterpret_cast<float4*>(lds), threadIdx.x);
    for (int i = threadIdx.x; i < off_layers(); i += 256) small[i] = a.blob[i];
    for (int i = threadIdx.x; i < LY::kTailFloats; i += 256) small[off_layers() + i] = a.blob[off_bias(NRES) + i];
  }
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
  const int q = (lane >> 2) & 3, p4 = lane & 3, g1 = (lane >> 4) & 1;
  const int mt = wave >> 1, nt = wave & 1;
  // per-lane LDS offsets (closed forms checked against the layouts' definitions: scripts/debug/lds_bank_check.py)
  const int base_r = 256 * j + 16 * (h ^ dual_swizzle(j));                                              // weight row reads
  const int base_t = 256 * (4 * h + q) + 8 * (p4 >> 1) + 16 * ((2 * g1 + (p4 & 1)) ^ h ^ (4 * q));      // weight transposed reads
  const int wx = 128 * j + 8 * (xchg_swizzle(j) ^ h);                                                    // exchange writes
  const int base_rx = 128 * (8 * h + q) + 8 * ((4 * g1 + p4) ^ ((q & 1) | (h << 1) | ((q >> 1) << 3)));  // exchange transposed reads
  const int rxa = base_rx ^ (64 * mt), rxb = base_rx ^ (64 * nt);

  float neg_one = -1.0f;  // in an SGPR, opaque to the optimiser (split_pair)
  asm volatile("" : "+s"(neg_one));

  // acc += W_l in  (rows of the image: the forward product)
  auto layer_fwd = [&](int li, const SplitAct<1>& sp, Act<1>& acc) {
    asm volatile("" ::: "memory");  // see mfma_layer: keep LICM from hoisting the fragment reads
    const unsigned char* img = ldsb + li * (kLayerFloats * 4);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const half8 ahi = *reinterpret_cast<const half8*>(img + ((base_r ^ (32 * s)) + 8192 * t));
        const half8 alo = *reinterpret_cast<const half8*>(img + ((base_r ^ (128 + 32 * s)) + 8192 * t));
        acc.v[t][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, sp.hi[s][0], acc.v[t][0], 0, 0, 0);
        acc.v[t][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, sp.lo[s][0], acc.v[t][0], 0, 0, 0);
        acc.v[t][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo, sp.hi[s][0], acc.v[t][0], 0, 0, 0);
      }
  };
  // acc += W_l^T in  (columns of the same image through the transposing read: the backward product)
  auto layer_bwd = [&](int li, const SplitAct<1>& sp, Act<1>& acc) {
    asm volatile("" ::: "memory");
    const unsigned char* img = ldsb + li * (kLayerFloats * 4);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const half8 ahi = lds_tr_pair(img + ((base_t ^ (64 * t)) + 4096 * s), img + ((base_t ^ (64 * t + 32)) + 2048 + 4096 * s));
        const half8 alo = lds_tr_pair(img + ((base_t ^ (128 + 64 * t)) + 4096 * s), img + ((base_t ^ (128 + 64 * t + 32)) + 2048 + 4096 * s));
        acc.v[t][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, sp.hi[s][0], acc.v[t][0], 0, 0, 0);
        acc.v[t][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, sp.lo[s][0], acc.v[t][0], 0, 0, 0);
        acc.v[t][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo, sp.hi[s][0], acc.v[t][0], 0, 0, 0);
      }
  };

  // this wave's quadrant of every layer's weight gradient, and (nt == 0) its half of the bias gradient
  f32x16 accW[BWD ? NLAY : 1];
  float accB[BWD ? NLAY : 1];
  int Eacc[BWD ? NLAY : 1];  // exponent field the accumulators are relative to: accW = dW 2^(141 - Eacc); 0 = nothing summed yet
  if constexpr (BWD) {
#pragma unroll
    for (int l = 0; l < NLAY; ++l) {
#pragma unroll
      for (int r = 0; r < 16; ++r) accW[l][r] = 0.f;
      accB[l] = 0.f;
      Eacc[l] = 0;
    }
  }

  const int ntiles = (a.R + 31) / 32;
  const int ngroups = (ntiles + 3) / 4;
  for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int base = (grp * 4 + wave) * 32;
    const bool valid = base + j < a.R;  // a tile past the end: every lane invalid, contributes zeros
    const int row = valid ? base + j : a.R - 1;
    const int traj = row / a.M;

    Act<1> X, H;
    SplitAct<1> sp;
    half8 st[BWD ? NLAY : 1][4];  // f16 copy of every layer's input (the hi halves of its operand split)
    auto keep = [&](int li) {
      if constexpr (BWD) {
#pragma unroll
        for (int k = 0; k < 4; ++k) st[li][k] = sp.hi[k][0];
      }
    };

    // ------------------------------------------------------------------ forward (the inference kernels' arithmetic)
    if constexpr (FIRST) {
      zero_act(X);
      constexpr int KS0 = (D + 2) / 2;
#pragma unroll
      for (int s = 0; s < KS0; ++s) {
        const int comp = 2 * s + h;
        const float b = comp < D ? a.states[static_cast<size_t>(row) * D + comp] : (comp == D ? 1.f : 0.f);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const float w = sW0[(32 * t + j) * kW0Cols + comp];
          X.v[t][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w, b, X.v[t][0], 0, 0, 0);
        }
      }
      relu<1, false>(X, true);
      split_act_nr(X, sp, neg_one);
      keep(0);
      add_bias<1, false>(sBias, H, h, 1.f);
      layer_fwd(0, sp, H);
      relu<1, false, true>(H, true);
      split_act_nr(H, sp, neg_one);
      keep(1);
      add_bias_packed<1>(sBias + kUnits, X, h);
      layer_fwd(1, sp, X);
      relu<1, false, true>(X, true);
    } else {
      load_rows_f32(a.act_in, X, row, h, 1.f);
    }
    if constexpr (PART == kEncFwd) stash_store(a.act_out, X, row, valid, h);

    if constexpr (BWD) {
    float raw[NOUT];
    if constexpr (JOIN) {
      load_rows_f32(a.traj_bias, H, traj, h, 1.f);
      split_act_nr(X, sp, neg_one);
      keep(2 - L0);
      layer_fwd(2 - L0, sp, H);
      if constexpr (KIND == kMeasure) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) H.v[t][0][r] = relu_keepnan(H.v[t][0][r]);
      } else {
        saturate<1>(H);
      }
      static_for<NRES>([&](auto ic) {
        constexpr int l1 = 3 + 2 * decltype(ic)::value;
        split_act_nr(H, sp, neg_one);
        keep(l1 - L0);
        add_bias<1, false>(sBias + l1 * kUnits, X, h, 1.f);
        layer_fwd(l1 - L0, sp, X);
        relu<1, false, true>(X, true);
        split_act_nr(X, sp, neg_one);
        keep(l1 + 1 - L0);
        add_bias_packed<1>(sBias + (l1 + 1) * kUnits, H, h);
        layer_fwd(l1 + 1 - L0, sp, H);
        relu<1, false, true>(H, true);
      });
      // head: each lane holds 32 of the 64 features of its particle
#pragma unroll
      for (int o = 0; o < NOUT; ++o) {
        float part = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(sWhead + o * kUnits + 32 * t + 8 * g + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) part = __builtin_fmaf(w[e], H.v[t][0][4 * g + e], part);
          }
        raw[o] = part + __shfl_xor(part, 32) + sBhead[o];
      }
      stash_store_h(a.h_last_h, H, row, valid, h);
    }

    // ------------------------------------------------------------------ backward
    Act<1> G, T;
    // in -> (scaled, split) operand of W_l^T; exchange (dz_l, a_l) and add this workgroup's 128 particles to dW_l
    auto bwd_layer = [&](auto lc, const Act<1>& in, Act<1>& acc, float m) {
      constexpr int li = decltype(lc)::value - L0;
      // c = 2^(134 - e): m c in [2^7, 2^8); rows that are all zero / non-finite / below 2^-110 keep c = 1
      const int e = (__float_as_int(m) >> 23) & 0xff;
      const bool ok = e > 16 && e < 255;
      const float c = ok ? __int_as_float((261 - e) << 23) : 1.f;
      const float inv = ok ? __int_as_float((e - 7) << 23) : 1.f;
      // the group's largest row magnitude: every wave announces its tile's, all read the four after the barrier
      const float mT = mmf::wave_max(m);
      if (lane == 0) tmax[wave] = mT;
      Act<1> x;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          x.v[t][0][r] = in.v[t][0][r] * c;
          acc.v[t][0][r] *= c;
        }
      SplitAct<1> bs;
      split_act_nr(x, bs, neg_one);
      __syncthreads();  // (A) every wave is done reading the previous layer's slots; the four tile maxima are written
      const float mG = fmaxf(fmaxf(tmax[0], tmax[1]), fmaxf(tmax[2], tmax[3]));
      const int eG = __builtin_amdgcn_readfirstlane((__float_as_int(mG) >> 23) & 0xff);
      // running exponent of this layer's accumulators: a larger group rescales what has been summed so far (exact
      // powers of two, rare: only when a new largest magnitude appears), a smaller one is stored relative to it
      if (eG > 40 && eG < 255 && eG > Eacc[li]) {
        const int sh = Eacc[li] ? Eacc[li] - eG + 127 : 0;
        const float down = sh > 0 ? __int_as_float(sh << 23) : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) accW[li][r] *= down;
        accB[li] *= down;
        Eacc[li] = eG;
      }
      const int delta = Eacc[li] - e;
      // exchanged value = dz 2^(141 - Eacc): the group's largest row at most in [2^14, 2^15)
      const float fac32 = (ok && Eacc[li] != 0 && delta <= 31 && delta >= 0) ? __int_as_float((134 - delta) << 23) : 0.f;
      const _Float16 fac = static_cast<_Float16>(fac32);
      const half8 fac8 = {fac, fac, fac, fac, fac, fac, fac, fac};
      unsigned char* slot = ldsb + LY::kXchgOff + wave * LY::kSlotBytes;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const u32x4 dzh = __builtin_bit_cast(u32x4, bs.hi[k][0] * fac8);
        const u32x4 ah = __builtin_bit_cast(u32x4, st[li][k]);
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
          const int off = wx ^ (64 * (k >> 1) + 16 * (2 * (k & 1) + gg));
          *reinterpret_cast<uint2*>(slot + off) = make_uint2(dzh[2 * gg], dzh[2 * gg + 1]);
          *reinterpret_cast<uint2*>(slot + 4096 + off) = make_uint2(ah[2 * gg], ah[2 * gg + 1]);
        }
      }
      layer_bwd(li, bs, acc);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc.v[t][0][r] *= inv;
      __syncthreads();  // (B) the four slots are written
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const unsigned char* sl = ldsb + LY::kXchgOff + s * LY::kSlotBytes;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const half8 A = lds_tr_pair(sl + (rxa + 2048 * kk), sl + ((rxa ^ 32) + 512 + 2048 * kk));
          const half8 B = lds_tr_pair(sl + 4096 + (rxb + 2048 * kk), sl + 4096 + ((rxb ^ 32) + 512 + 2048 * kk));
          accW[li] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, accW[li], 0, 0, 0);
          if (nt == 0) {
            const half2v one2 = {static_cast<_Float16>(1.f), static_cast<_Float16>(1.f)};
#pragma unroll
            for (int d2 = 0; d2 < 4; ++d2) accB[li] = __builtin_amdgcn_fdot2(half2v{A[2 * d2], A[2 * d2 + 1]}, one2, accB[li], false);
          }
        }
      }
    };
    auto bwd_block = [&](auto l1c) {  // residual block of layers l1, l1 + 1; G = dL / d (block output, post-ReLU)
      constexpr int l1 = decltype(l1c)::value;
      // (the caller has applied the block output's ReLU mask to G)       dz2 = G
      zero_act(T);
      bwd_layer(std::integral_constant<int, l1 + 1>{}, G, T, row_absmax(G));   // T = W2^T dz2
      mask_by_halves(st[l1 + 1 - L0], T);                                        // dz1 = T * [h > 0]
      bwd_layer(std::integral_constant<int, l1>{}, T, G, row_absmax(T));       // G = dz2 + W1^T dz1
    };

    if constexpr (HEAD) {
      float go[NOUT];
      if constexpr (KIND == kMeasure) {
        go[0] = valid ? a.d_out[row] : 0.f;
      } else {
        // x' = x + dir sigmoid(gate): d dir_i = g_i s, d gate = (sum_i g_i dir_i) s (1 - s)
        const float s = 1.0f / (1.0f + expf(-raw[D]));
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < D; ++i) {
          const float gi = valid ? a.g_next[static_cast<size_t>(row) * D + i] : 0.f;
          go[i] = gi * s;
          dot += gi * raw[i];
        }
        go[D] = dot * s * (1.0f - s);
        if (valid && h == 0) {
#pragma unroll
          for (int o = 0; o < NOUT; ++o) a.d_raw[static_cast<size_t>(row) * NOUT + o] = go[o];
        }
      }
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float s = 0.f;
#pragma unroll
          for (int o = 0; o < NOUT; ++o) s += sWhead[o * kUnits + 32 * t + rowmap(r, h)] * go[o];
          G.v[t][0][r] = s;
        }
      mask_by_act(H, G);  // the ReLU in front of the head
      static_for<NRES>([&](auto ic) {
        constexpr int b = NRES - 1 - decltype(ic)::value;
        constexpr int l1 = 3 + 2 * b;
        bwd_block(std::integral_constant<int, l1>{});
        // G = dL / d (input of layer l1) = the previous block's output (post-ReLU), or the join layer's output
        if constexpr (l1 > 3 || KIND == kMeasure) mask_by_halves(st[l1 - L0], G);
      });
      // join layer: G = dL / d (its pre-activation output)
      {
        const float mj = row_absmax(G);
        dz_store_h(a.dz_join_h, a.sc_join, G, row, valid, h, tile_absmax(mj));
        zero_act(T);
        bwd_layer(std::integral_constant<int, 2>{}, G, T, mj);
      }
      if constexpr (PART == kTrunk) {
        stash_store(a.g_out, T, row, valid, h);
      } else {
#pragma unroll
        for (int t = 0; t < 2; ++t) G.v[t][0] = T.v[t][0];
      }
    } else {
      load_rows_f32(a.g_in, G, row, h, valid ? 1.f : 0.f);
    }
    if constexpr (FIRST) {
      // G = dL / d (encoder output, post-ReLU): the ReLU that ends block 0 (kEnc: X still holds that output)
      if constexpr (PART == kEnc) mask_by_act(X, G);
      else mask_by_halves(st[2], G);
      bwd_block(std::integral_constant<int, 0>{});
      mask_by_halves(st[0], G);  // first layer (d -> 64): dz_in = da0 * [a0 > 0]
      const float mf = row_absmax(G);
      dz_store_h(a.dz_first_h, a.sc_first, G, row, valid, h, tile_absmax(mf));
      // d states[i] = sum_f W_in[f][i] dz_in[f]
      for (int i = 0; i < D; ++i) {
        float part = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) part += sW0[(32 * t + rowmap(r, h)) * kW0Cols + i] * G.v[t][0][r];
        part += __shfl_xor(part, 32);
        if (valid && h == 0) a.d_states[static_cast<size_t>(row) * D + i] = part;
      }
    }
    }  // BWD
  }

  if constexpr (BWD) {
    // accW[li][r] of lane (c, h2) = dW_l[32 mt + rowmap(r, h2)][32 nt + c] 2^(141 - Eacc): added to this workgroup's partial
#pragma unroll
    for (int li = 0; li < NLAY; ++li) {
      const float fs = Eacc[li] ? __int_as_float((Eacc[li] - 14) << 23) : 0.f;
      float* pw = a.pw + (static_cast<size_t>(L0 + li) * a.slots + blockIdx.x) * kLayerFloats + 32 * nt + j;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float* qp = pw + (32 * mt + rowmap(r, h)) * kUnits;
        *qp += accW[li][r] * fs;
      }
      if (nt == 0) {
        const float v = (accB[li] + __shfl_xor(accB[li], 32)) * fs;
        if (h == 0) a.pb[(static_cast<size_t>(L0 + li) * a.slots + blockIdx.x) * kUnits + 32 * mt + j] += v;
      }
    }
  }
}

template <int D, int NRES, int KIND, int PART>
int launch_fused(const FusedArgs& a, hipStream_t s) {
  using LY = FusedLayout<NRES, PART>;
  const int ntiles = (a.R + 31) / 32;
  int grid = (ntiles + 3) / 4;
  const int cap = PART == kEncFwd ? 256 : (a.slots < 256 ? a.slots : 256);
  grid = grid > cap ? cap : (grid < 1 ? 1 : grid);
  auto k = particle_net_train_fused_kernel<D, NRES, KIND, PART>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, LY::kBytes);
  if (e != hipSuccess) return static_cast<int>(e);
  k<<<grid, 256, LY::kBytes, s>>>(a);
  MMF_CHECK_LAUNCH();
  return 0;
}

template <int D>
int launch_fused_net(const MmfTrainFusedArgs* c, FusedArgs a, hipStream_t s) {
  if (c->kind == kMeasure) {
    if (c->n_res != 2) return MMF_EINVAL;
    return launch_fused<D, 2, kMeasure, kFull>(a, s);
  }
  if (c->n_res != 3) return MMF_EINVAL;
  // dynamics: encoder forward -> trunk (forward + backward) -> encoder (recompute + backward)
  FusedArgs e = a;
  e.act_out = c->act;
  int rc = launch_fused<D, 3, kDynamics, kEncFwd>(e, s);
  if (rc) return rc;
  FusedArgs t = a;
  t.act_in = c->act;
  t.g_out = c->g_act;
  rc = launch_fused<D, 3, kDynamics, kTrunk>(t, s);
  if (rc) return rc;
  FusedArgs b = a;
  b.g_in = c->g_act;
  return launch_fused<D, 3, kDynamics, kEnc>(b, s);
}

}  // namespace

extern "C" int mmf_particle_net_train_fused(const MmfTrainFusedArgs* c, void* stream) {
  if (!c || !c->packed_dual || !c->states || !c->traj_bias || !c->d_states || !c->dz_first_h || !c->sc_first ||
      !c->dz_join_h || !c->sc_join || !c->h_last_h || !c->pw || !c->pb)
    return MMF_EINVAL;
  if (c->N < 0 || c->M < 1 || c->n_slots < 1 || (c->d != 2 && c->d != 3)) return MMF_EINVAL;
  if (c->kind != kDynamics && c->kind != kMeasure) return MMF_EINVAL;
  if (c->kind == kMeasure && !c->d_out) return MMF_EINVAL;
  if (c->kind == kDynamics && (!c->g_next || !c->d_raw || !c->act || !c->g_act)) return MMF_EINVAL;
  if (static_cast<long long>(c->N) * c->M > 0x7fffffffLL / 64) return MMF_ETOOLARGE;
  if (c->N == 0) return 0;
  FusedArgs a{};
  a.blob = c->packed_dual; a.states = c->states; a.traj_bias = c->traj_bias; a.d_out = c->d_out; a.g_next = c->g_next;
  a.d_raw = c->d_raw; a.d_states = c->d_states;
  a.dz_first_h = static_cast<_Float16*>(c->dz_first_h); a.sc_first = c->sc_first;
  a.dz_join_h = static_cast<_Float16*>(c->dz_join_h); a.sc_join = c->sc_join;
  a.h_last_h = static_cast<_Float16*>(c->h_last_h);
  a.pw = c->pw; a.pb = c->pb; a.R = c->N * c->M; a.M = c->M; a.slots = c->n_slots;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return c->d == 2 ? launch_fused_net<2>(c, a, s) : launch_fused_net<3>(c, a, s);
}
