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
//  * ONE wave per SIMD (the kernel needs ~450 registers per lane) has nothing to hide a round trip behind, so the
//    instruction stream itself is pipelined: weight fragments are requested one group of three MFMAs ahead of their
//    use, across layer and tile boundaries; the exchange slots' operands one product ahead; the transposed product is
//    issued BEFORE the exchange's barrier, so the hand-off runs under the matrix pipe.  Only 256 of the registers can be
//    VGPRs: the two groups that never need one between their definition and their use -- weight-gradient accumulators
//    and the f16 layer inputs -- are parked in AGPRs BY HAND (v_accvgpr_write / _read); left to the register allocator
//    the split costs scratch reloads in the middle of every layer, each an exposed L2 round trip.
//
// The dynamics network (9 layers = 144 KB) does not leave room for the exchange slots: it runs as three launches --
// encoder forward (first layer + block 0: writes its 64 outputs per particle, fp32), trunk (join + 3 blocks + head +
// the sigmoid-gate epilogue, forward and backward; writes the gradient w.r.t. the encoder output) and encoder
// backward -- 1 KB of traffic per particle instead of 5.8.
//
// What still travels through HBM in the round-4 compact format: dz of the first layer and of the join layer and the
// head's input (3 x 128 B per particle), for the narrow reductions of small_grads_h_kernel.

#include "particle_net_train_common.h"

// Phase clocks for scripts/ubench/fused_phases.hip (compiled out of the library): wave 0 of workgroup 0 accumulates
// s_memtime differences per phase of a tile.  The stamps serialise what they measure: read shares.
#ifdef MMF_FUSED_PHASE_CLOCKS
__device__ unsigned long long g_fused_phases[16];
#define FUSED_STAMP(i)                                                                \
  do {                                                                                \
    unsigned long long t_;                                                            \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");        \
    phase_acc[i] += t_ - phase_prev;                                                  \
    phase_prev = t_;                                                                  \
  } while (0)
#else
#define FUSED_STAMP(i)
#endif

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
  // small section: W0 (64 x 8) | biases (NL x 64) | head weights (4 x 64) | head bias (8) | W0 transposed (4 x 64)
  static constexpr int kTailFloats = blob_floats(NRES) - off_bias(NRES);
  static constexpr int kSmallFloats = off_layers() + kTailFloats + 4 * kUnits;
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
  float* d_states;         // (R, D): through the network only ..
  const float* d_states_base;  // .. or, when given, that plus these (R, D) values (may be d_states itself: accumulate)
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
// (r & 7) of fragment 2 t + (r >> 3)); an activation below 2^-25 counts as 0.  The mask is arithmetic: clamp(a 2^24)
// is exactly 0 or 1 for a non-negative f16 (v_fma_mix_f32 reads the half in place, the clamp is its output modifier),
// so a masked element costs two VALU instructions and no VCC round trip (compare + select: three and a wait state).
__device__ __forceinline__ void mask_by_halves(const half8 (&s)[4], Act<1>& g) {
  const float big = 16777216.f;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const unsigned w = __builtin_bit_cast(u32x4, s[2 * t + (r >> 3)])[(r & 7) >> 1];
      float m0, m1;
      asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0] clamp" : "=v"(m0) : "v"(w), "v"(big));
      asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0] clamp" : "=v"(m1) : "v"(w), "v"(big));
      g.v[t][0][r] *= m0;
      g.v[t][0][r + 1] *= m1;
    }
}
__device__ __forceinline__ void mask_by_act(const Act<1>& a, Act<1>& g) {
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) g.v[t][0][r] = a.v[t][0][r] > 0.f ? g.v[t][0][r] : 0.f;
}

// ---- the forward chain, software-pipelined on output ROW tiles (the scheme of particle_net.hip's rowpipe_net_f16, for a
// lone 32-particle tile): an MFMA holds the SIMD's vector issue for 8 of its 32 cycles, so ~6 independent VALU
// instructions per MFMA are free.  Rows 0..31 of a layer's output (tile 0) feed k-steps 0, 1 of the next layer and rows
// 32..63 (tile 1) k-steps 2, 3, so the fragment groups run (t0,s0) (t0,s1) (t1,s0) (t1,s1) | (t0,s2) (t0,s3) |
// (t1,s2) (t1,s3): tile 1 of the PREVIOUS layer is activated / split / parked under the first four groups, the next
// layer's accumulator is initialised under the next two, tile 0 of THIS layer is post-processed under the last two.
constexpr int kPipeT[8] = {0, 0, 1, 1, 0, 0, 1, 1};
constexpr int kPipeS[8] = {0, 1, 0, 1, 2, 3, 2, 3};
enum FusedAct { kFaReluSat = 0, kFaReluKeepNan = 1, kFaSaturate = 2 };

template <int T, int ACT>
__device__ __forceinline__ void act_rows(Act<1>& a) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float v = a.v[T][0][r];
    a.v[T][0][r] = ACT == kFaReluSat ? relu_sat(v) : ACT == kFaReluKeepNan ? relu_keepnan(v) : clamp_sat(v);
  }
}
// rows of tile T -> k-steps 2 T, 2 T + 1 of the next operand
template <int T>
__device__ __forceinline__ void split_rows(const Act<1>& x, SplitAct<1>& o, float neg_one) {
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    u32x4 hh, ll;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      unsigned a, b;
      split_pair_plain(x.v[T][0][8 * u + 2 * p], x.v[T][0][8 * u + 2 * p + 1], neg_one, a, b);
      hh[p] = a;
      ll[p] = b;
    }
    o.hi[2 * T + u][0] = __builtin_bit_cast(half8, hh);
    o.lo[2 * T + u][0] = __builtin_bit_cast(half8, ll);
  }
}
template <int GROUPS, int READS, int VPM>
__device__ __forceinline__ void pin_groups() {
#pragma unroll
  for (int g = 0; g < GROUPS; ++g) {
    __builtin_amdgcn_sched_group_barrier(0x100, READS, 0);  // the fragments of a later group
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);  // VALU in its shadow
    }
  }
}

// lanes j and j + 32 hold the two halves of particle j's features: combine them with ONE v_permlane32_swap (a VALU
// operation) instead of a __shfl_xor (ds_bpermute: an LDS-pipe round trip a lone wave per SIMD cannot hide)
__device__ __forceinline__ float halves_max(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float halves_sum(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float row_absmax_swap(const Act<1>& a) {
  float m = 0.f;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) m = fmaxf(m, fabsf(a.v[t][0][r]));
  return halves_max(m);
}

// ---- registers parked in AGPRs by hand (see the header).  The compiler cannot look into the asm, so the wait states
// it would insert around MFMA results are spelled out where they can matter: `settle_mfma_result` on a value an MFMA
// has just written, before it is parked; `settle_mfma_operand` on one fetched for an MFMA's C operand.
__device__ __forceinline__ unsigned agpr_park(unsigned v) {
  unsigned a;
  asm("v_accvgpr_write_b32 %0, %1" : "=a"(a) : "v"(v));
  return a;
}
__device__ __forceinline__ unsigned agpr_fetch(unsigned a) {
  unsigned v;
  asm("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a));
  return v;
}
// (wait states tied to the VALUE, so that neither the scheduler nor the asm's neighbours can slip between them and it)
__device__ __forceinline__ void settle_mfma_result(f32x16& acc) { asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc)); }
__device__ __forceinline__ void settle_mfma_operand(f32x16& acc) { asm volatile("s_nop 3" : "+v"(acc)); }
__device__ __forceinline__ unsigned agpr_fetch_settled(unsigned a) {  // right after the park that wrote it
  unsigned v;
  asm volatile("s_nop 3\n\tv_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a));
  return v;
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

// blockIdx.y = network of the call (several measurement networks of one step differentiate independently: one launch)
struct FusedArgsMulti {
  FusedArgs net[MMF_LOOP_MAX_MEAS];
};

template <int D, int NRES, int KIND, int PART>
__global__ __launch_bounds__(256, 1) void particle_net_train_fused_kernel(FusedArgsMulti multi) {
  const FusedArgs& a = multi.net[blockIdx.y];
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
  float* sW0T = small + off_layers() + LY::kTailFloats;  // [i][f] = W_in[f][i]: d states reads 4 features per ds_read_b128
  float* tmax = lds + LY::kScaleOff / 4;                 // the four waves' tile maxima of the layer being exchanged
  {
    const float4* src = reinterpret_cast<const float4*>(a.blob + off_layers() + L0 * kLayerFloats);
    mmf::stage_to_lds<NLAY * kLayerFloats / 4, 256, 4>(src, reinterpret_cast<float4*>(lds), threadIdx.x);
    for (int i = threadIdx.x; i < off_layers(); i += 256) small[i] = a.blob[i];
    for (int i = threadIdx.x; i < LY::kTailFloats; i += 256) small[off_layers() + i] = a.blob[off_bias(NRES) + i];
    sW0T[threadIdx.x] = a.blob[(threadIdx.x & 63) * kW0Cols + (threadIdx.x >> 6)];
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

  // ---- weight fragments, one group (3 MFMAs) ahead of their use, across layer and tile boundaries.
  // Group g = 4 t + s: output row tile t, k-step s.
  struct Frag {
    half8 hi, lo;
  };
  auto frag_rows = [&](int li, int g) -> Frag {  // rows of the image: the forward product's A operand
    const unsigned char* img = ldsb + li * (kLayerFloats * 4);
    const int t = g >> 2, s = g & 3;
    Frag f;
    f.hi = *reinterpret_cast<const half8*>(img + ((base_r ^ (32 * s)) + 8192 * t));
    f.lo = *reinterpret_cast<const half8*>(img + ((base_r ^ (128 + 32 * s)) + 8192 * t));
    return f;
  };
  auto frag_cols = [&](int li, int g) -> Frag {  // columns of the same image through the transposing read: W_l^T
    const unsigned char* img = ldsb + li * (kLayerFloats * 4);
    const int t = g >> 2, s = g & 3;
    Frag f;
    f.hi = lds_tr_pair(img + ((base_t ^ (64 * t)) + 4096 * s), img + ((base_t ^ (64 * t + 32)) + 2048 + 4096 * s));
    f.lo = lds_tr_pair(img + ((base_t ^ (128 + 64 * t)) + 4096 * s), img + ((base_t ^ (128 + 64 * t + 32)) + 2048 + 4096 * s));
    return f;
  };
  Frag cur = frag_rows(0, 0);  // the first product of every tile is the forward product of local layer 0
  // acc += W_li in (TR: W_li^T in); the last group requests group 0 of the NEXT product (layer `ln`, NTR: transposed)
  auto product = [&](auto trc, auto ntrc, int li, int ln, const SplitAct<1>& sp, Act<1>& acc) {
    constexpr bool TR = decltype(trc)::value, NTR = decltype(ntrc)::value;
    asm volatile("" ::: "memory");  // see mfma_layer: keep LICM from hoisting the fragment reads out of the tile loop
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      Frag nxt;
      if (g < 7) nxt = TR ? frag_cols(li, g + 1) : frag_rows(li, g + 1);
      else nxt = NTR ? frag_cols(ln, 0) : frag_rows(ln, 0);
      const int t = g >> 2, s = g & 3;
      acc.v[t][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.hi, sp.hi[s][0], acc.v[t][0], 0, 0, 0);
      acc.v[t][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.hi, sp.lo[s][0], acc.v[t][0], 0, 0, 0);
      acc.v[t][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.lo, sp.hi[s][0], acc.v[t][0], 0, 0, 0);
      cur = nxt;
    }
    // issue order: the reads of group g + 1, then the MFMAs of group g
#pragma unroll
    for (int g = 0; g < 7; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x100, TR ? 4 : 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x100, NTR ? 4 : 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  using Rows = std::false_type;
  using Cols = std::true_type;

  // this wave's quadrant of every layer's weight gradient (parked: 16 AGPRs per layer), its half of the bias gradient
  // (stored by nt == 0), and the exponent field the accumulators are relative to: accW = dW 2^(141 - Eacc), 0 = nothing yet
  unsigned accWp[BWD ? NLAY : 1][16];
  unsigned accBp[BWD ? NLAY : 1];
  int Eacc[BWD ? NLAY : 1];
  if constexpr (BWD) {
#pragma unroll
    for (int l = 0; l < NLAY; ++l) {
#pragma unroll
      for (int r = 0; r < 16; ++r) accWp[l][r] = agpr_park(0u);
      accBp[l] = agpr_park(0u);
      Eacc[l] = 0;
    }
  }

#ifdef MMF_FUSED_PHASE_CLOCKS
  unsigned long long phase_acc[16] = {}, phase_prev;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(phase_prev)::"memory");
#endif
  const int ntiles = (a.R + 31) / 32;
  const int ngroups = (ntiles + 3) / 4;
  // a tile's first inputs are requested one tile ahead (a dependent HBM access with nothing to issue otherwise)
  constexpr int KS0 = (D + 2) / 2;
  auto first_inputs = [&](int grp, float (&b)[KS0]) {
    int r = (grp * 4 + wave) * 32 + j;
    r = r < a.R ? r : a.R - 1;
#pragma unroll
    for (int s = 0; s < KS0; ++s) {
      const int comp = 2 * s + h;
      b[s] = comp < D ? a.states[static_cast<size_t>(r) * D + comp] : (comp == D ? 1.f : 0.f);
    }
  };
  float bnext[KS0];
  if constexpr (FIRST) {
    if (static_cast<int>(blockIdx.x) < ngroups) first_inputs(blockIdx.x, bnext);
  }

  for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int base = (grp * 4 + wave) * 32;
    const bool valid = base + j < a.R;  // a tile past the end: every lane invalid, contributes zeros
    const int row = valid ? base + j : a.R - 1;
    const int traj = row / a.M;

    Act<1> X, H;
    SplitAct<1> sp;
    // f16 copy of every layer's input (the hi halves of its operand split), parked in AGPRs until the backward
    unsigned st[BWD ? NLAY : 1][16];
    half8 sv[4];  // the copy the backward is working with: fetched once per layer (exchange, then the ReLU mask)
    auto fetch = [&](int li) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        u32x4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = agpr_fetch(st[li][4 * k + e]);
        sv[k] = __builtin_bit_cast(half8, w);
      }
    };
    // the incoming gradient of the tile, requested now (one or D registers)
    float gin[NOUT];
    if constexpr (JOIN) {
      if constexpr (KIND == kMeasure) {
        gin[0] = valid ? a.d_out[row] : 0.f;
      } else {
#pragma unroll
        for (int i = 0; i < D; ++i) gin[i] = valid ? a.g_next[static_cast<size_t>(row) * D + i] : 0.f;
      }
    }
    FUSED_STAMP(0);  // tile top: input requests

    // ------------------------------------------------------------------ forward (the inference kernels' arithmetic,
    // pipelined on output row tiles: see kPipeT / kPipeS)
    Frag fr[3];  // fragment groups in flight: the one being multiplied and the next two
    fr[0] = cur;
    fr[1] = frag_rows(0, 4 * kPipeT[1] + kPipeS[1]);
    auto park_ksteps = [&](int li, int k0) {  // hi halves of k-steps k0, k0 + 1 of the operand split -> st[li]
      if constexpr (BWD) {
#pragma unroll
        for (int k = k0; k < k0 + 2; ++k) {
          const u32x4 w = __builtin_bit_cast(u32x4, sp.hi[k][0]);
#pragma unroll
          for (int e = 0; e < 4; ++e) st[li][4 * k + e] = agpr_park(w[e]);
        }
      }
    };
    if constexpr (FIRST) {
      float bcur[KS0];
#pragma unroll
      for (int s = 0; s < KS0; ++s) bcur[s] = bnext[s];
      if (grp + static_cast<int>(gridDim.x) < ngroups) first_inputs(grp + gridDim.x, bnext);
      zero_act(X);
#pragma unroll
      for (int s = 0; s < KS0; ++s) {
        const int comp = 2 * s + h;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const float w = sW0[(32 * t + j) * kW0Cols + comp];
          X.v[t][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w, bcur[s], X.v[t][0], 0, 0, 0);
        }
      }
      relu<1, false>(X, true);
      add_bias<1, false>(sBias, H, h, 1.f);
    } else {
      load_rows_f32(a.act_in, X, row, h, 1.f);
      load_rows_f32(a.traj_bias, H, traj, h, 1.f);
    }
    split_rows<0>(X, sp, neg_one);
    split_rows<1>(X, sp, neg_one);
    park_ksteps(0, 0);
    park_ksteps(0, 2);
    static_for<NLAY>([&](auto lc) {
      constexpr int li = decltype(lc)::value, l = L0 + li;
      constexpr bool out_h = l == 0 || l == 2 || (l > 2 && (l - 3) % 2 == 1);
      Act<1>& out = out_h ? H : X;
      Act<1>& prev = out_h ? X : H;  // this layer's input activation; its registers become the next layer's accumulator
      constexpr bool last = li == NLAY - 1;
      // the activation between layer l - 1 and l, and between l and l + 1 (the join layer's output: ReLU for the
      // measurement networks, none -- only the f16 saturation -- for the dynamics)
      constexpr int act_in = l == 3 ? (KIND == kMeasure ? kFaReluKeepNan : kFaSaturate) : kFaReluSat;
      constexpr int act_out = l + 1 == 3 ? (KIND == kMeasure ? kFaReluKeepNan : kFaSaturate) : kFaReluSat;
      auto step = [&](auto gc) {
        constexpr int G = decltype(gc)::value, i = 8 * li + G, n = i + 2;
        if constexpr (n < 8 * NLAY) fr[n % 3] = frag_rows(n / 8, 4 * kPipeT[n % 8] + kPipeS[n % 8]);
        else if constexpr (n == 8 * NLAY) fr[n % 3] = BWD ? frag_cols(NLAY - 1, 0) : frag_rows(0, 0);  // what follows the forward
        constexpr int t = kPipeT[G], sk = kPipeS[G];
        const Frag f = fr[i % 3];
        out.v[t][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.hi, sp.hi[sk][0], out.v[t][0], 0, 0, 0);
        out.v[t][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.hi, sp.lo[sk][0], out.v[t][0], 0, 0, 0);
        out.v[t][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.lo, sp.hi[sk][0], out.v[t][0], 0, 0, 0);
      };
      asm volatile("" ::: "memory");  // keep the fragment reads inside the tile loop (see mfma_layer)
      __builtin_amdgcn_sched_barrier(0);
      // ---- region 1: k-steps 0, 1 of both output tiles || tile 1 of the previous layer -> k-steps 2, 3
      step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{});
      step(std::integral_constant<int, 2>{}); step(std::integral_constant<int, 3>{});
      if constexpr (li > 0) {
        act_rows<1, act_in>(prev);
        split_rows<1>(prev, sp, neg_one);
        park_ksteps(li, 2);
      }
      pin_groups<4, 2, 6>();
      __builtin_amdgcn_sched_barrier(0);
      // ---- region 2: tile 0 completes || the next layer's accumulator: bias, skip + bias, or the per-trajectory term
      step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
      if constexpr (!last) {
        constexpr int m = l + 1;
        if constexpr (m == 2) {
          // the join's per-trajectory term (an L2 hit; consumed by the join layer's first MFMAs, twelve MFMAs from here)
          load_rows_f32(a.traj_bias, prev, traj, h, 1.f);
        } else if constexpr (m == 1 || (m > 3 && (m - 3) % 2 == 1)) {
          add_bias_packed<1>(sBias + m * kUnits, prev, h);  // the block's skip + bias
        } else {
          add_bias<1, false>(sBias + m * kUnits, prev, h, 1.f);
        }
      }
      pin_groups<2, 2, 6>();
      __builtin_amdgcn_sched_barrier(0);
      // ---- region 3: tile 1 completes || tile 0 of this layer -> k-steps 0, 1 of the next
      step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{});
      act_rows<0, act_out>(out);
      if constexpr (!last) {
        split_rows<0>(out, sp, neg_one);
        park_ksteps(li + 1, 0);
      }
      pin_groups<2, 2, 6>();
      __builtin_amdgcn_sched_barrier(0);
    });
    cur = fr[(8 * NLAY) % 3];
    {
      constexpr bool last_is_h = (L0 + NLAY - 1) == 0 || (L0 + NLAY - 1) == 2 || ((L0 + NLAY - 1) > 2 && (L0 + NLAY - 4) % 2 == 1);
      act_rows<1, kFaReluSat>(last_is_h ? H : X);  // tile 1 of the part's last layer
    }
    if constexpr (PART == kEncFwd) stash_store(a.act_out, X, row, valid, h);

    if constexpr (BWD) {
    float raw[NOUT];
    if constexpr (JOIN) {
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
        raw[o] = halves_sum(part) + sBhead[o];
      }
      stash_store_h(a.h_last_h, H, row, valid, h);
    }
    FUSED_STAMP(1);  // forward

    // ------------------------------------------------------------------ backward
    Act<1> G, T;
    // in -> (scaled, split) operand of W_l^T (RES: accumulated onto acc, the block's skip path; else acc = W_l^T in);
    // exchange (dz_l, a_l) and add this workgroup's 128 particles to dW_l
    auto bwd_layer = [&](auto lc, auto resc, const Act<1>& in, Act<1>& acc, float m) {
      constexpr int li = decltype(lc)::value - L0;
      constexpr bool RES = decltype(resc)::value;
      FUSED_STAMP(2);  // between layers: masks, row maxima, head / join stores
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
          if constexpr (RES) acc.v[t][0][r] *= c;
          else acc.v[t][0][r] = 0.f;
        }
      SplitAct<1> bs;
      split_act_nr(x, bs, neg_one);
      FUSED_STAMP(3);  // prologue: scale, split
      // the transposed product is issued first: the exchange below runs while the matrix pipe works through it
      if constexpr (li > 0) product(Cols{}, Cols{}, li, li - 1, bs, acc);
      else product(Cols{}, Rows{}, 0, 0, bs, acc);  // next: the first forward product of the next tile
      FUSED_STAMP(4);  // transposed product issued
      __syncthreads();  // (A) every wave is done reading the previous layer's slots; the four tile maxima are written
      FUSED_STAMP(5);  // barrier A
      const float mG = fmaxf(fmaxf(tmax[0], tmax[1]), fmaxf(tmax[2], tmax[3]));
      const int eG = __builtin_amdgcn_readfirstlane((__float_as_int(mG) >> 23) & 0xff);
      // running exponent of this layer's accumulators: a larger group rescales what has been summed so far (exact
      // powers of two, rare: only when a new largest magnitude appears), a smaller one is stored relative to it
      float down = 1.f;
      const bool rescale = eG > 40 && eG < 255 && eG > Eacc[li];
      if (rescale) {
        const int sh = Eacc[li] ? Eacc[li] - eG + 127 : 0;
        down = sh > 0 ? __int_as_float(sh << 23) : 0.f;
        Eacc[li] = eG;
      }
      // exchanged value = dz 2^(141 - Eacc): the group's largest row at most in [2^14, 2^15).  (Straight-line: a
      // conditional here becomes an exec-masked block in the middle of the layer.)
      const unsigned delta = static_cast<unsigned>(Eacc[li] - e);  // < 0: only when nothing of this layer is summed yet
      const unsigned dcl = delta < 32u ? delta : 32u;
      const float fac_all = __int_as_float((134 - static_cast<int>(dcl)) << 23);
      const float fac32 = (ok & (Eacc[li] != 0) & (delta < 32u)) ? fac_all : 0.f;
      const _Float16 fac = static_cast<_Float16>(fac32);
      const half8 fac8 = {fac, fac, fac, fac, fac, fac, fac, fac};
      unsigned char* slot = ldsb + LY::kXchgOff + wave * LY::kSlotBytes;
      fetch(li);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const u32x4 dzh = __builtin_bit_cast(u32x4, bs.hi[k][0] * fac8);
        const u32x4 ah = __builtin_bit_cast(u32x4, sv[k]);
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
          const int off = wx ^ (64 * (k >> 1) + 16 * (2 * (k & 1) + gg));
          *reinterpret_cast<uint2*>(slot + off) = make_uint2(dzh[2 * gg], dzh[2 * gg + 1]);
          *reinterpret_cast<uint2*>(slot + 4096 + off) = make_uint2(ah[2 * gg], ah[2 * gg + 1]);
        }
      }
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc.v[t][0][r] *= inv;
      // this layer's accumulators come out of their AGPRs for the eight products below
      f32x16 accW;
#pragma unroll
      for (int r = 0; r < 16; ++r) accW[r] = __uint_as_float(agpr_fetch(accWp[li][r]));
      if (rescale) {  // wave-uniform, rare
#pragma unroll
        for (int r = 0; r < 16; ++r) accW[r] *= down;
      }
      settle_mfma_operand(accW);
      FUSED_STAMP(6);  // exchange writes + epilogue
      __syncthreads();  // (B) the four slots are written
      FUSED_STAMP(7);  // barrier B
      // operands of product i = 2 slot + k-step, one product ahead of the MFMA that takes them (four ahead: no faster,
      // 48 more registers)
      auto xload = [&](int i, half8& A, half8& B) {
        const unsigned char* sl = ldsb + LY::kXchgOff + (i >> 1) * LY::kSlotBytes + 2048 * (i & 1);
        A = lds_tr_pair(sl + rxa, sl + ((rxa ^ 32) + 512));
        B = lds_tr_pair(sl + 4096 + rxb, sl + 4096 + ((rxb ^ 32) + 512));
      };
      __builtin_amdgcn_sched_barrier(0);
      half8 A, B;
      xload(0, A, B);
      const half2v one2 = {static_cast<_Float16>(1.f), static_cast<_Float16>(1.f)};
      float bsum = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        half8 An, Bn;
        if (i < 7) xload(i + 1, An, Bn);
        accW = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, accW, 0, 0, 0);
#pragma unroll
        for (int d2 = 0; d2 < 4; ++d2) bsum = __builtin_amdgcn_fdot2(half2v{A[2 * d2], A[2 * d2 + 1]}, one2, bsum, false);
        A = An;
        B = Bn;
      }
      __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);  // operands of products 0 and 1
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        if (i < 6) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      accBp[li] = agpr_park(__float_as_uint(__uint_as_float(agpr_fetch(accBp[li])) * (rescale ? down : 1.f) + bsum));
      settle_mfma_result(accW);
#pragma unroll
      for (int r = 0; r < 16; ++r) accWp[li][r] = agpr_park(__float_as_uint(accW[r]));
      FUSED_STAMP(8);  // weight-gradient products
    };
    using Skip = std::true_type;
    using NoSkip = std::false_type;
    auto bwd_block = [&](auto l1c) {  // residual block of layers l1, l1 + 1; G = dL / d (block output, post-ReLU)
      constexpr int l1 = decltype(l1c)::value;
      // (the caller has applied the block output's ReLU mask to G)       dz2 = G
      bwd_layer(std::integral_constant<int, l1 + 1>{}, NoSkip{}, G, T, row_absmax_swap(G));  // T = W2^T dz2
      mask_by_halves(sv, T);                                                                   // dz1 = T * [h > 0]  (sv = st[l1 + 1])
      bwd_layer(std::integral_constant<int, l1>{}, Skip{}, T, G, row_absmax_swap(T));        // G = dz2 + W1^T dz1
    };

    if constexpr (HEAD) {
      float go[NOUT];
      if constexpr (KIND == kMeasure) {
        go[0] = gin[0];
      } else {
        // x' = x + dir sigmoid(gate): d dir_i = g_i s, d gate = (sum_i g_i dir_i) s (1 - s)
        const float s = 1.0f / (1.0f + expf(-raw[D]));
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < D; ++i) {
          go[i] = gin[i] * s;
          dot += gin[i] * raw[i];
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
        if constexpr (l1 > 3 || KIND == kMeasure) mask_by_halves(sv, G);  // sv = st[l1], left by the block's last layer
      });
      // join layer: G = dL / d (its pre-activation output)
      {
        const float mj = row_absmax_swap(G);
        dz_store_h(a.dz_join_h, a.sc_join, G, row, valid, h, tile_absmax(mj));
        bwd_layer(std::integral_constant<int, 2>{}, NoSkip{}, G, T, mj);
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
      else mask_by_halves(sv, G);  // sv = st[2], left by the join layer
      bwd_block(std::integral_constant<int, 0>{});
      mask_by_halves(sv, G);  // first layer (d -> 64): dz_in = da0 * [a0 > 0]  (sv = st[0])
      const float mf = row_absmax_swap(G);
      dz_store_h(a.dz_first_h, a.sc_first, G, row, valid, h, tile_absmax(mf));
      // d states[i] = sum_f W_in[f][i] dz_in[f]: four features of the lane's 32 per ds_read_b128 of the transposed copy
      float ds[D];
#pragma unroll
      for (int i = 0; i < D; ++i) ds[i] = 0.f;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int i = 0; i < D; ++i) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(sW0T + i * kUnits + 32 * t + 8 * g + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) ds[i] = __builtin_fmaf(w[e], G.v[t][0][4 * g + e], ds[i]);
          }
#pragma unroll
      for (int i = 0; i < D; ++i) {
        float v = halves_sum(ds[i]);
        if (valid && h == 0) {
          if (a.d_states_base) v += a.d_states_base[static_cast<size_t>(row) * D + i];
          a.d_states[static_cast<size_t>(row) * D + i] = v;
        }
      }
    }
    FUSED_STAMP(9);  // first layer, d states
    }  // BWD
  }
#ifdef MMF_FUSED_PHASE_CLOCKS
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (int i = 0; i < 16; ++i) g_fused_phases[i] = phase_acc[i];
#endif

  if constexpr (BWD) {
    // accW[r] of lane (c, h2) = dW_l[32 mt + rowmap(r, h2)][32 nt + c] 2^(141 - Eacc): added to this workgroup's partial
#pragma unroll
    for (int li = 0; li < NLAY; ++li) {
      const float fs = Eacc[li] ? __int_as_float((Eacc[li] - 14) << 23) : 0.f;
      float* pw = a.pw + (static_cast<size_t>(L0 + li) * a.slots + blockIdx.x) * kLayerFloats + 32 * nt + j;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float* qp = pw + (32 * mt + rowmap(r, h)) * kUnits;
        *qp += __uint_as_float(agpr_fetch_settled(accWp[li][r])) * fs;
      }
      const float v = halves_sum(__uint_as_float(agpr_fetch_settled(accBp[li]))) * fs;
      if (nt == 0 && h == 0) a.pb[(static_cast<size_t>(L0 + li) * a.slots + blockIdx.x) * kUnits + 32 * mt + j] += v;
    }
  }
}

template <int D, int NRES, int KIND, int PART>
int launch_fused(const FusedArgs* nets, int n, hipStream_t s) {
  using LY = FusedLayout<NRES, PART>;
  const FusedArgs& a = nets[0];  // every network of a launch has the same rows and slots
  const int ntiles = (a.R + 31) / 32;
  int grid = (ntiles + 3) / 4;
  const int cap = PART == kEncFwd ? 256 : (a.slots < 256 ? a.slots : 256);
  grid = grid > cap ? cap : (grid < 1 ? 1 : grid);
  auto k = particle_net_train_fused_kernel<D, NRES, KIND, PART>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, LY::kBytes);
  if (e != hipSuccess) return static_cast<int>(e);
  FusedArgsMulti m{};
  for (int i = 0; i < n; ++i) m.net[i] = nets[i];
  k<<<dim3(grid, n), 256, LY::kBytes, s>>>(m);
  MMF_CHECK_LAUNCH();
  return 0;
}

template <int D, int NRES, int KIND, int PART>
int launch_fused(const FusedArgs& a, hipStream_t s) {
  return launch_fused<D, NRES, KIND, PART>(&a, 1, s);
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

namespace {

int fill_fused_args(const MmfTrainFusedArgs* c, FusedArgs& a) {
  if (!c || !c->packed_dual || !c->states || !c->traj_bias || !c->d_states || !c->dz_first_h || !c->sc_first ||
      !c->dz_join_h || !c->sc_join || !c->h_last_h || !c->pw || !c->pb)
    return MMF_EINVAL;
  if (c->N < 0 || c->M < 1 || c->n_slots < 1 || (c->d != 2 && c->d != 3)) return MMF_EINVAL;
  if (c->kind != kDynamics && c->kind != kMeasure) return MMF_EINVAL;
  if (c->kind == kMeasure && !c->d_out) return MMF_EINVAL;
  if (c->kind == kDynamics && (!c->g_next || !c->d_raw || !c->act || !c->g_act)) return MMF_EINVAL;
  if (static_cast<long long>(c->N) * c->M > 0x7fffffffLL / 64) return MMF_ETOOLARGE;
  a = FusedArgs{};
  a.blob = c->packed_dual; a.states = c->states; a.traj_bias = c->traj_bias; a.d_out = c->d_out; a.g_next = c->g_next;
  a.d_raw = c->d_raw; a.d_states = c->d_states; a.d_states_base = c->d_states_base;
  a.dz_first_h = static_cast<_Float16*>(c->dz_first_h); a.sc_first = c->sc_first;
  a.dz_join_h = static_cast<_Float16*>(c->dz_join_h); a.sc_join = c->sc_join;
  a.h_last_h = static_cast<_Float16*>(c->h_last_h);
  a.pw = c->pw; a.pb = c->pb; a.R = c->N * c->M; a.M = c->M; a.slots = c->n_slots;
  return 0;
}

}  // namespace

extern "C" int mmf_particle_net_train_fused(const MmfTrainFusedArgs* c, void* stream) {
  FusedArgs a;
  const int rc = fill_fused_args(c, a);
  if (rc) return rc;
  if (c->N == 0) return 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return c->d == 2 ? launch_fused_net<2>(c, a, s) : launch_fused_net<3>(c, a, s);
}

extern "C" int mmf_particle_net_train_fused_multi(const MmfTrainFusedArgs* nets, int n, void* stream) {
  if (!nets || n < 1 || n > MMF_LOOP_MAX_MEAS) return MMF_EINVAL;
  FusedArgs a[MMF_LOOP_MAX_MEAS];
  for (int i = 0; i < n; ++i) {
    const int rc = fill_fused_args(nets + i, a[i]);
    if (rc) return rc;
    // one grid for all: measurement networks of the same depth over the same rows, each with its own outputs
    if (nets[i].kind != kMeasure || nets[i].n_res != 2 || nets[i].d != nets[0].d || nets[i].N != nets[0].N || nets[i].M != nets[0].M ||
        nets[i].n_slots != nets[0].n_slots)
      return MMF_EINVAL;
    for (int j = 0; j < i; ++j)
      if (nets[j].d_states == nets[i].d_states || nets[j].pw == nets[i].pw || nets[j].dz_first_h == nets[i].dz_first_h) return MMF_EINVAL;
  }
  if (nets[0].N == 0) return 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return nets[0].d == 2 ? launch_fused<2, 2, kMeasure, kFull>(a, n, s) : launch_fused<3, 2, kMeasure, kFull>(a, n, s);
}
