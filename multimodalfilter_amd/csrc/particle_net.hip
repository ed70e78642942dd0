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

#include "particle_net_tiles.h"

namespace {

// ------------------------------------------------------------------------------ packing
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
      } else if (precision == MMF_PREC_F16X3_DUAL) {
        // dual-use image (particle_net_fused.hip): byte b of the layer <-> (row, swizzled chunk, run, element)
        unsigned short hb[2];
        for (int z = 0; z < 2; ++z) {
          const int b = 4 * e + 2 * z;
          const int row = b >> 8, ch = ((b >> 4) & 15) ^ dual_swizzle(row);
          const int run = 2 * (ch & 7) + ((b >> 3) & 1), el = (b >> 1) & 3;
          const int col = 16 * (run >> 2) + 8 * (run & 1) + 4 * ((run >> 1) & 1) + el;
          const float w = W[row * stride + coff + col];
          const __half hi = __float2half_rn(w);
          hb[z] = (ch >> 3) ? f16_bits_rz(w - __half2float(hi)) : __half_as_ushort(hi);
        }
        v = __uint_as_float(static_cast<unsigned>(hb[0]) | (static_cast<unsigned>(hb[1]) << 16));
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

// Per-workgroup start / end stamps + XCD of the LAST launch (scripts/debug/k2_wg_spread.py builds a library of its own with
// -DMMF_K2_WG_STAMPS; compiled out of the product).
#ifdef MMF_K2_WG_STAMPS
__device__ long long g_k2_stamp[2][512];
__device__ int g_k2_xcc[512];
#define K2_WG_STAMP(which)                                                                          \
  do {                                                                                              \
    if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 512) {                                  \
      unsigned x_;                                                                                  \
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x_));                             \
      g_k2_stamp[which][blockIdx.x] = wall_clock64();                                               \
      g_k2_xcc[blockIdx.x] = x_ & 0xf;                                                              \
    }                                                                                               \
  } while (0)
#else
#define K2_WG_STAMP(which)
#endif

// blockIdx.y selects one of up to MMF_LOOP_MAX_MEAS independent problems of the same shape (the
// sub-filters of a fused EKF evaluate their Jacobians in one launch).
struct NetArgsMulti {
  NetArgs a[MMF_LOOP_MAX_MEAS];
};

template <int D, int NRES, int KIND, int CT, int PREC, int WPS, bool PIPE = false>
__global__ __launch_bounds__(WPS * 256, WPS) void particle_net_kernel(NetArgsMulti multi) {
  const NetArgs a = multi.a[blockIdx.y];
  K2_WG_STAMP(0);
  static_assert(!PIPE || (CT == 2 && PREC == MMF_PREC_F16X3 && KIND != kJacobian), "pipelined halves: f16x3, 64-particle tiles");
  constexpr int kThreads = WPS * 256;           // WPS waves per SIMD, one workgroup per CU (LDS)
  constexpr int kWavesPerBlock = kThreads / MMF_WAVE;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr bool JAC = KIND == kJacobian;
  constexpr bool F16 = PREC == MMF_PREC_F16X3;
  constexpr int NOUT = (KIND == kMeasure) ? 1 : D + 1;
  constexpr int TILE = 32 * CT;
  static_assert(!JAC || D <= 3, "jacobian groups are 4 columns: primal + up to 3 tangents");

  // stage this network's fragment-ordered weights in LDS once per workgroup.  ASYNC (round 6, the pipelined variant): only the
  // first-layer / bias / head sections are copied through registers; the 64 x 64 layers (7 or 9 x 16 KB) travel by LDS-DMA
  // (global_load_lds_dwordx4: no destination registers, 1 KB per wave-instruction, the LDS image is the blob's own order), layers
  // 0 and 1 are waited for before the first tile, the rest lands UNDER that tile's first layer and a half and is waited
  // for (one more barrier, in every wave's first tile) before layer 2's fragments are requested.  The copy was 4-7 us of
  // every launch -- ~4 % of a 256 x 4096 launch, a quarter of a 32 x 4096 one.
#ifdef MMF_K2_SYNC_STAGING   // scripts/debug/k2_async_staging_ab.sh builds the synchronous copy for the A/B
  constexpr bool ASYNC_STAGE = false;
#else
  constexpr bool ASYNC_STAGE = PIPE && kThreads == 512;
#endif
  constexpr int NLAYERS = 3 + 2 * NRES;
  const int lane = threadIdx.x & 63;
  const int j = lane & 31, h = lane >> 5;
  const int wave_global = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int waves_total = gridDim.x * kWavesPerBlock;
  const int ntiles = (a.R + TILE - 1) / TILE;
  if constexpr (!ASYNC_STAGE) {
    const float4* src = reinterpret_cast<const float4*>(a.packed);
    float4* dst = reinterpret_cast<float4*>(lds);
    mmf::stage_to_lds<blob_floats(NRES) / 4, kThreads>(src, dst, threadIdx.x);
    __syncthreads();
  } else {
    const float4* src = reinterpret_cast<const float4*>(a.packed);
    float4* dst = reinterpret_cast<float4*>(lds);
    static_assert(off_layers() % 256 == 0 && off_bias(NRES) % 4 == 0 && blob_floats(NRES) % 4 == 0, "16-byte sections, 1 KB pieces");
    for (int i = threadIdx.x; i < off_layers() / 4; i += kThreads) dst[i] = src[i];
    for (int i = off_bias(NRES) / 4 + threadIdx.x; i < blob_floats(NRES) / 4; i += kThreads) dst[i] = src[i];
  }

  // -1.0f in an SGPR, opaque to the optimiser (see split_pair); the asm emits no instruction
  float neg_one = -1.0f;
  asm volatile("" : "+s"(neg_one));

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
  bool layers_pending = false;  // wave-uniform: layers 2 .. of the weights are still on their way
  __shared__ int s_next_index;  // the tile claims' counter (below)
  if constexpr (ASYNC_STAGE) {
    if (threadIdx.x == 0) s_next_index = 2 * kWavesPerBlock;
    // the first tile's inputs are WAITED FOR here, before the DMAs are issued: the memory counter retires in order, so a
    // wait for an ordinary load issued after them would be a wait for all of them
#pragma unroll
    for (int s = 0; s < KS0; ++s)
#pragma unroll
      for (int c = 0; c < CT; ++c) asm volatile("" : "+v"(bnext[s][c]));
    const unsigned lds_base = static_cast<unsigned>(reinterpret_cast<uintptr_t>(lds)) + off_layers() * 4;
    const unsigned char* gsrc = reinterpret_cast<const unsigned char*>(a.packed + off_layers()) + lane * 16;
    const int wv = threadIdx.x >> 6;
#pragma unroll
    for (int l = 0; l < NLAYERS; ++l)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const unsigned piece = static_cast<unsigned>(l * 16 + wv * 2 + q) * 1024u;   // this wave's 2 KB of layer l
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + piece);
        const unsigned char* sp = gsrc + piece;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(sp), "s"(dst) : "memory");
      }
    // layers 0 and 1 of every wave's share (the counter retires in order: all but the last 2 (NLAYERS - 2) DMAs), and the
    // register-staged sections
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * (NLAYERS - 2)) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    layers_pending = true;
  }
  auto await_layers = [&]() {  // every wave exactly once: in its first tile, or after the loop if it has none
    if (layers_pending) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      layers_pending = false;
    }
  };

  // Tiles of a workgroup are CLAIMED, not dealt: the waves that share a SIMD do not advance at the same rate (the
  // arbiter favours one of two equally old waves), and with a fixed stride the favoured one runs out of tiles while its
  // partner still has one or two to go -- alone on a SIMD whose schedule is built for two.  Index i of a workgroup is
  // tile (i / W) * (G W) + b W + i % W (the old stride); the first W are taken by wave id, the rest from an LDS counter,
  // one ahead (the next tile's first-layer inputs are prefetched).  Which wave computes a tile does not change it.
  // (measured, same box, alternating: headline step 0.5581 -> 0.5542 ms; with one tile per wave -- 32 x 4096 -- there is
  // nothing to claim and the counter is not touched.)
  const bool claims = ntiles > 2 * waves_total;  // wave-uniform
  if constexpr (!ASYNC_STAGE) {  // (ASYNC_STAGE: initialised ahead of the barrier that follows the DMAs' issue -- a
    if (claims) {                //  __syncthreads() here would drain them)
      if (threadIdx.x == 0) s_next_index = 2 * kWavesPerBlock;
      __syncthreads();
    }
  }
  auto tile_of = [&](int i) { return (i / kWavesPerBlock) * waves_total + blockIdx.x * kWavesPerBlock + i % kWavesPerBlock; };
  auto claim = [&](int after) {
    if (!claims) return after + kWavesPerBlock;  // the fixed stride
    int v = 0;
    if (lane == 0) v = atomicAdd(&s_next_index, 1);
    return __builtin_amdgcn_readfirstlane(v);
  };
  int i_next = (threadIdx.x >> 6) + kWavesPerBlock;  // the second round is still dealt: its inputs are requested below

  for (int tile = wave_global; tile < ntiles;) {
    const int tile_next = tile_of(i_next);
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
    if (tile_next < ntiles) first_layer_inputs(tile_next, bnext);
    i_next = claim(i_next);
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
    relu<CT, JAC>(X, primal);

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
        if constexpr (ASYNC_STAGE && l == 1) await_layers();  // the next stage requests layer 2's fragments
        mstage(I1{}, layer);
        if constexpr (l + 1 < NL) vstage(I0{}, std::integral_constant<int, l + 1>{});
        else relu_half<0>(H);
        pin_mfma_valu_interleave<5>();
        __builtin_amdgcn_sched_barrier(0);
      });
      relu_half<1>(H);
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
            a.states_out[traj * D + i] = jac_primal(a.states_in[traj * D + i], dirp[i], sg);
        } else if (role <= D) {
          const float dgate = mine[D];  // tangent columns carry no bias
#pragma unroll
          for (int i = 0; i < D; ++i) {
            const float dv = jac_tangent(mine[i], sg, dirp[i], dgate, (i == role - 1) ? 1.f : 0.f);
            a.jac[(static_cast<size_t>(traj) * D + i) * D + (role - 1)] = dv;
          }
        }
      }
    }
    tile = tile_next;
  }
  if constexpr (ASYNC_STAGE) await_layers();
  K2_WG_STAMP(1);
}

template <int D, int NRES, int KIND, int PREC, int CT, int WPS, bool PIPE = false>
int launch_variant(const NetArgsMulti& m, int count, hipStream_t s) {
  const NetArgs& a = m.a[0];
  const size_t lds = static_cast<size_t>(blob_floats(NRES)) * sizeof(float);
  constexpr int waves = WPS * 4, tile = 32 * CT;
  const int ntiles = (a.R + tile - 1) / tile;
  int grid = (ntiles + waves - 1) / waves;
  if (grid > 256) grid = 256;
  if (grid < 1) grid = 1;
  auto k = particle_net_kernel<D, NRES, KIND, CT, PREC, WPS, PIPE>;
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
  // f16x3: the two 32-particle halves of a 64-particle tile half a layer apart (PIPE).  The organisations that were
  // measured and lost -- 32-particle tiles at 2 or 3 waves per SIMD, the unpipelined 64-particle tile, the row-tile
  // pipeline (bit-identical, 1-6 % slower) -- are gone from the product; profiles/r02 .. r04 keep their A/B files.
  if constexpr (PREC == MMF_PREC_F16X3 && KIND != kJacobian) {
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

#ifdef MMF_K2_WG_STAMPS
extern "C" int mmf_debug_k2_stamps(long long* stamps /* [2][512] */, int* xcc /* [512] */) {
  if (hipDeviceSynchronize() != hipSuccess) return MMF_EINVAL;
  if (hipMemcpyFromSymbol(stamps, HIP_SYMBOL(g_k2_stamp), sizeof(long long) * 1024) != hipSuccess) return MMF_EINVAL;
  return hipMemcpyFromSymbol(xcc, HIP_SYMBOL(g_k2_xcc), sizeof(int) * 512) == hipSuccess ? 0 : MMF_EINVAL;
}
#endif

extern "C" size_t mmf_particle_net_floats(int n_res) {
  if (n_res < 0 || n_res > MMF_MAX_RES) return 0;
  return static_cast<size_t>(blob_floats(n_res));
}

extern "C" int mmf_pack_particle_net(const MmfParticleNetDesc* d, float* packed, int precision,
                                     void* stream) {
  if (!d || !packed) return MMF_EINVAL;
  if (precision != MMF_PREC_F32 && precision != MMF_PREC_F16X3 && precision != MMF_PREC_F16X3_DUAL) return MMF_EINVAL;
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

extern "C" int mmf_pf_measure_multi(const float* const* packed, int count, int n_res, int precision, const float* states,
                                    const float* const* traj_bias, const float* const* modality_logw, int logw_stride,
                                    float* const* loglik, int* range_flag, int N, int M, int d, void* stream) {
  if (!packed || !states || !traj_bias || !modality_logw || !loglik || count < 1 || count > MMF_LOOP_MAX_MEAS) return MMF_EINVAL;
  if (N < 0 || M < 1) return MMF_EINVAL;
  if (static_cast<long long>(N) * M > 0x7fffffffLL / 8) return MMF_ETOOLARGE;
  if (N == 0) return 0;
  NetArgsMulti m{};
  for (int k = 0; k < count; ++k) {
    if (!packed[k] || !traj_bias[k] || !loglik[k]) return MMF_EINVAL;
    NetArgs& a = m.a[k];
    a.packed = packed[k]; a.states_in = states; a.traj_bias = traj_bias[k]; a.mod_logw = modality_logw[k];
    a.logw_stride = logw_stride; a.loglik = loglik[k]; a.combine = 0; a.R = N * M; a.M = M;
    a.range_flag = range_flag;
  }
  return launch_multi<kMeasure>(m, count, d, n_res, precision, static_cast<hipStream_t>(stream));
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
#include "ekf_persistent.inc"
