// K4: the 32x32 image encoder (R5).
//
// Replaces nn.Sequential(Conv 1->32 k5, ReLU, ResConv 32 k3, Conv 32->16 k3, ReLU,
// Conv 16->8 k3, Flatten, Linear 8192->64, ReLU, ResLinear 64) of
//   /root/reference/crossmodal/door_models/layers.py:43-63  (push_models/layers.py:77-104)
// which the reference runs as ~12 stock torch launches per encoder (26.1 M MAC per image, 72 %
// of it in the two 32->32 3x3 convolutions).  All encoders of a step are batched on blockIdx.y
// (they share the input image but not the weights).
//
// Three paths, selected by `precision`:
//   MMF_PREC_F32    one launch per layer, fp32 activations in HBM, exact fp32 products on
//                   v_mfma_f32_16x16x4_f32 (conv_kernel below: M = output channels, N = 16 pixels of a
//                   row, K = (tap, 4 input channels); input band + zero halo in LDS as [ci][row][40]).
//   MMF_PREC_F16X3  (default) image_encoder_resident.inc: the whole convolution stack as ONE resident
//                   producer / consumer kernel, every activation in LDS row rings as split f16 planes
//                   (only the image and the 8-channel map E cross HBM); then the split-K linear tail below.
//   MMF_PREC_BF16   the same kernel with single bf16 products in the stem and the 32 -> 32 / 32 -> 16 convolutions.
// The TRAINING forward in the f16x3 / bf16 modes is the same resident kernel with every activation the backward reads
// kept on the way (mmf_image_convs_train_forward); round 2's per-layer f16x3 kernels, its previous form, are deleted.
// The 8192->64 linear is a split-K MFMA GEMM followed by a one-wave-per-image tail (bias, ReLU,
// ResLinear 64).  Rooflines and measurements: DESIGN.md section 3, K4.
#include <hip/hip_fp16.h>

#include "mmf_common.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using half8 = __attribute__((ext_vector_type(8))) _Float16;
using half2v = __attribute__((ext_vector_type(2))) _Float16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;

constexpr int kImg = 32;         // images are 32 x 32
constexpr int kBand = 16;        // output rows per workgroup
constexpr int kWP = 40;          // padded LDS row: 4 | 32 pixels | 4
constexpr int kConvThreads = 512;
constexpr int kMaxNets = 4;
constexpr int kFeat = 64;
// f16x3 weight fragments hold w * kWScale: a weight of 0.03 has its residual (2^-11 of it) deep in the
// f16 subnormals, where the "lo" half keeps 3-4 bits instead of 11; scaled by 2^8 both halves are normal
// for |w| >= 5e-4 and the split is exact to 2^-22 again.  Biases enter the accumulators scaled likewise
// and every epilogue multiplies by 1 / kWScale (powers of two: exact).  Measured on the EKF's encoders:
// max error against fp64 5.4e-7 -> see DESIGN.md K4 (the exact-f32-product mode: 1.6e-7).
constexpr float kWScale = 256.0f, kWInv = 1.0f / 256.0f;
constexpr int kFcK = 8 * kImg * kImg;  // 8192
constexpr int kFcSplit = 16;

__host__ __device__ constexpr int ksteps(int cin, int ks) {  // K / 4, padded to a multiple of 4
  return ((cin == 1 ? (ks * ks + 3) / 4 : ks * ks * (cin / 4)) + 3) / 4 * 4;
}
__host__ __device__ constexpr int mtiles(int cout) { return (cout + 15) / 16; }
__host__ __device__ constexpr int conv_w_floats(int cin, int cout, int ks) {
  return mtiles(cout) * ksteps(cin, ks) * 64;
}

// ---- packed blob of one encoder (floats) -------------------------------------------------
struct Layout {
  int w1, b1, w2a, b2a, w2b, b2b, w3, b3, w4, b4, fcw, fcb, r1t, r1b, r2t, r2b;
  int h2a, h2b;          // f16x3 fragment-ordered copies of the two 32 -> 32 convolutions' weights (32x32x16 A fragments)
  int hfc;               // f16x3 fragment-ordered copy of the 8192 -> 64 linear layer's weights
  int hs;                // f16x3 5x5 stem as 2 k-steps of 16 taps: [k-step][hi|lo][lane][8 halves]
  int g3;                // f16x3 conv 32->16 for v_mfma_f32_16x16x32_f16: [tap][hi|lo][lane][8 halves]
  int q2a, q2b, q3, qs;  // MMF_PREC_BF16 twins of h2a / h2b / g3 / hs: same fragment order, slot "hi" = bf16(w), slot "lo" unused
  int g4x;               // f16x3 conv 16->8 for v_mfma_f32_32x32x16_f16 with the three kx taps in M (row 8 kx + co, rows 24-31 zero): [ky][hi|lo][lane][8 halves]
  int total;
};
// floats (= halves / 2) of a 3x3 conv in f16x3 fragment order [tap][kc][hi|lo][lane][8 halves]
__host__ __device__ constexpr int conv_h_floats(int cin) { return 9 * (cin / 16) * 2 * 64 * 8 / 2; }
__host__ __device__ constexpr Layout layout() {
  Layout L{};
  int o = 0;
  L.w1 = o; o += conv_w_floats(1, 32, 5);
  L.b1 = o; o += 32;
  L.w2a = o; o += conv_w_floats(32, 32, 3);
  L.b2a = o; o += 32;
  L.w2b = o; o += conv_w_floats(32, 32, 3);
  L.b2b = o; o += 32;
  L.w3 = o; o += conv_w_floats(32, 16, 3);
  L.b3 = o; o += 16;
  L.w4 = o; o += conv_w_floats(16, 8, 3);
  L.b4 = o; o += 16;
  L.fcw = o; o += kFeat * kFcK;
  L.fcb = o; o += kFeat;
  L.r1t = o; o += kFeat * kFeat;
  L.r1b = o; o += kFeat;
  L.r2t = o; o += kFeat * kFeat;
  L.r2b = o; o += kFeat;
  L.h2a = o; o += conv_h_floats(32);
  L.h2b = o; o += conv_h_floats(32);
  L.hfc = o; o += kFcSplit * (kFcK / kFcSplit / 16) * 2 * 2 * 64 * 8 / 2;  // [split][k-step][tile][hi|lo][lane][8 halves]
  L.hs = o; o += 2 * 2 * 64 * 8 / 2;
  L.g3 = o; o += 9 * 2 * 64 * 8 / 2;
  L.q2a = o; o += conv_h_floats(32);
  L.q2b = o; o += conv_h_floats(32);
  L.q3 = o; o += 9 * 2 * 64 * 8 / 2;
  L.qs = o; o += 2 * 2 * 64 * 8 / 2;
  L.g4x = o; o += 3 * 2 * 64 * 8 / 2;
  L.total = o;
  return L;
}

// k-step s, lane quarter q -> (input channel, ky, kx); false if it is a zero-padding slot
__host__ __device__ inline bool kdecode(int cin, int ks, int s, int q, int* ci, int* ky, int* kx) {
  if (cin == 1) {
    const int t = 4 * s + q;
    if (t >= ks * ks) return false;
    *ci = 0; *ky = t / ks; *kx = t % ks;
    return true;
  }
  const int groups = cin / 4;
  const int tap = s / groups, cg = s % groups;
  if (tap >= ks * ks) return false;
  *ci = 4 * cg + q; *ky = tap / ks; *kx = tap % ks;
  return true;
}

struct PackConv {
  const float* w;  // (cout, cin, ks, ks) torch layout
  const float* b;
  int cin, cout, ks, woff, boff;
};

__global__ void pack_encoder_kernel(MmfImageEncoderDesc d, float* __restrict__ out) {
  constexpr Layout L = layout();
  const PackConv convs[5] = {
      {d.conv_w[0], d.conv_b[0], 1, 32, 5, L.w1, L.b1}, {d.conv_w[1], d.conv_b[1], 32, 32, 3, L.w2a, L.b2a},
      {d.conv_w[2], d.conv_b[2], 32, 32, 3, L.w2b, L.b2b}, {d.conv_w[3], d.conv_b[3], 32, 16, 3, L.w3, L.b3},
      {d.conv_w[4], d.conv_b[4], 16, 8, 3, L.w4, L.b4}};
  // spanning-pool variant: the last convolution has 2 output channels (the other 6 of the 8-wide
  // tile get zero weights and bias) and the linear layer is (64, 64)
  const int cout4 = d.variant == MMF_ENCODER_SPANNING_POOL ? 2 : 8;
  const int fc_in = d.variant == MMF_ENCODER_SPANNING_POOL ? kFeat : kFcK;
  for (int q0 = blockIdx.x * blockDim.x + threadIdx.x; q0 < L.total; q0 += gridDim.x * blockDim.x) {
    float v = 0.f;
    if (q0 < L.fcw) {
      for (int c = 0; c < 5; ++c) {
        const PackConv& p = convs[c];
        const int nw = conv_w_floats(p.cin, p.cout, p.ks);
        if (q0 >= p.woff && q0 < p.woff + nw) {
          // [mt][s4][lane][4]
          const int e = q0 - p.woff;
          const int ksub = e & 3, lane = (e >> 2) & 63, rest = e >> 8;
          const int S4 = ksteps(p.cin, p.ks) / 4;
          const int s4 = rest % S4, mt = rest / S4;
          const int s = 4 * s4 + ksub, i = lane & 15, q = lane >> 4;
          const int co = 16 * mt + i;
          const int cout = c == 4 ? cout4 : p.cout;
          int ci, ky, kx;
          if (co < cout && kdecode(p.cin, p.ks, s, q, &ci, &ky, &kx))
            v = p.w[((co * p.cin + ci) * p.ks + ky) * p.ks + kx];
        } else if (q0 >= p.boff && q0 < p.boff + (p.cout < 16 ? 16 : p.cout)) {
          const int co = q0 - p.boff;
          if (co < (c == 4 ? cout4 : p.cout)) v = p.b[co];
        }
      }
    } else if (q0 < L.fcb) {
      v = (q0 - L.fcw) < kFeat * fc_in ? d.fc_w[q0 - L.fcw] : 0.f;
    } else if (q0 < L.r1t) {
      v = d.fc_b[q0 - L.fcb];
    } else if (q0 < L.r1b) {  // transposed: [k][o]
      const int e = q0 - L.r1t, k = e / kFeat, o = e % kFeat;
      v = d.res_w[0][o * kFeat + k];
    } else if (q0 < L.r2t) {
      v = d.res_b[0][q0 - L.r1b];
    } else if (q0 < L.r2b) {
      const int e = q0 - L.r2t, k = e / kFeat, o = e % kFeat;
      v = d.res_w[1][o * kFeat + k];
    } else if (q0 < L.h2a) {
      v = d.res_b[1][q0 - L.r2b];
    } else if (q0 >= L.g4x) {
      // conv 16->8 in 32x32x16 A fragments, one per ky: element i of lane (row, h) = W[co][ci = 8 h + i][ky][kx] with
      // row = 8 kx + co (rows 24 .. 31 zero): the three kx taps of a row of taps share one pass over the input fragment
      unsigned short hb[2];
      for (int z = 0; z < 2; ++z) {
        const int he = 2 * (q0 - L.g4x) + z;
        const int i = he & 7, lane = (he >> 3) & 63, part = (he >> 9) & 1, ky = he >> 10;
        const int row = lane & 31, kx = row >> 3, co = row & 7, ci = 8 * (lane >> 5) + i;
        const float w = kWScale * (row < 24 && co < cout4 ? d.conv_w[4][(co * 16 + ci) * 9 + 3 * ky + kx] : 0.f);
        const __half hi = __float2half_rn(w);
        hb[z] = part ? __half_as_ushort(__float2half_rn(w - __half2float(hi))) : __half_as_ushort(hi);
      }
      v = __uint_as_float(static_cast<unsigned>(hb[0]) | (static_cast<unsigned>(hb[1]) << 16));
    } else if (q0 >= L.q2a) {
      // bf16 twins (round to nearest even): same index -> weight maps as h2a / h2b / g3 / hs, "lo" slots zero
      const int offs[5] = {L.q2a, L.q2b, L.q3, L.qs, L.g4x};
      int c = 0;
      while (q0 >= offs[c + 1]) ++c;
      unsigned short hb[2];
      for (int z = 0; z < 2; ++z) {
        const int he = 2 * (q0 - offs[c]) + z;
        const int i = he & 7, lane = (he >> 3) & 63, part = (he >> 9) & 1, blk = he >> 10;
        float w = 0.f;
        if (c < 2) {  // conv 32->32: blk = tap * 2 + kc
          const int kc = blk & 1, tap = blk >> 1;
          w = d.conv_w[1 + c][((lane & 31) * 32 + 16 * kc + 8 * (lane >> 5) + i) * 9 + tap];
        } else if (c == 2) {
          w = d.conv_w[3][((lane & 15) * 32 + 8 * (lane >> 4) + i) * 9 + blk];
        } else {
          const int t = 16 * blk + 8 * (lane >> 5) + i;
          w = t < 25 ? d.conv_w[0][(lane & 31) * 25 + t] : 0.f;
        }
        unsigned u = __float_as_uint(w);
        u = (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;  // finite weights: round to nearest even
        hb[z] = part ? 0 : static_cast<unsigned short>(u);
      }
      v = __uint_as_float(static_cast<unsigned>(hb[0]) | (static_cast<unsigned>(hb[1]) << 16));
    } else if (q0 >= L.hs) {
      // fused path: stem (element i of lane (co, h) in k-step s = tap 16 s + 8 h + i, zero past 24) and
      // conv 32->16 in 16x16x32 fragments (element i of lane (co, q) for a tap = input channel 8 q + i)
      const bool stem = q0 < L.g3;
      unsigned short hb[2];
      for (int z = 0; z < 2; ++z) {
        const int he = 2 * (q0 - (stem ? L.hs : L.g3)) + z;
        const int i = he & 7, lane = (he >> 3) & 63, part = (he >> 9) & 1, blk = he >> 10;
        float w;
        if (stem) {
          const int t = 16 * blk + 8 * (lane >> 5) + i;
          w = t < 25 ? d.conv_w[0][(lane & 31) * 25 + t] : 0.f;
        } else {
          w = kWScale * d.conv_w[3][((lane & 15) * 32 + 8 * (lane >> 4) + i) * 9 + blk];
        }
        const __half hi = __float2half_rn(w);
        hb[z] = part ? __half_as_ushort(__float2half_rn(w - __half2float(hi))) : __half_as_ushort(hi);
      }
      v = __uint_as_float(static_cast<unsigned>(hb[0]) | (static_cast<unsigned>(hb[1]) << 16));
    } else if (q0 >= L.hfc) {
      // f16x3 linear layer: element i of lane (row, h) in k-step ks of split sp is W[32 tile + row][512 sp + 16 ks + 8 h + i]
      unsigned short hb[2];
      for (int z = 0; z < 2; ++z) {
        const int he = 2 * (q0 - L.hfc) + z;
        const int i = he & 7, lane = (he >> 3) & 63, part = (he >> 9) & 1, tile = (he >> 10) & 1, rest = he >> 11;
        const int ks = rest % (kFcK / kFcSplit / 16), sp = rest / (kFcK / kFcSplit / 16);
        const int o = 32 * tile + (lane & 31), k = sp * (kFcK / kFcSplit) + 16 * ks + 8 * (lane >> 5) + i;
        const float w = kWScale * (fc_in == kFcK ? d.fc_w[static_cast<size_t>(o) * kFcK + k] : 0.f);
        const __half hi = __float2half_rn(w);
        hb[z] = part ? __half_as_ushort(__float2half_rn(w - __half2float(hi))) : __half_as_ushort(hi);
      }
      v = __uint_as_float(static_cast<unsigned>(hb[0]) | (static_cast<unsigned>(hb[1]) << 16));
    } else {
      // f16x3 sections: two halves per float slot
      const int offs[3] = {L.h2a, L.h2b, L.hfc};
      int c = 0;
      while (q0 >= offs[c + 1]) ++c;
      const float* W = d.conv_w[c + 1];
      const int cin = 32, cout = 32, KC = cin / 16;
      unsigned short hb[2];
      for (int z = 0; z < 2; ++z) {
        const int he = 2 * (q0 - offs[c]) + z;
        const int i = he & 7, lane = (he >> 3) & 63, part = (he >> 9) & 1, rest = he >> 10;
        const int kc = rest % KC, tap = rest / KC;
        const int co = lane & 31, ci = 16 * kc + 8 * (lane >> 5) + i;
        const float w = kWScale * (co < cout ? W[(co * cin + ci) * 9 + tap] : 0.f);
        const __half hi = __float2half_rn(w);
        hb[z] = part ? __half_as_ushort(__float2half_rn(w - __half2float(hi))) : __half_as_ushort(hi);
      }
      v = __uint_as_float(static_cast<unsigned>(hb[0]) | (static_cast<unsigned>(hb[1]) << 16));
    }
    out[q0] = v;
  }
}

// ---- one convolution layer ----------------------------------------------------------------
struct ConvArgs {
  const float* packed[kMaxNets];  // encoder blobs
  const float* in;                // (nets?, N, CIN, 32, 32)
  const float* mask;              // MASK: (nets, N, COUT, 32, 32); the output is zeroed where mask <= 0 (ReLU backward)
  const float* skip;              // (nets, N, COUT, 32, 32) or null
  float* out;                     // (nets, N, COUT, 32, 32)
  long long in_net_stride;        // 0 when every net reads the same input (layer 1)
  int N;
  int woff, boff;
};

template <int CIN, int COUT, int KS, bool RELU, bool SKIP, bool MASK = false>
__global__ __launch_bounds__(kConvThreads) void conv_kernel(ConvArgs a) {
  constexpr int HALO = KS / 2;
  constexpr int RB = kBand + 2 * HALO;
  constexpr int CS = RB * kWP;  // channel stride in LDS
  constexpr int S = ksteps(CIN, KS);
  constexpr int MT = mtiles(COUT);
  constexpr int NW = MT * S * 64;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* tile = lds;            // [CIN][RB][kWP]
  float* wl = lds + CIN * CS;   // [MT][S/4][64][4]

  const int img = blockIdx.x >> 1, band = blockIdx.x & 1, net = blockIdx.y;
  const int y0 = band * kBand;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, q = lane >> 4;
  const float* blob = a.packed[net];

  // ---- stage weights and the input band (zero halo) into LDS
  {
    const float4* src = reinterpret_cast<const float4*>(blob + a.woff);
    float4* dst = reinterpret_cast<float4*>(wl);
    for (int i = tid; i < NW / 4; i += kConvThreads) dst[i] = src[i];
    const float* in = a.in + net * a.in_net_stride + static_cast<size_t>(img) * CIN * kImg * kImg;
    for (int i = tid; i < CIN * RB * 10; i += kConvThreads) {
      const int ci = i / (RB * 10), rem = i % (RB * 10), rr = rem / 10, x4 = rem % 10;
      const int y = y0 - HALO + rr;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (x4 >= 1 && x4 <= 8 && y >= 0 && y < kImg)
        v = *reinterpret_cast<const float4*>(in + (ci * kImg + y) * kImg + 4 * (x4 - 1));
      *reinterpret_cast<float4*>(tile + ci * CS + rr * kWP + 4 * x4) = v;
    }
  }
  __syncthreads();

  // ---- each wave: 2 output rows x 2 half-rows (4 pixel tiles of 16) x MT channel tiles
  const int r0 = 2 * wave;
  f32x4 acc[MT][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const f32x4 b = *reinterpret_cast<const f32x4*>(blob + a.boff + 16 * mt + 4 * q);
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) acc[mt][pt] = b;
  }
  int base[4];
#pragma unroll
  for (int pt = 0; pt < 4; ++pt)
    base[pt] = (CIN == 1 ? 0 : q * CS) + (r0 + (pt >> 1)) * kWP + (4 - HALO) + 16 * (pt & 1) + j;
  int toff[S];  // CIN == 1: the tap (hence the offset) depends on the lane quarter
  if (CIN == 1) {
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const int t = 4 * s + q;
      toff[s] = t < KS * KS ? (t / KS) * kWP + (t % KS) : 0;  // zero weight there anyway
    }
  }
#pragma unroll
  for (int s4 = 0; s4 < S / 4; ++s4) {
    f32x4 a4[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
      a4[mt] = *reinterpret_cast<const f32x4*>(wl + ((mt * (S / 4) + s4) * 64 + lane) * 4);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int s = 4 * s4 + ks;
      int off;
      if (CIN == 1) {
        off = toff[s];
      } else {
        constexpr int G = CIN / 4;
        const int tap = s / G, cg = s % G;
        if (tap >= KS * KS) continue;  // padding k-steps carry zero weights
        off = 4 * cg * CS + (tap / KS) * kWP + (tap % KS);
      }
#pragma unroll
      for (int pt = 0; pt < 4; ++pt) {
        const float b = tile[base[pt] + off];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          acc[mt][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[mt][ks], b, acc[mt][pt], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: (+skip) (ReLU) store; lane (j, q) reg r -> channel 16mt + 4q + r, pixel j
  const size_t obase = (static_cast<size_t>(net) * a.N + img) * COUT * kImg * kImg;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int pt = 0; pt < 4; ++pt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ch = 16 * mt + 4 * q + r;
        if (ch < COUT) {
          const size_t o = obase + (static_cast<size_t>(ch) * kImg + (y0 + r0 + (pt >> 1))) * kImg + 16 * (pt & 1) + j;
          float v = acc[mt][pt][r];
          if (SKIP) v += a.skip[o];
          if (RELU) v = fmaxf(v, 0.f);
          if (MASK) v = a.mask[o] > 0.f ? v : 0.f;
          a.out[o] = v;
        }
      }
}

// ---- 8192 -> 64 linear, split-K partial sums ----------------------------------------------
struct FcArgs {
  const float* packed[kMaxNets];
  const float* act;   // (nets, N, 8192)
  float* partial;     // (nets, kFcSplit, N, 64)
  float* feat;        // (nets, N, 64)
  int* range_flag;    // f16x3 path: OR-ed with 1 when an activation leaves the split range
  int N;
};

__global__ __launch_bounds__(256) void fc_partial_kernel(FcArgs a) {
  constexpr Layout L = layout();
  const int tile = blockIdx.x, split = blockIdx.y, net = blockIdx.z;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;  // wave = 16-output tile
  const int j = lane & 15, q = lane >> 4;
  const int img = min(tile * 16 + j, a.N - 1);
  constexpr int KS = kFcK / kFcSplit;  // 512 per split
  const float* W = a.packed[net] + L.fcw + static_cast<size_t>(16 * wave + j) * kFcK + split * KS + 4 * q;
  const float* X = a.act + (static_cast<size_t>(net) * a.N + img) * kFcK + split * KS + 4 * q;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (int k = 0; k < KS; k += 16) {
    // both operands take k = k0 + 4q + e for element e: the same permutation of K on A and B
    const f32x4 w = *reinterpret_cast<const f32x4*>(W + k);
    const f32x4 x = *reinterpret_cast<const f32x4*>(X + k);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[0], x[0], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[1], x[1], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[2], x[2], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w[3], x[3], acc1, 0, 0, 0);
  }
  if (tile * 16 + j < a.N) {
    float* p = a.partial + ((static_cast<size_t>(net) * kFcSplit + split) * a.N + tile * 16 + j) * kFeat + 16 * wave + 4 * q;
    *reinterpret_cast<f32x4*>(p) = acc0 + acc1;  // lane (j, q) reg r -> output 16*wave + 4q + r of image j
  }
}

// The same split-K partial sums with f16x3 products (v_mfma_f32_32x32x16_f16): a wave owns 32
// images x 64 outputs of one K slice; the activations are split on load (8 consecutive k per
// lane = one B fragment), the weights come pre-split in fragment order (hfc section).  The f32
// version above runs at a quarter of the f32-MFMA peak; this one is bound by reading the
// activations once (134 MB per 2048 images x 2 encoders).
__global__ __launch_bounds__(256) void fc_partial_f16x3_kernel(FcArgs a) {
  constexpr Layout L = layout();
  constexpr int KSTEPS = kFcK / kFcSplit / 16;  // 32
  const int split = blockIdx.y, net = blockIdx.z;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int img0 = (blockIdx.x * 4 + wave) * 32;
  if (img0 >= a.N) return;
  const int img = min(img0 + j, a.N - 1);
  const float* X = a.act + (static_cast<size_t>(net) * a.N + img) * kFcK + split * (kFcK / kFcSplit) + 8 * h;
  const unsigned char* W = reinterpret_cast<const unsigned char*>(a.packed[net] + L.hfc) +
                           static_cast<size_t>(split) * KSTEPS * 4096 + lane * 16;
  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
  float amax = 0.f;
  // rolling requests (the loop is fully unrolled: every buffer index is static): activations four k-steps ahead (HBM), weight
  // fragments two ahead (L2).  Left to the compiler the fragments were requested right before their MFMAs: an L2 round
  // trip on every k-step of every wave.
  f32x4 xb[4][2];
  half8 wb[2][4];
  auto load_x = [&](int ks, int b) {
    xb[b][0] = *reinterpret_cast<const f32x4*>(X + 16 * ks);
    xb[b][1] = *reinterpret_cast<const f32x4*>(X + 16 * ks + 4);
  };
  auto load_w = [&](int ks, int b) {
    const unsigned char* wp = W + static_cast<size_t>(ks) * 4096;
#pragma unroll
    for (int f = 0; f < 4; ++f) wb[b][f] = *reinterpret_cast<const half8*>(wp + 1024 * f);  // tile 0 hi, lo, tile 1 hi, lo
  };
  load_x(0, 0);
  load_w(0, 0);
  load_x(1, 1);
  load_w(1, 1);
  load_x(2, 2);
  load_x(3, 3);
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) {
    const f32x4 x0 = xb[ks & 3][0], x1 = xb[ks & 3][1];
    const float xv[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
    u32x4 hv, lv;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const float v0 = xv[2 * p], v1 = xv[2 * p + 1];
      amax = fmaxf(amax, fmaxf(fabsf(v0), fabsf(v1)));
      const f32x2 xs = {v0, v1};
      const half2v hh = __builtin_convertvector(xs, half2v);  // round to nearest even
      const f32x2 hf = {static_cast<float>(hh[0]), static_cast<float>(hh[1])};
      const f32x2 r = xs - hf;
      const half2v ll = __builtin_convertvector(r, half2v);
      hv[p] = __builtin_bit_cast(unsigned, hh);
      lv[p] = __builtin_bit_cast(unsigned, ll);
    }
    if (ks + 4 < KSTEPS) load_x(ks + 4, ks & 3);
    const half8 bhi = __builtin_bit_cast(half8, hv), blo = __builtin_bit_cast(half8, lv);
    const half8 a0hi = wb[ks & 1][0], a0lo = wb[ks & 1][1], a1hi = wb[ks & 1][2], a1lo = wb[ks & 1][3];
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0hi, bhi, acc0, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0hi, blo, acc0, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0lo, bhi, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1hi, bhi, acc1, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1hi, blo, acc1, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1lo, bhi, acc1, 0, 0, 0);
    if (ks + 2 < KSTEPS) load_w(ks + 2, ks & 1);
    __builtin_amdgcn_sched_barrier(0);  // left alone the scheduler sinks every request to just before its use
  }
  if (a.range_flag != nullptr && !(amax < 65504.0f)) atomicOr(a.range_flag, 1);
  if (img0 + j < a.N) {
    // lane (image j, h), register r of tile t -> output 32 t + (r & 3) + 8 (r >> 2) + 4 h
    float* p = a.partial + ((static_cast<size_t>(net) * kFcSplit + split) * a.N + img0 + j) * kFeat + 4 * h;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 v0 = {acc0[4 * g] * kWInv, acc0[4 * g + 1] * kWInv, acc0[4 * g + 2] * kWInv, acc0[4 * g + 3] * kWInv};
      const f32x4 v1 = {acc1[4 * g] * kWInv, acc1[4 * g + 1] * kWInv, acc1[4 * g + 2] * kWInv, acc1[4 * g + 3] * kWInv};
      *reinterpret_cast<f32x4*>(p + 8 * g) = v0;
      *reinterpret_cast<f32x4*>(p + 32 + 8 * g) = v1;
    }
  }
}

// one wave per (image, net): bias + ReLU, then ResLinear(64).  SPAN: the linear layer's input is
// the 64 values of the two spanning average pools over the 2-channel map in `act`
// (push_models/layers.py:43-65: pool_h = mean over all rows x 2 columns -> [c][16], pool_w = mean
// over 2 rows x all columns -> [c][16], concatenated), computed here instead of the split-K sums.
template <bool SPAN>
__global__ __launch_bounds__(256) void fc_tail_kernel(FcArgs a) {
  constexpr Layout L = layout();
  const int lane = threadIdx.x & 63;
  const int img = blockIdx.x * 4 + (threadIdx.x >> 6), net = blockIdx.y;
  if (img >= a.N) return;
  const float* blob = a.packed[net];
  float h = blob[L.fcb + lane];
  if (SPAN) {
    const float* map = a.act + (static_cast<size_t>(net) * a.N + img) * kFcK;  // (8, 32, 32), channels 0-1 live
    const int c = (lane & 31) >> 4, g = lane & 15;
    const float* m = map + c * kImg * kImg;
    float sum = 0.f;
    if (lane < 32) {
      for (int y = 0; y < kImg; ++y) sum += m[y * kImg + 2 * g] + m[y * kImg + 2 * g + 1];
    } else {
      for (int x = 0; x < kImg; ++x) sum += m[(2 * g) * kImg + x] + m[(2 * g + 1) * kImg + x];
    }
    const float pooled = sum * (1.0f / 64.0f);
#pragma unroll 8
    for (int k = 0; k < kFeat; ++k) h += blob[L.fcw + lane * kFeat + k] * __shfl(pooled, k);
  } else {
    float ps[kFcSplit];  // the 16 partial sums are requested together and added in slice order
#pragma unroll
    for (int s = 0; s < kFcSplit; ++s) ps[s] = a.partial[((static_cast<size_t>(net) * kFcSplit + s) * a.N + img) * kFeat + lane];
#pragma unroll
    for (int s = 0; s < kFcSplit; ++s) h += ps[s];
  }
  // ResLinear(64): this lane's column of each 64 x 64 matrix is requested as ONE batch of 64 loads (eight at a time, each
  // batch a dependent L2 round trip on the chain, was most of this kernel's 19 us), lane k's value travels by v_readlane
  // (what __shfl(v, k) returns, without the LDS crossbar); same fused multiply-adds in the same order
  auto bcast = [](float v, int k) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), k)); };
  float w[kFeat];
#pragma unroll
  for (int k = 0; k < kFeat; ++k) w[k] = blob[L.r1t + k * kFeat + lane];
  const float r1b = blob[L.r1b + lane], r2b = blob[L.r2b + lane];
  h = fmaxf(h, 0.f);
  float t = r1b;
#pragma unroll
  for (int k = 0; k < kFeat; ++k) t = __builtin_fmaf(w[k], bcast(h, k), t);  // explicit chains: strict mode
#pragma unroll
  for (int k = 0; k < kFeat; ++k) w[k] = blob[L.r2t + k * kFeat + lane];
  t = fmaxf(t, 0.f);
  float y = r2b + h;
#pragma unroll
  for (int k = 0; k < kFeat; ++k) y = __builtin_fmaf(w[k], bcast(t, k), y);
  a.feat[(static_cast<size_t>(net) * a.N + img) * kFeat + lane] = fmaxf(y, 0.f);
}

template <int CIN, int COUT, int KS, bool RELU, bool SKIP, bool MASK = false>
int launch_conv(const ConvArgs& a, int nets, hipStream_t s) {
  constexpr int RB = kBand + 2 * (KS / 2);
  constexpr size_t lds = (static_cast<size_t>(CIN) * RB * kWP + mtiles(COUT) * ksteps(CIN, KS) * 64) * sizeof(float);
  static_assert(lds <= 160 * 1024, "conv tile + weights must fit LDS");
  auto k = conv_kernel<CIN, COUT, KS, RELU, SKIP, MASK>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    if (e != hipSuccess) return static_cast<int>(e);
  }
  k<<<dim3(2 * a.N, nets), kConvThreads, lds, s>>>(a);
  MMF_CHECK_LAUNCH();
  return 0;
}


// ---- K6 for the image encoder: training forward with every activation kept, backward data path
// (the forward conv kernel on transposed + flipped weights, ReLU masks in its epilogue) and the
// weight gradients as split-K MFMA correlations.  Exact fp32 (f32 MFMA), one encoder per call.
// Replaces torch autograd / MIOpen through the convolution stack of
//   /root/reference/crossmodal/door_models/layers.py:43-58  in  torchfilter.train.* (train_helpers.py:76-162);
// the 8192 -> 64 linear and the ResLinear behind it stay library GEMMs (rocBLAS through torch).
struct BwdLayout {
  int w4t, w3t, w2bt, w2at, zeros;  // dgrad weights in conv_kernel's fragment layout; 32 zero biases
  int h3t, h2bt, h2at;              // round 6: the 32-output-channel layers again as f16x3 A fragments [tap][kc][hi|lo][lane][8 halves]
  int total;
};
__host__ __device__ constexpr BwdLayout bwd_layout() {
  BwdLayout L{};
  int o = 0;
  L.w4t = o; o += conv_w_floats(8, 16, 3);     // dgrad of conv 16->8: a convolution 8 -> 16
  L.w3t = o; o += conv_w_floats(16, 32, 3);    // dgrad of conv 32->16: 16 -> 32
  L.w2bt = o; o += conv_w_floats(32, 32, 3);
  L.w2at = o; o += conv_w_floats(32, 32, 3);
  L.zeros = o; o += 32;
  L.h3t = o; o += 9 * 1 * 2 * 64 * 8 / 2;      // 16 -> 32: one k-chunk of 16 input channels per tap
  L.h2bt = o; o += conv_h_floats(32);
  L.h2at = o; o += conv_h_floats(32);
  L.total = o;
  return L;
}

// dgrad weights: W'[ci_fwd][co_fwd][ky][kx] = W[co_fwd][ci_fwd][2 - ky][2 - kx]
__global__ void pack_convs_backward_kernel(MmfImageEncoderDesc d, float* __restrict__ out) {
  constexpr BwdLayout L = bwd_layout();
  const int offs[5] = {L.w4t, L.w3t, L.w2bt, L.w2at, L.zeros};
  const int cin_b[4] = {8, 16, 32, 32}, cout_b[4] = {16, 32, 32, 32};  // of the BACKWARD convolution
  const int src[4] = {4, 3, 2, 1};                                      // forward conv index in desc
  for (int q0 = blockIdx.x * blockDim.x + threadIdx.x; q0 < L.total; q0 += gridDim.x * blockDim.x) {
    float v = 0.f;
    if (q0 >= L.h3t) {
      // f16x3 A fragments of the backward convolutions with 32 output channels: element i of lane (row, half) in k-chunk kc
      // of a tap = W'[co_b = row][ci_b = 16 kc + 8 half + i][ky][kx] = W[ci_b][co_b][2 - ky][2 - kx] of the forward layer, x 2^8
      const int hoffs[4] = {L.h3t, L.h2bt, L.h2at, L.total};
      const int hcin[3] = {16, 32, 32}, hsrc[3] = {3, 2, 1};
      int c = 0;
      while (q0 >= hoffs[c + 1]) ++c;
      const int KC = hcin[c] / 16;
      unsigned short hb[2];
      for (int z = 0; z < 2; ++z) {
        const int he = 2 * (q0 - hoffs[c]) + z;
        const int i = he & 7, lane = (he >> 3) & 63, part = (he >> 9) & 1, rest = he >> 10;
        const int kc = rest % KC, tap = rest / KC, ky = tap / 3, kx = tap % 3;
        const int co_b = lane & 31, ci_b = 16 * kc + 8 * (lane >> 5) + i;
        const float w = kWScale * d.conv_w[hsrc[c]][((ci_b * 32 + co_b) * 3 + (2 - ky)) * 3 + (2 - kx)];
        const __half hi = __float2half_rn(w);
        hb[z] = part ? __half_as_ushort(__float2half_rn(w - __half2float(hi))) : __half_as_ushort(hi);
      }
      v = __uint_as_float(static_cast<unsigned>(hb[0]) | (static_cast<unsigned>(hb[1]) << 16));
    } else if (q0 < L.zeros) {
      int c = 0;
      while (q0 >= offs[c + 1]) ++c;
      const int e = q0 - offs[c];
      const int ksub = e & 3, lane = (e >> 2) & 63, rest = e >> 8;
      const int S4 = ksteps(cin_b[c], 3) / 4;
      const int s4 = rest % S4, mt = rest / S4;
      const int s = 4 * s4 + ksub, i = lane & 15, q = lane >> 4;
      const int co_b = 16 * mt + i;  // backward output channel = forward input channel
      int ci_b, ky, kx;              // backward input channel = forward output channel
      if (co_b < cout_b[c] && kdecode(cin_b[c], 3, s, q, &ci_b, &ky, &kx))
        v = d.conv_w[src[c]][((ci_b * cout_b[c] + co_b) * 3 + (2 - ky)) * 3 + (2 - kx)];
    }
    out[q0] = v;
  }
}

// dW[tap][co][ci] = sum over images, rows, pixels of g[n][co][y][x] * act[n][ci][y + ky - 1][x + kx - 1]
// as v_mfma_f32_32x32x2_f32: A = g (rows = co, k = pixel), B = shifted act (k = pixel, cols = ci).  Work
// unit = half an image (16 output rows): a wave owns two rows of the unit and keeps all nine taps' 32x32
// accumulators in registers; k-step s of a row pairs pixels (s, 16 + s), so a lane holds 16 consecutive
// pixels of g (reused by the nine taps) and 18 of each of the three act rows (reused by the three kx).
// The eight waves' accumulators meet in LDS in a fixed tree; ONE partial per workgroup, summed by the
// caller.  db[co] = sum of g comes from the same A registers (`partial_b`).
constexpr int kWgradTile = 9 * 32 * 32;
constexpr size_t kLdsWgrad = 4 * kWgradTile * sizeof(float);
static_assert(kLdsWgrad <= 160 * 1024, "weight-gradient reduction must fit LDS");

__device__ __forceinline__ void wgrad_load16(const float* __restrict__ p, bool ok, float* v) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 t = ok ? *reinterpret_cast<const float4*>(p + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
  }
}

// sum of the eight waves' `acc[T]` into wave 0's, fixed order ((0+4)+(2+6)) + ((1+5)+(3+7))
template <int T>
__device__ __forceinline__ void wgrad_tree(f32x16* acc, float* red, int wave, int lane) {
#pragma unroll
  for (int step = 4; step >= 1; step >>= 1) {
    if (wave >= step && wave < 2 * step) {
      float* slot = red + (wave - step) * kWgradTile;
#pragma unroll
      for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) slot[(t * 16 + r) * 64 + lane] = acc[t][r];
    }
    __syncthreads();
    if (wave < step) {
      const float* slot = red + wave * kWgradTile;
#pragma unroll
      for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] += slot[(t * 16 + r) * 64 + lane];
    }
    __syncthreads();
  }
}

__device__ __forceinline__ void wgrad_bias(float bsum, float* red, float* __restrict__ partial_b, int wave, int lane) {
  bsum += __shfl_xor(bsum, 32);
  if (lane < 32) red[wave * 32 + lane] = bsum;
  __syncthreads();
  if (wave == 0 && lane < 32) {
    float t = red[lane];
#pragma unroll
    for (int w = 1; w < 8; ++w) t += red[w * 32 + lane];
    partial_b[blockIdx.x * 32 + lane] = t;
  }
}

template <int CO, int CI>
__global__ __launch_bounds__(512) void conv_wgrad_kernel(const float* __restrict__ g, const float* __restrict__ act,
                                                         float* __restrict__ partial, float* __restrict__ partial_b, int N) {
  extern __shared__ float red[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, kk = lane >> 5;
  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;
  for (int u = blockIdx.x; u < 2 * N; u += gridDim.x) {
    const int n = u >> 1, y0 = 16 * (u & 1);
    const float* gn = g + (static_cast<size_t>(n) * CO + (i < CO ? i : 0)) * kImg * kImg + 16 * kk;
    const float* an = act + (static_cast<size_t>(n) * CI + (i < CI ? i : 0)) * kImg * kImg + 16 * kk;
#pragma unroll 1
    for (int y = y0 + wave; y < y0 + 16; y += 8) {
      float a[16];
      wgrad_load16(gn + y * kImg, i < CO, a);
#pragma unroll
      for (int s = 0; s < 16; ++s) bsum += a[s];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int yy = y + ky - 1;
        if (yy < 0 || yy >= kImg) continue;  // wave-uniform
        float b[18];                         // pixels 16 kk - 1 .. 16 kk + 16 of the act row
        wgrad_load16(an + yy * kImg, i < CI, b + 1);
        b[0] = (i < CI && kk == 1) ? an[yy * kImg - 1] : 0.f;
        b[17] = (i < CI && kk == 0) ? an[yy * kImg + 16] : 0.f;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int s = 0; s < 16; ++s)
            acc[3 * ky + kx] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s + kx], acc[3 * ky + kx], 0, 0, 0);
      }
    }
  }
  wgrad_tree<9>(acc, red, wave, lane);
  wgrad_bias(bsum, red, partial_b, wave, lane);
  if (wave == 0) {
    // lane (column ci = i, half kk), register r -> row co = (r & 3) + 8 (r >> 2) + 4 kk
    float* p = partial + static_cast<size_t>(blockIdx.x) * kWgradTile;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int r = 0; r < 16; ++r) p[(tap * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk) * 32 + i] = acc[tap][r];
  }
}

// The 5x5 stem's weight gradient, same scheme with the 25 taps as the B columns:
// dW1[co][t] = sum g1[n][co][y][x] * image[n][y + t / 5 - 2][x + t % 5 - 2]; one 32x32 accumulator
// (co x tap, 7 columns idle) per wave, partial [co][32] at the head of the workgroup's slot.
__global__ __launch_bounds__(512) void conv_wgrad_stem_kernel(const float* __restrict__ g, const float* __restrict__ images,
                                                              float* __restrict__ partial, float* __restrict__ partial_b, int N) {
  extern __shared__ float red[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, kk = lane >> 5;
  const int ty = i / 5 - 2, tx = i % 5 - 2;
  f32x16 acc[1];
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[0][r] = 0.f;
  float bsum = 0.f;
  for (int u = blockIdx.x; u < 2 * N; u += gridDim.x) {
    const int n = u >> 1, y0 = 16 * (u & 1);
    const float* gn = g + (static_cast<size_t>(n) * 32 + i) * kImg * kImg + 16 * kk;
    const float* im = images + static_cast<size_t>(n) * kImg * kImg;
    for (int y = y0 + wave; y < y0 + 16; y += 8) {
      const int yy = y + ty;
      const bool row_ok = i < 25 && yy >= 0 && yy < kImg;
      float a[16], b[16];
      wgrad_load16(gn + y * kImg, true, a);
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        bsum += a[s];
        const int x = 16 * kk + s + tx;
        b[s] = (row_ok && x >= 0 && x < kImg) ? im[yy * kImg + x] : 0.f;
      }
#pragma unroll
      for (int s = 0; s < 16; ++s) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc[0], 0, 0, 0);
    }
  }
  wgrad_tree<1>(acc, red, wave, lane);
  wgrad_bias(bsum, red, partial_b, wave, lane);
  if (wave == 0) {
    float* p = partial + static_cast<size_t>(blockIdx.x) * kWgradTile;
#pragma unroll
    for (int r = 0; r < 16; ++r) p[((r & 3) + 8 * (r >> 2) + 4 * kk) * 32 + i] = acc[0][r];
  }
}

constexpr int kWPh = 34;  // padded row of the f16 planes: 1 | 32 pixels | 1

#include "image_encoder_fused.inc"
#include "image_encoder_resident.inc"
#include "image_encoder_train_h.inc"

}  // namespace

extern "C" size_t mmf_image_encoder_floats(void) { return static_cast<size_t>(layout().total); }

extern "C" size_t mmf_image_encoder_workspace_bytes(int n_images, int n_nets) {
  if (n_images < 0 || n_nets < 1 || n_nets > kMaxNets) return 0;
  // three 32-channel activation tensors (ping, pong, skip: the exact-fp32 per-layer path; the resident kernel uses the
  // first 8 channels of the third for E) + FC partial sums
  const size_t act = static_cast<size_t>(n_nets) * n_images * 32 * kImg * kImg;
  return (3 * act + static_cast<size_t>(n_nets) * kFcSplit * n_images * kFeat) * sizeof(float);
}

extern "C" int mmf_pack_image_encoder(const MmfImageEncoderDesc* d, float* packed, void* stream) {
  if (!d || !packed || !d->fc_w || !d->fc_b) return MMF_EINVAL;
  for (int i = 0; i < 5; ++i)
    if (!d->conv_w[i] || !d->conv_b[i]) return MMF_EINVAL;
  for (int i = 0; i < 2; ++i)
    if (!d->res_w[i] || !d->res_b[i]) return MMF_EINVAL;
  const int total = layout().total;
  pack_encoder_kernel<<<(total + 255) / 256, 256, 0, static_cast<hipStream_t>(stream)>>>(*d, packed);
  MMF_CHECK_LAUNCH();
  return 0;
}

extern "C" int mmf_image_encoder(const float* const* packed, int n_nets, const float* images,
                                 float* feat, void* workspace, int32_t* range_flag, int precision,
                                 int variant, int N, void* stream) {
  if (!packed || !images || !feat || !workspace) return MMF_EINVAL;
  if (n_nets < 1 || n_nets > kMaxNets || N < 0) return MMF_EINVAL;
  if (precision != MMF_PREC_F32 && precision != MMF_PREC_F16X3 && precision != MMF_PREC_BF16) return MMF_EINVAL;
  const bool bf16 = precision == MMF_PREC_BF16;
  if (bf16) precision = MMF_PREC_F16X3;  // conv 16->8 and the linear tail (6 % of the MACs) stay f16x3
  if (variant != MMF_ENCODER_DEFAULT && variant != MMF_ENCODER_SPANNING_POOL) return MMF_EINVAL;
  if (N == 0) return 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  constexpr Layout L = layout();
  const size_t act = static_cast<size_t>(n_nets) * N * 32 * kImg * kImg;
  float* bufA = static_cast<float*>(workspace);
  float* bufB = bufA + act;
  float* bufC = bufB + act;
  float* partial = bufC + act;

  ConvArgs c{};
  for (int i = 0; i < n_nets; ++i) {
    if (!packed[i]) return MMF_EINVAL;
    c.packed[i] = packed[i];
  }
  c.N = N;
  int rc;
  if (precision == MMF_PREC_F16X3) {
    // fused path (image_encoder_fused.inc): image -> B (bufA), B -> E (bufC): D never reaches HBM
    FusedArgs fa{};
    for (int i = 0; i < n_nets; ++i) fa.packed[i] = packed[i];
    fa.images = images; fa.N = N; fa.range_flag = range_flag;
    fa.out_e = bufC;
    if ((rc = launch_resident(fa, n_nets, bf16, s))) return rc;
    bufB = bufC;  // the linear tail reads E
  } else {
  // conv 1 -> 32, k5, ReLU                      images -> A
  c.in = images; c.in_net_stride = 0; c.skip = nullptr; c.out = bufA; c.woff = L.w1; c.boff = L.b1;
  if ((rc = launch_conv<1, 32, 5, true, false>(c, n_nets, s))) return rc;
  // ResConv block1: conv 32 -> 32, ReLU          A -> B
  c.in = bufA; c.in_net_stride = static_cast<long long>(N) * 32 * kImg * kImg; c.out = bufB;
  c.woff = L.w2a; c.boff = L.b2a;
  if ((rc = launch_conv<32, 32, 3, true, false>(c, n_nets, s))) return rc;
  // ResConv block2: conv 32 -> 32, + A, ReLU     B -> C
  c.in = bufB; c.skip = bufA; c.out = bufC; c.woff = L.w2b; c.boff = L.b2b;
  if ((rc = launch_conv<32, 32, 3, true, true>(c, n_nets, s))) return rc;
  // conv 32 -> 16, ReLU                          C -> A
  c.in = bufC; c.skip = nullptr; c.out = bufA; c.woff = L.w3; c.boff = L.b3;
  if ((rc = launch_conv<32, 16, 3, true, false>(c, n_nets, s))) return rc;
  // conv 16 -> 8 (no ReLU before Flatten)        A -> B
  c.in = bufA; c.in_net_stride = static_cast<long long>(N) * 16 * kImg * kImg; c.out = bufB;
  c.woff = L.w4; c.boff = L.b4;
  if ((rc = launch_conv<16, 8, 3, false, false>(c, n_nets, s))) return rc;
  }
  // Linear 8192 -> 64 (split-K partials), then bias + ReLU + ResLinear
  FcArgs f{};
  for (int i = 0; i < n_nets; ++i) f.packed[i] = packed[i];
  f.act = bufB; f.partial = partial; f.feat = feat; f.N = N;
  if (variant == MMF_ENCODER_SPANNING_POOL) {
    fc_tail_kernel<true><<<dim3((N + 3) / 4, n_nets), 256, 0, s>>>(f);
    MMF_CHECK_LAUNCH();
    return 0;
  }
  if (precision == MMF_PREC_F16X3) {
    f.range_flag = range_flag;
    fc_partial_f16x3_kernel<<<dim3((N + 127) / 128, kFcSplit, n_nets), 256, 0, s>>>(f);
  } else {
    fc_partial_kernel<<<dim3((N + 15) / 16, kFcSplit, n_nets), 256, 0, s>>>(f);
  }
  MMF_CHECK_LAUNCH();
  fc_tail_kernel<false><<<dim3((N + 3) / 4, n_nets), 256, 0, s>>>(f);
  MMF_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------ K6: image-encoder training
extern "C" size_t mmf_image_convs_backward_floats(void) { return static_cast<size_t>(bwd_layout().total); }

extern "C" int mmf_pack_image_convs_backward(const MmfImageEncoderDesc* d, float* packed, void* stream) {
  if (!d || !packed) return MMF_EINVAL;
  for (int i = 1; i < 5; ++i)
    if (!d->conv_w[i]) return MMF_EINVAL;
  if (d->variant != MMF_ENCODER_DEFAULT) return MMF_EINVAL;
  const int total = bwd_layout().total;
  pack_convs_backward_kernel<<<(total + 255) / 256, 256, 0, static_cast<hipStream_t>(stream)>>>(*d, packed);
  MMF_CHECK_LAUNCH();
  return 0;
}

extern "C" int mmf_image_convs_train_forward(const float* packed, const float* images, float* a1, float* h,
                                             float* a2, float* a3, float* a4, int32_t* range_flag, int precision,
                                             int N, void* stream) {
  if (!packed || !images || !a1 || !h || !a2 || !a3 || !a4 || N < 0) return MMF_EINVAL;
  if (precision != MMF_PREC_F32 && precision != MMF_PREC_F16X3 && precision != MMF_PREC_BF16) return MMF_EINVAL;
  if (N == 0) return 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  constexpr Layout L = layout();
  ConvArgs c{};
  c.packed[0] = packed;
  c.N = N;
  int rc;
  if (precision != MMF_PREC_F32) {
    // round 6: the inference path's resident kernel with every activation the backward reads kept on the way (fp32,
    // (N, C, 32, 32)): one launch instead of five per-layer kernels with the activations through HBM in between
    // (the reference-sized training step spent a tenth of its time in those: profiles/r05/train_refsize_kernel_stats.csv)
    FusedArgs fa{};
    fa.packed[0] = packed;
    fa.images = images; fa.N = N; fa.range_flag = range_flag;
    fa.out_e = a4; fa.keep_a1 = a1; fa.keep_h = h; fa.keep_a2 = a2; fa.keep_a3 = a3;
    return launch_resident(fa, 1, precision == MMF_PREC_BF16, s);
  }
  // the 5x5 stem (3 % of the MACs, one input channel)
  c.in = images; c.in_net_stride = 0; c.out = a1; c.woff = L.w1; c.boff = L.b1;
  if ((rc = launch_conv<1, 32, 5, true, false>(c, 1, s))) return rc;
  c.in = a1; c.out = h; c.woff = L.w2a; c.boff = L.b2a;
  if ((rc = launch_conv<32, 32, 3, true, false>(c, 1, s))) return rc;
  c.in = h; c.skip = a1; c.out = a2; c.woff = L.w2b; c.boff = L.b2b;
  if ((rc = launch_conv<32, 32, 3, true, true>(c, 1, s))) return rc;
  c.in = a2; c.skip = nullptr; c.out = a3; c.woff = L.w3; c.boff = L.b3;
  if ((rc = launch_conv<32, 16, 3, true, false>(c, 1, s))) return rc;
  c.in = a3; c.out = a4; c.woff = L.w4; c.boff = L.b4;
  return launch_conv<16, 8, 3, false, false>(c, 1, s);
}

extern "C" int mmf_image_convs_train_backward(const float* packed_bwd, const float* a1, const float* h,
                                              const float* a2, const float* a3, const float* g_a4, float* g1,
                                              float* gh, float* g2, float* g3, int N, void* stream) {
  if (!packed_bwd || !a1 || !h || !a2 || !a3 || !g_a4 || !g1 || !gh || !g2 || !g3 || N < 0) return MMF_EINVAL;
  if (N == 0) return 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  constexpr BwdLayout B = bwd_layout();
  ConvArgs c{};
  c.packed[0] = packed_bwd;
  c.N = N;
  c.boff = B.zeros;
  int rc;
  // g3 = dgrad(conv 16->8)(g_a4) where a3 > 0
  c.in = g_a4; c.mask = a3; c.out = g3; c.woff = B.w4t;
  if ((rc = launch_conv<8, 16, 3, false, false, true>(c, 1, s))) return rc;
  // g2 = dgrad(conv 32->16)(g3) where a2 > 0
  c.in = g3; c.mask = a2; c.out = g2; c.woff = B.w3t;
  if ((rc = launch_conv<16, 32, 3, false, false, true>(c, 1, s))) return rc;
  // gh = dgrad(ResConv block2)(g2) where h > 0
  c.in = g2; c.mask = h; c.out = gh; c.woff = B.w2bt;
  if ((rc = launch_conv<32, 32, 3, false, false, true>(c, 1, s))) return rc;
  // g1 = (g2 + dgrad(ResConv block1)(gh)) where a1 > 0
  c.in = gh; c.skip = g2; c.mask = a1; c.out = g1; c.woff = B.w2at;
  return launch_conv<32, 32, 3, false, true, true>(c, 1, s);
}

extern "C" int mmf_image_convs_train_backward_h(const float* packed_bwd, const float* a1, const float* h,
                                                const float* a2, const float* a3, const float* g_a4, float* g1,
                                                float* gh, float* g2, float* g3, float* scratch, int N, void* stream) {
  if (!packed_bwd || !a1 || !h || !a2 || !a3 || !g_a4 || !g1 || !gh || !g2 || !g3 || !scratch || N < 0) return MMF_EINVAL;
  if (N == 0) return 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  constexpr BwdLayout B = bwd_layout();
  ConvArgs c{};
  c.packed[0] = packed_bwd;
  c.N = N;
  c.boff = B.zeros;
  int rc;
  if (hipMemsetAsync(scratch, 0, 4 * sizeof(float), s) != hipSuccess) return MMF_EINVAL;
  // scratch: max |g3|, max |g2|, max |gh|, max |g_a4| -- the scales of the layers' operand splits here AND of
  // mmf_conv_weight_grads_h afterwards (the caller hands the words on); each dgrad launch leaves the next one's
  unsigned* mx = reinterpret_cast<unsigned*>(scratch);
  auto absmax = [&](const float* x, size_t n, int slot) {
    int blocks = static_cast<int>((n / 4 + 255) / 256);
    if (blocks > 512) blocks = 512;
    absmax_kernel<<<blocks, 256, 0, s>>>(x, n, mx + slot);
  };
  const size_t plane = static_cast<size_t>(N) * kImg * kImg;
  absmax(g_a4, 8 * plane, 3);
  // g3 = dgrad(conv 16->8)(g_a4) where a3 > 0: 8 input channels, exact fp32 (4 % of the backward's MACs)
  c.in = g_a4; c.mask = a3; c.out = g3; c.woff = B.w4t;
  if ((rc = launch_conv<8, 16, 3, false, false, true>(c, 1, s))) return rc;
  absmax(g3, 16 * plane, 0);
  DgradHArgs d{};
  d.packed = packed_bwd; d.N = N;
  // g2 = dgrad(conv 32->16)(g3) where a2 > 0
  d.in = g3; d.in_absmax = scratch; d.skip = nullptr; d.mask = a2; d.out = g2; d.out_absmax = mx + 1; d.hoff = B.h3t;
  if ((rc = launch_dgrad_h<16, false>(d, s))) return rc;
  // gh = dgrad(ResConv block2)(g2) where h > 0
  d.in = g2; d.in_absmax = scratch + 1; d.mask = h; d.out = gh; d.out_absmax = mx + 2; d.hoff = B.h2bt;
  if ((rc = launch_dgrad_h<32, false>(d, s))) return rc;
  // g1 = (g2 + dgrad(ResConv block1)(gh)) where a1 > 0
  d.in = gh; d.in_absmax = scratch + 2; d.skip = g2; d.mask = a1; d.out = g1; d.out_absmax = nullptr; d.hoff = B.h2at;
  rc = launch_dgrad_h<32, true>(d, s);
  MMF_CHECK_LAUNCH();
  return rc;
}

using WgradKernel = void (*)(const float*, const float*, float*, float*, int);

static int launch_wgrad(WgradKernel k, const float* g, const float* act, float* partial, float* partial_b, int N,
                        int n_blocks, hipStream_t s) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     static_cast<int>(kLdsWgrad));
  if (e != hipSuccess) return static_cast<int>(e);
  k<<<n_blocks, 512, kLdsWgrad, s>>>(g, act, partial, partial_b, N);
  MMF_CHECK_LAUNCH();
  return 0;
}

// partial slots -> the layer's gradients in nn.Conv2d layout, slots in ascending order.  One thread per value of a slot,
// in the SLOT's order (neighbouring threads read neighbouring words of every slot); the permutation is in the store.
__global__ __launch_bounds__(256) void conv_wgrad_finalize_kernel(const float* __restrict__ partial, const float* __restrict__ partial_b,
                                                                int n_blocks, int co, int ci, float* __restrict__ dw,
                                                                float* __restrict__ db) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int n_slot = ci == 1 ? 1024 : 9216;
  if (e < n_slot) {
    int dst;
    if (ci == 1) {                                   // [co 32][tap 32], 25 taps live
      const int o = e >> 5, tap = e & 31;
      if (tap >= 25) return;
      dst = o * 25 + tap;
    } else {                                         // [tap 9][co 32][ci 32]
      const int tap = e >> 10, o = (e >> 5) & 31, c = e & 31;
      if (o >= co || c >= ci) return;
      dst = (o * ci + c) * 9 + tap;
    }
    float s = 0.f;
#pragma unroll 8
    for (int b = 0; b < n_blocks; ++b) s = __fadd_rn(s, partial[static_cast<size_t>(b) * 9216 + e]);
    dw[dst] = s;
  } else if (e < n_slot + co && db) {
    const int o = e - n_slot;
    float s = 0.f;
#pragma unroll 8
    for (int b = 0; b < n_blocks; ++b) s = __fadd_rn(s, partial_b[b * 32 + o]);
    db[o] = s;
  }
}

extern "C" int mmf_conv_weight_grads_h(const float* g, const float* act, const float* g_absmax, float* partial, float* partial_b,
                                       int32_t* range_flag, int N, int co, int ci, int n_blocks, float* dw, float* db, void* stream) {
  if (!g || !act || !g_absmax || !partial || !partial_b || N < 0 || n_blocks < 1) return MMF_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  void (*k)(const float*, const float*, float*, float*, int, const float*, int*) = nullptr;
  if (co == 32 && ci == 32) k = conv_wgrad_h_kernel<32, 32>;
  else if (co == 16 && ci == 32) k = conv_wgrad_h_kernel<16, 32>;
  else if (co == 8 && ci == 16) k = conv_wgrad_h_kernel<8, 16>;
  else return MMF_EINVAL;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     static_cast<int>(kLdsWgrad));
  if (e != hipSuccess) return static_cast<int>(e);
  k<<<n_blocks, 512, kLdsWgrad, s>>>(g, act, partial, partial_b, N, g_absmax, range_flag);
  MMF_CHECK_LAUNCH();
  if (!dw) return 0;
  conv_wgrad_finalize_kernel<<<(9216 + co + 255) / 256, 256, 0, s>>>(partial, partial_b, n_blocks, co, ci, dw, db);
  MMF_CHECK_LAUNCH();
  return 0;
}

extern "C" int mmf_conv_weight_grads(const float* g, const float* act, float* partial, float* partial_b, int N, int co,
                                     int ci, int n_blocks, float* dw, float* db, void* stream) {
  if (!g || !act || !partial || !partial_b || N < 0 || n_blocks < 1) return MMF_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  WgradKernel k = nullptr;
  if (co == 32 && ci == 32) k = conv_wgrad_kernel<32, 32>;
  else if (co == 16 && ci == 32) k = conv_wgrad_kernel<16, 32>;
  else if (co == 8 && ci == 16) k = conv_wgrad_kernel<8, 16>;
  else if (co == 32 && ci == 1) k = conv_wgrad_stem_kernel;
  else return MMF_EINVAL;
  const int rc = launch_wgrad(k, g, act, partial, partial_b, N, n_blocks, s);
  if (rc || !dw) return rc;
  const int n = (ci == 1 ? 1024 : 9216) + co;
  conv_wgrad_finalize_kernel<<<(n + 255) / 256, 256, 0, s>>>(partial, partial_b, n_blocks, co, ci, dw, db);
  MMF_CHECK_LAUNCH();
  return 0;
}
