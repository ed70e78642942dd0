// K6 for the per-trajectory networks (round 5): what is left of a training step's backward once the K7 programs run
// their own reverse programs (traj_program.hip) --
//   * the parameter gradients of a program's LINEARs, dW = dz^T x over the rows the two programs stashed, and
//   * the image encoder's 8192 -> 64 linear layer, forward and both backward products.
// The reference differentiates all of it with torch autograd (train_helpers.py:124-162 over door_models/layers.py:11-63,
// door_models/pf.py:30-62, crossmodal_pf.py:52-106): per nn.Linear one library GEMM forward, two backward and a column
// reduction.  Here every product is exact fp32 on v_mfma_f32_16x16x4_f32 (bit for bit a k-ordered fma chain) with a
// fixed contraction order, so a training step is reproducible run to run; rows are few (T*N), the point is launch count
// and no library heuristics on the path.
#include "mmf_common.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// ---------------------------------------------------------------------------------------------------------------
// dW[o][k] = sum_row dz[row][o] x[row][k].  MFMA roles: M = outputs, N = input columns, K = rows: lane (i, q) supplies
// A = dz[row 4 s + q][16 ot + i] and B = x[row 4 s + q][16 ct + i], its accumulators are dW[16 ot + 4 q + r][16 ct + i].
// One workgroup per (descriptor, row slice); wave w owns output tiles w, w + 4.
constexpr int kGradWaves = 4;
constexpr int kMaxColTiles = 8;  // x_dim <= 128

__global__ __launch_bounds__(kGradWaves * MMF_WAVE) void traj_weight_grad_kernel(
    const MmfTrajGradDesc* __restrict__ desc, const float* __restrict__ stash, int stash_ld, const float* __restrict__ dz,
    int dz_ld, float* __restrict__ out, int n_grads, int rows_per_slice, int R) {
  const MmfTrajGradDesc d = desc[blockIdx.x];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, q = lane >> 4;
  const int r0 = blockIdx.y * rows_per_slice, r1 = min(R, r0 + rows_per_slice);
  float* dst = out + static_cast<size_t>(blockIdx.y) * n_grads;
  const int col_tiles = (d.x_dim + 15) >> 4, out_tiles = (d.out_dim + 15) >> 4;
  for (int ot = wave; ot < out_tiles; ot += kGradWaves) {
    f32x4 acc[kMaxColTiles];
#pragma unroll
    for (int ct = 0; ct < kMaxColTiles; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    const bool o_ok = 16 * ot + i < d.out_dim;
    const float* pa = dz + d.dz_col + 16 * ot + i;
    const float* pb = stash + d.x_col + i;
#pragma unroll 2
    for (int r = r0; r < r1; r += 4) {
      const int row = r + q;
      const bool ok = row < r1;
      const float a = (ok && o_ok) ? pa[static_cast<size_t>(row) * dz_ld] : 0.f;
      bsum += a;
      float b[kMaxColTiles];
#pragma unroll
      for (int ct = 0; ct < kMaxColTiles; ++ct)
        b[ct] = (ct < col_tiles && ok && 16 * ct + i < d.x_dim) ? pb[static_cast<size_t>(row) * stash_ld + 16 * ct] : 0.f;
#pragma unroll
      for (int ct = 0; ct < kMaxColTiles; ++ct)
        if (ct < col_tiles) acc[ct] = mfma4(a, b[ct], acc[ct]);
    }
#pragma unroll
    for (int ct = 0; ct < kMaxColTiles; ++ct)
      if (ct < col_tiles && 16 * ct + i < d.x_dim)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = 16 * ot + 4 * q + r;
          if (o < d.out_dim) dst[d.grad_off + static_cast<size_t>(o) * d.grad_ld + 16 * ct + i] = acc[ct][r];
        }
    if (d.bias_off >= 0) {  // rows q, q + 4, ..: the four lane groups in the order q = 0..3
      const float s1 = __shfl(bsum, i + 16), s2 = __shfl(bsum, i + 32), s3 = __shfl(bsum, i + 48);
      if (q == 0 && o_ok) dst[d.bias_off + 16 * ot + i] = __fadd_rn(__fadd_rn(__fadd_rn(bsum, s1), s2), s3);
    }
  }
}

__global__ void sum_slices_kernel(const float* __restrict__ partials, float* __restrict__ out, int n, int n_slices) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  float s = partials[j];
  for (int k = 1; k < n_slices; ++k) s = __fadd_rn(s, partials[static_cast<size_t>(k) * n + j]);
  out[j] = s;
}

// ---------------------------------------------------------------------------------------------------------------
// y (R, 64) = x (R, K) w^T + b.  grid = (16-row tiles, K chunks): a workgroup's four waves split its chunk, meet in LDS in
// wave order and write ONE partial; fc64_sum_kernel adds the chunks in order and the bias.  M = outputs, N = rows,
// contraction over k: lane (i, q) loads the 16 bytes x[row i][16 g + 4 q ..] and w[16 mt + i][16 g + 4 q ..] and feeds
// them to four MFMA steps element by element (any bijection of k onto steps is the same dot product).  With R = 512 the
// product is 0.5 GFLOP on 16 MB of x: what matters is that no wave walks a long chain of dependent round trips.
constexpr int kFcWaves = 4;
constexpr int kFcChunk = 512;      // k per workgroup
constexpr int kFcGroups = kFcChunk / kFcWaves / 16;  // 16-k groups per wave: 8, all requested up front

__global__ __launch_bounds__(kFcWaves * MMF_WAVE) void fc64_forward_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                         float* __restrict__ partial, int R, int K) {
  __shared__ f32x4 red[kFcWaves - 1][4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, q = lane >> 4;
  const int row0 = blockIdx.x * 16;
  const int row = min(row0 + i, R - 1);
  const int k0 = blockIdx.y * kFcChunk + wave * (kFcChunk / kFcWaves) + 4 * q;
  const float* px = x + static_cast<size_t>(row) * K + k0;
  const float* pw = w + static_cast<size_t>(i) * K + k0;
  f32x4 b[kFcGroups], a[kFcGroups][4];
#pragma unroll
  for (int g = 0; g < kFcGroups; ++g) {
    b[g] = *reinterpret_cast<const f32x4*>(px + 16 * g);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) a[g][mt] = *reinterpret_cast<const f32x4*>(pw + static_cast<size_t>(16 * mt) * K + 16 * g);
  }
  f32x4 acc[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int g = 0; g < kFcGroups; ++g)
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[mt] = mfma4(a[g][mt][e], b[g][e], acc[mt]);
  if (wave > 0)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) red[wave - 1][mt][lane] = acc[mt];
  __syncthreads();
  if (wave == 0 && row0 + i < R) {
    float* dst = partial + (static_cast<size_t>(blockIdx.y) * R + row0 + i) * 64;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      f32x4 s = acc[mt];
#pragma unroll
      for (int k = 0; k < kFcWaves - 1; ++k) {
        const f32x4 v = red[k][mt][lane];
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] = __fadd_rn(s[e], v[e]);
      }
      *reinterpret_cast<f32x4*>(dst + 16 * mt + 4 * q) = s;
    }
  }
}

// y[j] = sum over chunks (ascending) of partial[chunk][j] + bias[j & 63]
__global__ void fc64_sum_kernel(const float* __restrict__ partial, const float* __restrict__ bias, float* __restrict__ y, int n,
                                int chunks) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  float s = partial[j];
  for (int k = 1; k < chunks; ++k) s = __fadd_rn(s, partial[static_cast<size_t>(k) * n + j]);
  y[j] = bias ? __fadd_rn(s, bias[j & 63]) : s;
}

// dx (R, K) = g (R, 64) w (64, K).  M = columns of dx, N = rows, contraction over the 64 outputs: lane (i, q) holds
// B = g[row i][4 s + q] for s = 0..15 (loaded once) and per column tile A = w[4 s + q][16 ct + i]; its accumulators are
// dx[row i][16 ct + 4 q + r] = one 16-byte store.  grid = (row tiles, column chunks); a wave walks column tiles.
__global__ __launch_bounds__(256) void fc64_backward_data_kernel(const float* __restrict__ g, const float* __restrict__ w,
                                                                float* __restrict__ dx, int R, int K, int cols_per_block) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, q = lane >> 4;
  const int row0 = blockIdx.x * 16;
  const int row = min(row0 + i, R - 1);
  float b[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) b[s] = g[static_cast<size_t>(row) * 64 + 4 * s + q];
  const int c0 = blockIdx.y * cols_per_block, c1 = min(K, c0 + cols_per_block);
  for (int ct = c0 + 16 * wave; ct < c1; ct += 16 * 4) {
    float a[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) a[s] = w[static_cast<size_t>(4 * s + q) * K + ct + i];
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = mfma4(a[s], b[s], acc);
    if (row0 + i < R) *reinterpret_cast<f32x4*>(dx + static_cast<size_t>(row0 + i) * K + ct + 4 * q) = acc;
  }
}

// dw (64, K) = g^T x.  M = outputs, N = columns, contraction over rows: lane (i, q) supplies A = g[row 4 s + q][16 mt + i]
// and B = x[row 4 s + q][16 ct + i].  A workgroup owns 32 columns (128 contiguous bytes of every row of x); its eight
// waves split the rows into eight runs and meet in LDS in wave order.  Block 0 also writes db.
constexpr int kFcwWaves = 8;

__global__ __launch_bounds__(kFcwWaves * MMF_WAVE) void fc64_backward_weights_kernel(const float* __restrict__ g,
                                                                                   const float* __restrict__ x,
                                                                                   float* __restrict__ dw, float* __restrict__ db,
                                                                                   int R, int K) {
  __shared__ f32x4 red[kFcwWaves - 1][8][64];   // 56 KB
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, q = lane >> 4;
  const int col0 = blockIdx.x * 32;
  const int run = ((R + 4 * kFcwWaves - 1) / (4 * kFcwWaves)) * 4;  // rows per wave, a multiple of 4
  const int r0 = wave * run, r1 = min(R, r0 + run);
  f32x4 acc[4][2];  // [mt][ct]
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) acc[mt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (int r = r0; r < r1; r += 4) {
    const int row = r + q;
    const bool ok = row < r1;
    float a[4], b[2];
#pragma unroll
    for (int t = 0; t < 4; ++t) a[t] = ok ? g[static_cast<size_t>(row) * 64 + 16 * t + i] : 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t) b[t] = ok ? x[static_cast<size_t>(row) * K + col0 + 16 * t + i] : 0.f;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) acc[mt][ct] = mfma4(a[mt], b[ct], acc[mt][ct]);
  }
  if (wave > 0)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) red[wave - 1][2 * mt + ct][lane] = acc[mt][ct];
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        f32x4 s = acc[mt][ct];
        for (int k = 0; k < kFcwWaves - 1; ++k) {
          const f32x4 v = red[k][2 * mt + ct][lane];
#pragma unroll
          for (int e = 0; e < 4; ++e) s[e] = __fadd_rn(s[e], v[e]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) dw[static_cast<size_t>(16 * mt + 4 * q + e) * K + col0 + 16 * ct + i] = s[e];
      }
  }
  if (db && blockIdx.x == gridDim.x - 1) {  // column sums of g: wave v adds rows v, v + 8, ..; the eight meet in order
    __syncthreads();
    float s = 0.f;
#pragma unroll 8
    for (int r = wave; r < R; r += kFcwWaves) s = __fadd_rn(s, g[static_cast<size_t>(r) * 64 + lane]);
    float* part = reinterpret_cast<float*>(red);
    part[wave * 64 + lane] = s;
    __syncthreads();
    if (wave == 0) {
      float t = part[lane];
      for (int v = 1; v < kFcwWaves; ++v) t = __fadd_rn(t, part[v * 64 + lane]);
      db[lane] = t;
    }
  }
}

}  // namespace

extern "C" int mmf_traj_weight_grads(const MmfTrajGradDesc* desc, int n_desc, const float* stash, int stash_ld,
                                     const float* dz, int dz_ld, float* grads, int n_grads, float* partials, int n_slices,
                                     int R, void* stream) {
  if (!desc || !stash || !dz || !grads || n_desc < 1 || n_grads < 1 || R < 0 || stash_ld < 1 || dz_ld < 1) return MMF_EINVAL;
  if (n_slices < 1 || n_slices > 64 || (n_slices > 1 && !partials)) return MMF_EINVAL;
  auto s = static_cast<hipStream_t>(stream);
  const int per = ((((R + n_slices - 1) / n_slices) + 3) / 4) * 4;  // rows per slice, a multiple of the MFMA's four
  traj_weight_grad_kernel<<<dim3(n_desc, n_slices), kGradWaves * MMF_WAVE, 0, s>>>(
      desc, stash, stash_ld, dz, dz_ld, n_slices > 1 ? partials : grads, n_grads, per > 0 ? per : 4, R);
  MMF_CHECK_LAUNCH();
  if (n_slices > 1) {
    sum_slices_kernel<<<(n_grads + 255) / 256, 256, 0, s>>>(partials, grads, n_grads, n_slices);
    MMF_CHECK_LAUNCH();
  }
  return 0;
}

extern "C" int mmf_fc64_train_forward(const float* x, const float* w, const float* b, float* y, float* partial, int R, int K,
                                      void* stream) {
  if (!x || !w || !y || !partial || R < 0 || K < kFcChunk || K % kFcChunk) return MMF_EINVAL;
  if (R == 0) return 0;
  auto s = static_cast<hipStream_t>(stream);
  const int chunks = K / kFcChunk;
  fc64_forward_kernel<<<dim3((R + 15) / 16, chunks), kFcWaves * MMF_WAVE, 0, s>>>(x, w, partial, R, K);
  MMF_CHECK_LAUNCH();
  fc64_sum_kernel<<<(R * 64 + 255) / 256, 256, 0, s>>>(partial, b, y, R * 64, chunks);
  MMF_CHECK_LAUNCH();
  return 0;
}

extern "C" int mmf_fc64_train_backward(const float* g, const float* x, const float* w, float* dx, float* dw, float* db,
                                       int R, int K, void* stream) {
  if (!g || !x || !w || !dw || R < 0 || K < kFcChunk || K % kFcChunk) return MMF_EINVAL;
  auto s = static_cast<hipStream_t>(stream);
  if (R == 0) {
    hipError_t e = hipMemsetAsync(dw, 0, sizeof(float) * 64 * static_cast<size_t>(K), s);
    if (e == hipSuccess && db) e = hipMemsetAsync(db, 0, sizeof(float) * 64, s);
    return e == hipSuccess ? 0 : static_cast<int>(e);
  }
  if (dx) {
    const int chunks = K >= 2048 ? 16 : 1;
    const int cols = ((K / chunks + 63) / 64) * 64;
    fc64_backward_data_kernel<<<dim3((R + 15) / 16, (K + cols - 1) / cols), 256, 0, s>>>(g, w, dx, R, K, cols);
    MMF_CHECK_LAUNCH();
  }
  fc64_backward_weights_kernel<<<K / 32, kFcwWaves * MMF_WAVE, 0, s>>>(g, x, dw, db, R, K);
  MMF_CHECK_LAUNCH();
  return 0;
}
