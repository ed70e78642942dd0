// K3: batched small-matrix EKF predict/correct for K sub-filters + crossmodal / unimodal
// fusion of their beliefs, one trajectory per lane, all d x d algebra in registers.
//
// Replaces torchfilter's ExtendedKalmanFilter._predict_step / _update_step specialised to the
// virtual-sensor case C = I (external dependency of the reference) and
//   /root/reference/crossmodal/base_models/crossmodal_kf.py:153-167  (crossmodal fusion)
//   /root/reference/crossmodal/base_models/utility.py:4-11           (weighted_average)
//   /root/reference/crossmodal/base_models/unimodal_kf.py:204-242    (information-form fusion)
// HBM bytes per trajectory-step (K=2, d=3): 2*(A 36 + mu- 12 + z 12 + T 36 + w 12 + Sigma 36 r/w 72
// + mu 12) + fused 48 = 432 B for ~1.5 kFLOP: launch-latency-bound in isolation.
#include "ekf_algebra.h"

namespace {

using namespace mmf_ekf;


template <int D>
__global__ __launch_bounds__(256) void ekf_step_kernel(
    const float* __restrict__ A, const float* __restrict__ mu_pred, const float* __restrict__ q_tril,
    const float* __restrict__ z, const float* __restrict__ r_tril, const float* __restrict__ fuse_w,
    float* __restrict__ mu, float* __restrict__ Sigma, float* __restrict__ mu_f,
    float* __restrict__ Sigma_f, int N, int K, int fusion, int feedback, const int32_t* __restrict__ feedback_gate) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  // a batch-global, data-dependent branch of the reference (door_models/crossmodal_kf.py:59-62: ANY blacked-out
  // frame in the batch -> the branch that skips the write-back) as a device word: no host round trip per step
  if (feedback_gate != nullptr && *feedback_gate == 0) feedback = 0;

  float mus[kMaxK][D], w[kMaxK][D];
  Mat<D> Ss[kMaxK];
#pragma unroll
  for (int k = 0; k < kMaxK; ++k) {
    if (k >= K) break;
    const size_t row = static_cast<size_t>(k) * N + n;
    const Mat<D> Ak = load_mat<D>(A + row * D * D);
    const Mat<D> S0 = load_mat<D>(Sigma + row * D * D);
    const Mat<D> L = load_mat<D>(q_tril + static_cast<size_t>(k) * D * D);
    const Mat<D> T = load_mat<D>(r_tril + row * D * D);
    float mp[D], zk[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
      mp[i] = mu_pred[row * D + i];
      zk[i] = z[row * D + i];
    }
    predict_correct<D>(Ak, S0, L, T, mp, zk, mus[k], Ss[k]);
    if (fusion == 1) {
#pragma unroll
      for (int i = 0; i < D; ++i) w[k][i] = fuse_w[row * D + i];
    }
  }

  float mf[D];
  Mat<D> Sf;
  fuse<D>(K, fusion, w, mus, Ss, mf, Sf);

  if (fusion != 0) {
#pragma unroll
    for (int i = 0; i < D; ++i) mu_f[static_cast<size_t>(n) * D + i] = mf[i];
    store_mat<D>(Sigma_f + static_cast<size_t>(n) * D * D, Sf);
  }
#pragma unroll
  for (int k = 0; k < kMaxK; ++k) {
    if (k >= K) break;
    const size_t row = static_cast<size_t>(k) * N + n;
    const bool fb = fusion != 0 && feedback != 0;
#pragma unroll
    for (int i = 0; i < D; ++i) mu[row * D + i] = fb ? mf[i] : mus[k][i];
    store_mat<D>(Sigma + row * D * D, fb ? Sf : Ss[k]);
  }
}

// K6 (K3 backward): reverse mode of one sub-filter's predict + correct, one row per lane.
// Forward (recomputed in registers):  AS = A S0;  Sp = AS A^T + L L^T;  Rm = T T^T;  Si = (Sp + Rm)^-1;
//   G = Sp Si;  mu = mp + G (z - mp);  S = (I - G) Sp.
// Reverse, given g_mu and g_S:
//   g(I-G) = g_S Sp^T;  gSp = (I-G)^T g_S;  gG = g_mu (z - mp)^T - g(I-G);  g_z = G^T g_mu;  g_mp = g_mu - g_z;
//   gSp += gG Si^T;  gSi = Sp^T gG;  gSinn = -Si^T gSi Si^T;  gSp += gSinn;  gT = (gSinn + gSinn^T) T;
//   gA = gSp^T AS + (gSp A) S0^T;  gS0 = A^T (gSp A).
template <int D>
__device__ __forceinline__ Mat<D> transpose(const Mat<D>& x) {
  Mat<D> r;
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) r.a[i][j] = x.a[j][i];
  return r;
}

template <int D>
__global__ __launch_bounds__(256) void ekf_step_backward_kernel(
    const float* __restrict__ A, const float* __restrict__ mu_pred, const float* __restrict__ q_tril,
    const float* __restrict__ z, const float* __restrict__ r_tril, const float* __restrict__ Sigma_in,
    const float* __restrict__ g_mu, const float* __restrict__ g_Sigma, float* __restrict__ g_A,
    float* __restrict__ g_mu_pred, float* __restrict__ g_z, float* __restrict__ g_r_tril,
    float* __restrict__ g_Sigma_in, int N, int K) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= K * N) return;
  const int k = row / N;
  const Mat<D> Ak = load_mat<D>(A + static_cast<size_t>(row) * D * D);
  const Mat<D> S0 = load_mat<D>(Sigma_in + static_cast<size_t>(row) * D * D);
  const Mat<D> L = load_mat<D>(q_tril + static_cast<size_t>(k) * D * D);
  const Mat<D> T = load_mat<D>(r_tril + static_cast<size_t>(row) * D * D);
  const Mat<D> AS = matmul<D>(Ak, S0);
  Mat<D> Sp = matmul_nt<D>(AS, Ak);
  const Mat<D> Q = matmul_nt<D>(L, L);
  const Mat<D> Rm = matmul_nt<D>(T, T);
  Mat<D> Sinn;
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) {
      Sp.a[i][j] += Q.a[i][j];
      Sinn.a[i][j] = Sp.a[i][j] + Rm.a[i][j];
    }
  const Mat<D> Si = inverse<D>(Sinn);
  const Mat<D> G = matmul<D>(Sp, Si);
  float innov[D], gm[D];
  Mat<D> ImG, gS;
#pragma unroll
  for (int i = 0; i < D; ++i) {
    innov[i] = z[static_cast<size_t>(row) * D + i] - mu_pred[static_cast<size_t>(row) * D + i];
    gm[i] = g_mu ? g_mu[static_cast<size_t>(row) * D + i] : 0.f;
#pragma unroll
    for (int j = 0; j < D; ++j) {
      ImG.a[i][j] = ((i == j) ? 1.f : 0.f) - G.a[i][j];
      gS.a[i][j] = g_Sigma ? g_Sigma[(static_cast<size_t>(row) * D + i) * D + j] : 0.f;
    }
  }
  // S = ImG Sp
  const Mat<D> gImG = matmul_nt<D>(gS, Sp);               // g_S Sp^T
  Mat<D> gSp = matmul<D>(transpose<D>(ImG), gS);          // ImG^T g_S
  // mu = mp + G innov
  Mat<D> gG;
  float gz[D];
#pragma unroll
  for (int i = 0; i < D; ++i) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < D; ++j) {
      gG.a[i][j] = gm[i] * innov[j] - gImG.a[i][j];
      s += G.a[j][i] * gm[j];
    }
    gz[i] = s;
  }
  // G = Sp Si
  const Mat<D> gSp2 = matmul_nt<D>(gG, Si);               // gG Si^T
  const Mat<D> gSi = matmul<D>(transpose<D>(Sp), gG);     // Sp^T gG
  // Si = Sinn^-1:  gSinn = -Si^T gSi Si^T
  const Mat<D> SiT = transpose<D>(Si);
  const Mat<D> tmp = matmul<D>(SiT, gSi);
  Mat<D> gSinn = matmul<D>(tmp, SiT);
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) {
      gSinn.a[i][j] = -gSinn.a[i][j];
      gSp.a[i][j] += gSp2.a[i][j] + gSinn.a[i][j];
    }
  // Rm = T T^T:  gT = (gSinn + gSinn^T) T
  Mat<D> sym;
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) sym.a[i][j] = gSinn.a[i][j] + gSinn.a[j][i];
  const Mat<D> gT = matmul<D>(sym, T);
  // Sp = AS A^T (+ Q):  gAS = gSp A;  gA = gSp^T AS;  AS = A S0:  gA += gAS S0^T;  gS0 = A^T gAS
  const Mat<D> gAS = matmul<D>(gSp, Ak);
  Mat<D> gA = matmul<D>(transpose<D>(gSp), AS);
  const Mat<D> gA2 = matmul_nt<D>(gAS, S0);
  const Mat<D> gS0 = matmul<D>(transpose<D>(Ak), gAS);
#pragma unroll
  for (int i = 0; i < D; ++i) {
#pragma unroll
    for (int j = 0; j < D; ++j) gA.a[i][j] += gA2.a[i][j];
    if (g_z) g_z[static_cast<size_t>(row) * D + i] = gz[i];
    if (g_mu_pred) g_mu_pred[static_cast<size_t>(row) * D + i] = gm[i] - gz[i];
  }
  if (g_A) store_mat<D>(g_A + static_cast<size_t>(row) * D * D, gA);
  if (g_r_tril) store_mat<D>(g_r_tril + static_cast<size_t>(row) * D * D, gT);
  if (g_Sigma_in) store_mat<D>(g_Sigma_in + static_cast<size_t>(row) * D * D, gS0);
}

// R11: fusion of K virtual sensors BEFORE a single EKF, one trajectory per lane.
//   mode 1  /root/reference/crossmodal/base_models/crossmodal_kf.py:291-359
//           mu = sum_k w_k z_k / (sum_k w_k + 1e-9);  Sigma = (prod_k prod_i w_ki) sum_k T_k T_k^T;
//           returns chol(Sigma)
//   mode 2  /root/reference/crossmodal/base_models/unimodal_kf.py:56-115, quirk Q5 included:
//           "precision" = 1 / (T_k + 1e-9) ELEMENT-WISE, weights = its diagonal,
//           mu = sum_k w_k z_k / (sum_k w_k + 1e-9), returns inverse(sum_k P_k + 1e-9) (a
//           covariance where a scale matrix is expected); K == 1: z_0 and T_0 T_0^T
template <int D>
__global__ __launch_bounds__(256) void fuse_sensors_kernel(const float* __restrict__ z, const float* __restrict__ tril,
                                                           const float* __restrict__ w, float* __restrict__ z_out,
                                                           float* __restrict__ tril_out, int N, int K, int mode) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float num[D], den[D];
#pragma unroll
  for (int i = 0; i < D; ++i) num[i] = den[i] = 0.f;
  Mat<D> acc;
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) acc.a[i][j] = 0.f;
  float mult = 1.f;
  for (int k = 0; k < K; ++k) {
    const size_t row = static_cast<size_t>(k) * N + n;
    const Mat<D> T = load_mat<D>(tril + row * D * D);
    if (mode == 1) {
      const Mat<D> C = matmul_nt<D>(T, T);
#pragma unroll
      for (int i = 0; i < D; ++i) {
        const float wk = w[row * D + i];
        num[i] += wk * z[row * D + i];
        den[i] += wk;
        mult *= wk;
#pragma unroll
        for (int j = 0; j < D; ++j) acc.a[i][j] += C.a[i][j];
      }
    } else if (K == 1) {
      acc = matmul_nt<D>(T, T);
#pragma unroll
      for (int i = 0; i < D; ++i) { num[i] = z[row * D + i]; den[i] = 1.f - 1e-9f; }
    } else {
#pragma unroll
      for (int i = 0; i < D; ++i) {
#pragma unroll
        for (int j = 0; j < D; ++j) acc.a[i][j] += 1.0f / (T.a[i][j] + 1e-9f);
        const float wk = 1.0f / (T.a[i][i] + 1e-9f);
        num[i] += wk * z[row * D + i];
        den[i] += wk;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < D; ++i)
    z_out[static_cast<size_t>(n) * D + i] = (mode == 2 && K == 1) ? num[i] : num[i] / (den[i] + 1e-9f);
  Mat<D> out;
  if (mode == 1) {  // Cholesky of mult * acc
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j < D; ++j) { acc.a[i][j] *= mult; out.a[i][j] = 0.f; }
#pragma unroll
    for (int j = 0; j < D; ++j) {
      float s = acc.a[j][j];
#pragma unroll
      for (int k = 0; k < D; ++k)
        if (k < j) s -= out.a[j][k] * out.a[j][k];
      const float dj = sqrtf(s);
      out.a[j][j] = dj;
#pragma unroll
      for (int i = 0; i < D; ++i)
        if (i > j) {
          float t = acc.a[i][j];
#pragma unroll
          for (int k = 0; k < D; ++k)
            if (k < j) t -= out.a[i][k] * out.a[j][k];
          out.a[i][j] = t / dj;
        }
    }
  } else if (K == 1) {
    out = acc;
  } else {
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j < D; ++j) acc.a[i][j] += 1e-9f;
    out = inverse<D>(acc);
  }
  store_mat<D>(tril_out + static_cast<size_t>(n) * D * D, out);
}

}  // namespace

extern "C" int mmf_fuse_virtual_sensors(const float* z, const float* tril, const float* w, float* z_out,
                                        float* tril_out, int N, int d, int K, int mode, void* stream) {
  if (!z || !tril || !z_out || !tril_out) return MMF_EINVAL;
  if (N < 0 || K < 1 || (mode != 1 && mode != 2) || (mode == 1 && !w)) return MMF_EINVAL;
  if (N == 0) return 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int grid = (N + 255) / 256;
  if (d == 2) fuse_sensors_kernel<2><<<grid, 256, 0, s>>>(z, tril, w, z_out, tril_out, N, K, mode);
  else if (d == 3) fuse_sensors_kernel<3><<<grid, 256, 0, s>>>(z, tril, w, z_out, tril_out, N, K, mode);
  else return MMF_EINVAL;
  MMF_CHECK_LAUNCH();
  return 0;
}

extern "C" int mmf_ekf_step_gated(const float* A, const float* mu_pred, const float* q_tril,
                                  const float* z, const float* r_tril, const float* fuse_w, float* mu,
                                  float* Sigma, float* mu_f, float* Sigma_f, int N, int d, int K,
                                  int fusion, int feedback, const int32_t* feedback_gate, void* stream);

extern "C" int mmf_ekf_step(const float* A, const float* mu_pred, const float* q_tril,
                            const float* z, const float* r_tril, const float* fuse_w, float* mu,
                            float* Sigma, float* mu_f, float* Sigma_f, int N, int d, int K,
                            int fusion, int feedback, void* stream) {
  return mmf_ekf_step_gated(A, mu_pred, q_tril, z, r_tril, fuse_w, mu, Sigma, mu_f, Sigma_f, N, d, K, fusion, feedback,
                            nullptr, stream);
}

extern "C" int mmf_ekf_step_gated(const float* A, const float* mu_pred, const float* q_tril,
                                  const float* z, const float* r_tril, const float* fuse_w, float* mu,
                                  float* Sigma, float* mu_f, float* Sigma_f, int N, int d, int K,
                                  int fusion, int feedback, const int32_t* feedback_gate, void* stream) {
  if (!A || !mu_pred || !q_tril || !z || !r_tril || !mu || !Sigma) return MMF_EINVAL;
  if (N < 0 || d < 1 || d > MMF_MAX_STATE_DIM || K < 1 || K > kMaxK) return MMF_EINVAL;
  if (fusion < 0 || fusion > 2 || (fusion == 1 && !fuse_w)) return MMF_EINVAL;
  if (fusion != 0 && (!mu_f || !Sigma_f)) return MMF_EINVAL;
  if (N == 0) return 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int block = 256, grid = (N + block - 1) / block;
#define MMF_K3(D)                                                                               \
  case D:                                                                                       \
    ekf_step_kernel<D><<<grid, block, 0, s>>>(A, mu_pred, q_tril, z, r_tril, fuse_w, mu, Sigma,  \
                                              mu_f, Sigma_f, N, K, fusion, feedback, feedback_gate); \
    break;
  switch (d) {
    MMF_K3(1) MMF_K3(2) MMF_K3(3) MMF_K3(4)
  }
#undef MMF_K3
  MMF_CHECK_LAUNCH();
  return 0;
}

extern "C" int mmf_ekf_step_backward(const float* A, const float* mu_pred, const float* q_tril, const float* z,
                                     const float* r_tril, const float* Sigma_in, const float* g_mu,
                                     const float* g_Sigma, float* g_A, float* g_mu_pred, float* g_z,
                                     float* g_r_tril, float* g_Sigma_in, int N, int d, int K, void* stream) {
  if (!A || !mu_pred || !q_tril || !z || !r_tril || !Sigma_in) return MMF_EINVAL;
  if (N < 0 || K < 1 || d < 1 || d > MMF_MAX_STATE_DIM) return MMF_EINVAL;
  if (N == 0) return 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int grid = (K * N + 255) / 256;
#define MMF_K3B(D)                                                                                          \
  case D:                                                                                                   \
    ekf_step_backward_kernel<D><<<grid, 256, 0, s>>>(A, mu_pred, q_tril, z, r_tril, Sigma_in, g_mu, g_Sigma, \
                                                     g_A, g_mu_pred, g_z, g_r_tril, g_Sigma_in, N, K);      \
    break;
  switch (d) { MMF_K3B(1) MMF_K3B(2) MMF_K3B(3) MMF_K3B(4) }
#undef MMF_K3B
  MMF_CHECK_LAUNCH();
  return 0;
}
