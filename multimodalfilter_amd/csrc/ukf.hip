// Unscented transform around the dynamics model: sigma points of a batch of
// Gaussian beliefs, and the weighted moments of the propagated points.
//
// torchfilter ships UnscentedKalmanFilter / VirtualSensorUnscentedKalmanFilter next to the EKF the
// reference subclasses (door_models/kf.py:14-28); the package is absent from /root/reference and
// un-pinned, so its published algorithm is restated (oracle/tf/filters.py): per trajectory
//   X_0 = mu,  X_i = mu + sqrt(d + lambda) L[:, i],  X_{d+i} = mu - sqrt(d + lambda) L[:, i],   L = chol(Sigma)
// -> dynamics on all 2d+1 points (K2: they are rows of the per-particle network) ->
//   mu- = sum_i wm_i X'_i,   Sigma- = sum_i wc_i (X'_i - mu-)(X'_i - mu-)^T + Q.
// One trajectory per lane, d <= 4, everything in registers; HBM: 4(d + d^2) in, 4 d (2d+1) out
// per trajectory and back -- latency-bound by design, like K3.
#include <cmath>

#include "mmf_common.h"

namespace {

template <int D>
__global__ __launch_bounds__(256) void ukf_sigma_points_kernel(const float* __restrict__ mu, const float* __restrict__ Sigma,
                                                               float scale, float* __restrict__ points,
                                                               int* __restrict__ not_pd, int N) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  constexpr int P = 2 * D + 1;
  const float* C = Sigma + static_cast<size_t>(n) * D * D;
  float L[D][D];
  bool bad = false;
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) L[i][j] = 0.f;
#pragma unroll
  for (int j = 0; j < D; ++j) {
    float s = C[j * D + j];
#pragma unroll
    for (int k = 0; k < D; ++k)
      if (k < j) s -= L[j][k] * L[j][k];
    bad = bad || !(s > 0.f);
    const float dj = sqrtf(s);
    L[j][j] = dj;
#pragma unroll
    for (int i = 0; i < D; ++i)
      if (i > j) {
        float t = C[i * D + j];
#pragma unroll
        for (int k = 0; k < D; ++k)
          if (k < j) t -= L[i][k] * L[j][k];
        L[i][j] = t / dj;
      }
  }
  if (bad && not_pd) atomicOr(not_pd, 1);
  if (bad) {
    // not positive definite: reported through *not_pd (the caller raises, once per step or once per
    // loop); the points collapse onto the mean so the networks downstream see finite, in-range rows
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j < D; ++j) L[i][j] = 0.f;
  }
  float m[D];
#pragma unroll
  for (int i = 0; i < D; ++i) m[i] = mu[static_cast<size_t>(n) * D + i];
  float* out = points + static_cast<size_t>(n) * P * D;
#pragma unroll
  for (int i = 0; i < D; ++i) out[i] = m[i];
#pragma unroll
  for (int c = 0; c < D; ++c)
#pragma unroll
    for (int i = 0; i < D; ++i) {
      out[(1 + c) * D + i] = m[i] + scale * L[i][c];
      out[(1 + D + c) * D + i] = m[i] - scale * L[i][c];
    }
}

template <int D>
__global__ __launch_bounds__(256) void ukf_moments_kernel(const float* __restrict__ points, float wm0, float wc0, float wi,
                                                          const float* __restrict__ q_tril, float* __restrict__ mu_pred,
                                                          float* __restrict__ Sigma_pred, int N) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  constexpr int P = 2 * D + 1;
  const float* X = points + static_cast<size_t>(n) * P * D;
  float x[P][D], m[D];
#pragma unroll
  for (int p = 0; p < P; ++p)
#pragma unroll
    for (int i = 0; i < D; ++i) x[p][i] = X[p * D + i];
#pragma unroll
  for (int i = 0; i < D; ++i) {
    float s = wm0 * x[0][i];
#pragma unroll
    for (int p = 1; p < P; ++p) s += wi * x[p][i];
    m[i] = s;
    mu_pred[static_cast<size_t>(n) * D + i] = s;
  }
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) {
      float s = wc0 * (x[0][i] - m[i]) * (x[0][j] - m[j]);
#pragma unroll
      for (int p = 1; p < P; ++p) s += wi * (x[p][i] - m[i]) * (x[p][j] - m[j]);
      float q = 0.f;
#pragma unroll
      for (int k = 0; k < D; ++k) q += q_tril[i * D + k] * q_tril[j * D + k];
      Sigma_pred[(static_cast<size_t>(n) * D + i) * D + j] = s + q;
    }
}

}  // namespace

extern "C" int mmf_ukf_sigma_points(const float* mu, const float* Sigma, float scale, float* points,
                                    int32_t* not_pd, int N, int d, void* stream) {
  if (!mu || !Sigma || !points) return MMF_EINVAL;
  if (N < 0 || d < 1 || d > MMF_MAX_STATE_DIM || !(scale > 0.f)) return MMF_EINVAL;
  if (N == 0) return 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int grid = (N + 255) / 256;
  switch (d) {
    case 1: ukf_sigma_points_kernel<1><<<grid, 256, 0, s>>>(mu, Sigma, scale, points, not_pd, N); break;
    case 2: ukf_sigma_points_kernel<2><<<grid, 256, 0, s>>>(mu, Sigma, scale, points, not_pd, N); break;
    case 3: ukf_sigma_points_kernel<3><<<grid, 256, 0, s>>>(mu, Sigma, scale, points, not_pd, N); break;
    case 4: ukf_sigma_points_kernel<4><<<grid, 256, 0, s>>>(mu, Sigma, scale, points, not_pd, N); break;
  }
  MMF_CHECK_LAUNCH();
  return 0;
}

extern "C" int mmf_ukf_moments(const float* points, float wm0, float wc0, float wi, const float* q_tril,
                               float* mu_pred, float* Sigma_pred, int N, int d, void* stream) {
  if (!points || !q_tril || !mu_pred || !Sigma_pred) return MMF_EINVAL;
  if (N < 0 || d < 1 || d > MMF_MAX_STATE_DIM) return MMF_EINVAL;
  if (N == 0) return 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int grid = (N + 255) / 256;
  switch (d) {
    case 1: ukf_moments_kernel<1><<<grid, 256, 0, s>>>(points, wm0, wc0, wi, q_tril, mu_pred, Sigma_pred, N); break;
    case 2: ukf_moments_kernel<2><<<grid, 256, 0, s>>>(points, wm0, wc0, wi, q_tril, mu_pred, Sigma_pred, N); break;
    case 3: ukf_moments_kernel<3><<<grid, 256, 0, s>>>(points, wm0, wc0, wi, q_tril, mu_pred, Sigma_pred, N); break;
    case 4: ukf_moments_kernel<4><<<grid, 256, 0, s>>>(points, wm0, wc0, wi, q_tril, mu_pred, Sigma_pred, N); break;
  }
  MMF_CHECK_LAUNCH();
  return 0;
}
