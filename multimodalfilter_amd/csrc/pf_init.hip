// Particle-filter belief initialisation: particles ~ N(mean, covariance) from pre-drawn
// standard-normal noise, uniform log-weights.
//
// Replaces torchfilter's ParticleFilter.initialize_beliefs (external dependency of the
// reference; call site /root/reference/crossmodal/eval_helpers.py:125-131; SURVEY.md A.2:
// "particles ~ MVN(mean, covariance) ... log-weights -log M"): per trajectory the Cholesky
// factor L of the d x d covariance (d <= 4, in registers), x_m = mean + L eps_m.  A
// non-positive pivot (covariance not positive definite) sets *not_pd; the Python layer raises,
// as MultivariateNormal's argument validation does upstream.
#include <cmath>

#include "mmf_common.h"

namespace {

template <int D>
__global__ __launch_bounds__(256) void pf_init_particles_kernel(
    const float* __restrict__ mean, const float* __restrict__ cov, const float* __restrict__ eps,
    float* __restrict__ states, float* __restrict__ logw, int* __restrict__ not_pd, int M, float lw) {
  const int n = blockIdx.x;
  float L[D][D];
  const float* C = cov + static_cast<size_t>(n) * D * D;
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
  if (bad && threadIdx.x == 0 && not_pd) atomicOr(not_pd, 1);
  float mu[D];
#pragma unroll
  for (int i = 0; i < D; ++i) mu[i] = mean[static_cast<size_t>(n) * D + i];
  const size_t base = static_cast<size_t>(n) * M;
  for (int m = threadIdx.x; m < M; m += blockDim.x) {
    float e[D];
#pragma unroll
    for (int i = 0; i < D; ++i) e[i] = eps[(base + m) * D + i];
#pragma unroll
    for (int i = 0; i < D; ++i) {
      float v = 0.f;
#pragma unroll
      for (int k = 0; k < D; ++k)
        if (k <= i) v += L[i][k] * e[k];
      states[(base + m) * D + i] = mu[i] + v;
    }
    logw[base + m] = lw;
  }
}

// K6 (K1, no-resample path): backward of
//     a = logw_in + loglik;  lw = a - logsumexp_m(a);  estimate_i = sum_m exp(lw_m) x_mi
// (torchfilter's train-mode particle-filter step).  With w = exp(lw) and upstream
// gradients g_est (N, d), g_lw (N, M):
//     G_m = g_lw_m + w_m sum_i x_mi g_est_i;   d a_k = G_k - w_k sum_m G_m;   d x_mi = w_m g_est_i
// One workgroup per trajectory, two passes over its M particles.
template <int D>
__global__ __launch_bounds__(1024) void pf_reweight_backward_kernel(
    const float* __restrict__ lw, const float* __restrict__ states, const float* __restrict__ g_est,
    const float* __restrict__ g_lw, float* __restrict__ d_a, float* __restrict__ d_states, int M) {
  __shared__ float red[16];
  const int n = blockIdx.x, tid = threadIdx.x;
  const size_t base = static_cast<size_t>(n) * M;
  float ge[D];
#pragma unroll
  for (int i = 0; i < D; ++i) ge[i] = g_est[static_cast<size_t>(n) * D + i];
  float sum = 0.f;
  for (int m = tid; m < M; m += blockDim.x) {
    const float w = expf(lw[base + m]);
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < D; ++i) {
      dot += states[(base + m) * D + i] * ge[i];
      d_states[(base + m) * D + i] = w * ge[i];
    }
    const float G = (g_lw ? g_lw[base + m] : 0.f) + w * dot;
    d_a[base + m] = G;  // completed in the second pass
    sum += G;
  }
  sum = mmf::wave_sum(sum);
  if ((tid & 63) == 0) red[tid >> 6] = sum;
  __syncthreads();
  float S = 0.f;
  for (int w = 0; w < static_cast<int>(blockDim.x >> 6); ++w) S += red[w];  // fixed order
  for (int m = tid; m < M; m += blockDim.x) d_a[base + m] -= expf(lw[base + m]) * S;
}

}  // namespace

extern "C" int mmf_pf_reweight_backward(const float* logw_out, const float* states, const float* g_estimate,
                                        const float* g_logw_out, float* d_a, float* d_states, int N, int M,
                                        int d, void* stream) {
  if (!logw_out || !states || !g_estimate || !d_a || !d_states) return MMF_EINVAL;
  if (N < 0 || M < 1) return MMF_EINVAL;
  if (N == 0) return 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // one workgroup per trajectory: with few trajectories of many particles (config 5: 32 x 8192) the launch is ONE
  // latency chain per workgroup -- 1024 threads quarter it (43 -> ~12 us at 32 x 8192)
  const int threads = M >= 2048 ? 1024 : 256;
  if (d == 2) pf_reweight_backward_kernel<2><<<N, threads, 0, s>>>(logw_out, states, g_estimate, g_logw_out, d_a, d_states, M);
  else if (d == 3) pf_reweight_backward_kernel<3><<<N, threads, 0, s>>>(logw_out, states, g_estimate, g_logw_out, d_a, d_states, M);
  else return MMF_EINVAL;
  MMF_CHECK_LAUNCH();
  return 0;
}

extern "C" int mmf_pf_init_particles(const float* mean, const float* covariance, const float* eps,
                                     float* states, float* logw, int32_t* not_pd, int N, int M, int d,
                                     void* stream) {
  if (!mean || !covariance || !eps || !states || !logw) return MMF_EINVAL;
  if (N < 0 || M < 1 || d < 1 || d > MMF_MAX_STATE_DIM) return MMF_EINVAL;
  if (N == 0) return 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // -log M evaluated in double and rounded once, as the host formulation (float(-math.log(M))) does
  const float lw = static_cast<float>(-std::log(static_cast<double>(M)));
#define MMF_INIT_CASE(D)                                                                             \
  if (d == D) {                                                                                      \
    pf_init_particles_kernel<D><<<N, 256, 0, s>>>(mean, covariance, eps, states, logw, not_pd, M, lw); \
    MMF_CHECK_LAUNCH();                                                                              \
    return 0;                                                                                        \
  }
  MMF_INIT_CASE(1) MMF_INIT_CASE(2) MMF_INIT_CASE(3) MMF_INIT_CASE(4)
#undef MMF_INIT_CASE
  return MMF_EINVAL;
}
