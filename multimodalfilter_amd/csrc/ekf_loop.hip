// Host-side step loop of the fused extended Kalman filters: one C call enqueues, for every
// time step, the forward-mode Jacobian launch of each sub-filter's dynamics network (K5) and
// the predict / correct / fuse launch (K3).  Replaces the Python loop of torchfilter's
// Filter.forward_loop over CrossmodalKalmanFilter.forward / UnimodalKalmanFilter.forward
// (/root/reference/crossmodal/base_models/crossmodal_kf.py:88-151, unimodal_kf.py:162-250; call
// site eval_helpers.py:139-142): the EKF algebra is a few microseconds per step, so the
// recursion was bound by ~20 launches and tensor shuffles of interpreter work per step.
#include "mmf_common.h"

extern "C" int mmf_ekf_forward_loop(const MmfEkfLoopArgs* a, void* stream) {
  if (!a) return MMF_EINVAL;
  if (a->T < 0 || a->N < 1 || a->K < 1 || a->K > MMF_LOOP_MAX_MEAS || a->d < 1) return MMF_EINVAL;
  if (a->fusion < 0 || a->fusion > 2) return MMF_EINVAL;
  if (!a->q_tril || !a->z || !a->r_tril || !a->mu || !a->Sigma || !a->mu_pred || !a->A || !a->estimates)
    return MMF_EINVAL;
  if (a->fusion == 1 && !a->fuse_w) return MMF_EINVAL;
  if (a->fusion != 0 && (!a->Sigma_f)) return MMF_EINVAL;
  for (int k = 0; k < a->K; ++k)
    if (!a->dyn_packed[k] || !a->dyn_bias[k]) return MMF_EINVAL;
  if (a->persistent && a->T > 0) {  // ONE launch for all T steps (ekf_persistent.inc); same bits
    const int rc = mmf_internal_ekf_persistent(a, stream);
    if (rc != MMF_INTERNAL_NOT_RESIDENT) return rc;
    // this device cannot hold the persistent grid, or the problem is not eligible: the loop of launches
  }
  const size_t N = static_cast<size_t>(a->N), d = static_cast<size_t>(a->d), K = static_cast<size_t>(a->K);
  hipStream_t hs = static_cast<hipStream_t>(stream);
  for (int t = 0; t < a->T; ++t) {
    {  // every sub-filter's Jacobian in one launch (they are ~45 us of latency each on their own)
      const float* bias[MMF_LOOP_MAX_MEAS];
      for (size_t k = 0; k < K; ++k) bias[k] = a->dyn_bias[k] + t * N * MMF_UNITS;
      const int rc = mmf_dynamics_jacobian_multi(a->dyn_packed, a->n_res_dyn, a->precision, a->mu, bias, a->mu_pred,
                                                 a->A, a->range_flag, a->K, a->N, a->d, stream);
      if (rc) return rc;
    }
    float* est = a->estimates + t * N * d;
    const int rc = mmf_ekf_step_gated(a->A, a->mu_pred, a->q_tril, a->z + t * K * N * d, a->r_tril + t * K * N * d * d,
                                      a->fuse_w ? a->fuse_w + t * K * N * d : nullptr, a->mu, a->Sigma,
                                      a->fusion ? est : nullptr, a->fusion ? a->Sigma_f : nullptr, a->N, a->d,
                                      a->K, a->fusion, a->feedback, a->feedback_gate ? a->feedback_gate + t : nullptr, stream);
    if (rc) return rc;
    if (a->fusion == 0) {  // a single (enabled) sub-filter: its corrected mean is the estimate
      const hipError_t e = hipMemcpyAsync(est, a->mu, N * d * sizeof(float), hipMemcpyDeviceToDevice, hs);
      if (e != hipSuccess) return static_cast<int>(e);
    }
  }
  return 0;
}
