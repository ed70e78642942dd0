// Host-side step loop of the particle filter: one C call enqueues every kernel of T filter
// steps (dynamics, one measurement launch per modality, reweight + resample) on the caller's
// stream, back to back.  Replaces the Python loop of torchfilter's Filter.forward_loop
// (external dependency; call site /root/reference/crossmodal/eval_helpers.py:139-142) for the
// fused models: no interpreter work, tensor allocation or pointer marshalling between steps,
// so small problems (the reference's own evaluation: a few dozen trajectories x 300
// particles) are bound by kernel time instead of ~0.15 ms/step of host overhead.

#include "mmf_common.h"

static int pf_enqueue_steps(const MmfPfLoopArgs* a, void* stream, bool with_events) {
  if (!a) return MMF_EINVAL;
  if (a->T < 0 || a->N < 1 || a->M < 1 || a->n_meas < 1 || a->n_meas > MMF_LOOP_MAX_MEAS) return MMF_EINVAL;
  if (a->resample_mode < 0 || a->resample_mode > 2) return MMF_EINVAL;
  if (!a->dyn_packed || !a->dyn_bias || (!a->noise && a->noise_mode != 2) || !a->scale_tril || !a->states_a || !a->states_b ||
      !a->logw_a || !a->logw_b || !a->loglik || !a->estimates)
    return MMF_EINVAL;
  if (a->resample_mode != 0 && !a->uniforms) return MMF_EINVAL;
  const bool soft = a->resample_mode != 0 && a->soft_alpha > 0.f && a->soft_alpha < 1.f;
  if (a->estimate_argmax && !a->estimate_scratch) return MMF_EINVAL;
  const size_t row = static_cast<size_t>(a->N);
  const size_t nm = row * a->M;
  float* cur = a->states_a;   // belief on entry
  float* other = a->states_b;
  float* lw_cur = a->logw_a;
  float* lw_other = a->logw_b;
  hipStream_t hs = static_cast<hipStream_t>(stream);
  int ev = 0;  // optional timing events: [sample][dynamics, measure x n_meas, resample][start, end]
  const int stride = a->event_stride > 1 ? a->event_stride : 1;
  bool sampled = false;  // an event record costs a barrier packet: long loops sample every stride-th step
  auto mark = [&]() {
    if (sampled && with_events) {
      hipError_t e = hipEventRecord(static_cast<hipEvent_t>(a->events[ev++]), hs);
      if (e != hipSuccess) return static_cast<int>(e);
    }
    return 0;
  };
  for (int t = 0; t < a->T; ++t) {
    int rc;
    sampled = a->events && t % stride == stride / 2;  // the middle step of every stride-long window
    if ((rc = mark())) return rc;
    if (a->noise_mode == 2)
      rc = mmf_pf_dynamics_philox(a->dyn_packed, a->n_res_dyn, a->precision, cur, a->dyn_bias + t * row * MMF_UNITS,
                                  a->noise_seed, a->noise_step0 + static_cast<unsigned>(t), a->noise_traj0, a->scale_tril,
                                  other, a->range_flag, a->N, a->M, a->d, stream);
    else
      rc = mmf_pf_dynamics(a->dyn_packed, a->n_res_dyn, a->precision, cur, a->dyn_bias + t * row * MMF_UNITS,
                           a->noise + t * nm * a->d, a->scale_tril, other, a->range_flag, a->N, a->M, a->d,
                           stream);
    if (rc) return rc;
    if ((rc = mark())) return rc;
    // parity certificates keep every step's log-likelihoods and ancestors (null on the timed path)
    float* ll = a->loglik_steps ? a->loglik_steps + t * nm : a->loglik;
    int32_t* anc = a->indices_steps ? a->indices_steps + t * nm : nullptr;
    {
      for (int k = 0; k < a->n_meas; ++k) {
        const float* lw = a->meas_logw[k] ? a->meas_logw[k] + t * row * a->logw_stride : nullptr;
        if ((rc = mark())) return rc;
        rc = mmf_pf_measure(a->meas_packed[k], a->n_res_meas, a->precision, other,
                            a->meas_bias[k] + t * row * MMF_UNITS, lw, a->logw_stride, ll, k > 0,
                            a->range_flag, a->N, a->M, a->d, stream);
        if (rc) return rc;
        if ((rc = mark())) return rc;
      }
    }
    float* est = a->estimates + t * row * a->d;
    if (a->estimate_argmax) {
      // the particle with the largest pre-resampling weight; K1's weighted mean goes to the scratch.  In the plain
      // resampling loop the incoming weights are uniform from the second step on (see below)
      const bool uniform_in = a->resample_mode != 0 && !soft && t > 0;
      rc = mmf_pf_argmax_estimate(ll, uniform_in ? nullptr : lw_cur, other, est, a->N, a->M, a->d, stream);
      if (rc) return rc;
      est = a->estimate_scratch;
    }
    if ((rc = mark())) return rc;
    if (soft) {
      // torchfilter's soft resampling: survivors carry importance weights, so the log-weights travel every step
      const float* u = a->uniforms + t * (a->resample_mode == 1 ? row : nm);
      rc = mmf_pf_reweight_resample_soft(ll, lw_cur, other, u, est, cur, lw_other, anc, a->N, a->M, a->M, a->d,
                                         a->resample_mode, a->soft_alpha, stream);
      if (rc) return rc;
    } else if (a->resample_mode == 0) {
      rc = mmf_pf_reweight_resample(ll, lw_cur, other, nullptr, est, nullptr, lw_other, nullptr, a->N,
                                    a->M, a->M, a->d, 0, stream);
      if (rc) return rc;
      float* s = cur; cur = other; other = s;  // propagated particles are the new belief
    } else {
      const float* u = a->uniforms + t * (a->resample_mode == 1 ? row : nm);
      // every step of this loop resamples, so from the second step on the incoming weights are the
      // uniform -log M the previous step would have written, and only the last step's are ever read
      // again: 8 of the 40 B per particle-step stay out of HBM
      rc = mmf_pf_reweight_resample(ll, t == 0 ? lw_cur : nullptr, other, u, est, cur,
                                    t == a->T - 1 ? lw_other : nullptr, anc, a->N, a->M, a->M, a->d,
                                    a->resample_mode, stream);
      if (rc) return rc;  // resampled particles land back in `cur`
    }
    if ((rc = mark())) return rc;
    float* l = lw_cur; lw_cur = lw_other; lw_other = l;
  }
  // tell the caller where the belief ended up: bit 0 = states in states_b, bit 1 = log-weights in logw_b
  if (a->final_location)
    *a->final_location = (cur == a->states_b ? 1 : 0) | (lw_cur == a->logw_b ? 2 : 0);
  return 0;
}


extern "C" int mmf_pf_forward_loop(const MmfPfLoopArgs* a, void* stream) {
  MmfPfLoopArgs launches;
  if (a && a->persistent) {  // ONE launch for all T steps (small problems)
    const int rc = mmf_internal_pf_persistent(a, stream);
    if (rc != MMF_INTERNAL_NOT_RESIDENT) return rc;
    // this device cannot hold the persistent grid (a partition, or fewer CUs than planned for): the loop of launches
    // computes the same bits
    launches = *a;
    launches.persistent = 0;
    a = &launches;
  }
  return pf_enqueue_steps(a, stream, true);
}

// Open-loop rollout x_t = f(x_{t-1}, u_t): replaces torchfilter's DynamicsModel.forward_loop (call
// sites /root/reference/crossmodal/eval_helpers.py:135-137, scripts/door_task/eval_dynamics.py:36-38).
extern "C" int mmf_dynamics_forward_loop(const float* packed, int n_res, int precision, const float* x0,
                                         const float* traj_bias, float* out, int32_t* range_flag, int T, int N,
                                         int d, void* stream) {
  if (!packed || !x0 || !traj_bias || !out || T < 0 || N < 1) return MMF_EINVAL;
  const size_t row = static_cast<size_t>(N);
  const float* cur = x0;
  for (int t = 0; t < T; ++t) {
    float* nxt = out + t * row * d;
    const int rc = mmf_pf_dynamics(packed, n_res, precision, cur, traj_bias + t * row * MMF_UNITS, nullptr, nullptr,
                                   nxt, range_flag, N, 1, d, stream);
    if (rc) return rc;
    cur = nxt;
  }
  return 0;
}
