// K6, native recursion: forward and backward of a train-mode (no resampling) particle filter over T
// steps, each as ONE C call that enqueues every kernel on the caller's stream.
//
// Replaces torch autograd through torchfilter's ParticleFilter.forward_loop as the reference trains
// it (/root/reference/crossmodal/train_helpers.py:124-162 -> torchfilter.train.train_filter; sizes
// scripts/door_task/train_door.py:63-71: batch 32, 30 particles, subsequences of 16): at that size a
// step of the torch formulation is ~7,000 tiny launches; here it is ~20 per time step, and the
// gradient sums over time steps happen in place on the device.
//
// Memory / traffic design (MI355X): the forward keeps only the particle sets and log-weights.  The
// backward RECOMPUTES each step's layer inputs (mmf_particle_net_train_forward) for a chunk of
// trajectories into one reused stash buffer, runs the transposed network over it
// (mmf_particle_net_train_backward, ReLU masks as bits) and reduces the weight gradients
// (mmf_particle_net_weight_grads_acc) before the next chunk overwrites the buffers: peak memory no longer
// grows with T (25.4 -> 2.2 GB at 32 x 8192 x 16).  Chunks small enough for stash + dz to stay inside the
// 256 MiB Infinity Cache were measured and LOSE (32,768 rows: 65.7 ms per step, 262,144 rows: 45.0): each
// small launch re-stages 150 KB of weights per workgroup and half-fills the chip.  The caller sizes chunks
// for memory (engine.TRAIN_CHUNK_ROWS).
#include "mmf_common.h"

namespace {

constexpr int kThreads = 256;

// loglik = logsumexp_k ll_k        (ll_k = raw_k + b_k + beta_k, written by the measurement kernels)
__global__ __launch_bounds__(kThreads) void combine_fwd_kernel(const float* __restrict__ ll, float* __restrict__ out, int K,
                                                               size_t R) {
  const size_t r = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (r >= R) return;
  float m = -INFINITY;
  for (int k = 0; k < K; ++k) m = fmaxf(m, ll[k * R + r]);
  if (m == -INFINITY) { out[r] = m; return; }
  float s = 0.f;
  for (int k = 0; k < K; ++k) s += expf(ll[k * R + r] - m);
  out[r] = m + logf(s);
}

// d ll_k = d loglik * softmax_k(ll)      rows [r0, r0 + C) of the (K, R) array -> d_raw (K, C)
__global__ __launch_bounds__(kThreads) void combine_bwd_kernel(const float* __restrict__ ll, const float* __restrict__ d_a,
                                                               float* __restrict__ d_raw, int K, size_t R, size_t r0,
                                                               size_t C) {
  const size_t c = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (c >= C) return;
  const size_t r = r0 + c;
  float m = -INFINITY;
  for (int k = 0; k < K; ++k) m = fmaxf(m, ll[k * R + r]);
  float s = 0.f;
  for (int k = 0; k < K; ++k) s += (m == -INFINITY) ? 1.f : expf(ll[k * R + r] - m);
  const float g = d_a[r] / s;
  for (int k = 0; k < K; ++k) d_raw[k * C + c] = (m == -INFINITY) ? g : g * expf(ll[k * R + r] - m);
}

// dst[i] = a[i] (+ b[i])
__global__ __launch_bounds__(kThreads) void sum2_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                        float* __restrict__ dst, size_t n) {
  const size_t i = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (i < n) dst[i] = a[i] + (b ? b[i] : 0.f);
}

// dst[i] += src[i]
__global__ __launch_bounds__(kThreads) void add_kernel(float* __restrict__ dst, const float* __restrict__ src, size_t n) {
  const size_t i = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (i < n) dst[i] += src[i];
}

// x' = x + dir sigmoid(gate) (+ L eps): given g = dL/dx' and the raw head outputs (dir_0..dir_{D-1}, gate; bias
// included): d dir_i = g_i s, d gate = (sum_i g_i dir_i) s (1 - s)
template <int D>
__global__ __launch_bounds__(kThreads) void dyn_epilogue_bwd_kernel(const float* __restrict__ raw, const float* __restrict__ g,
                                                                    float* __restrict__ d_raw, size_t C) {
  const size_t c = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x;
  if (c >= C) return;
  const float gate = raw[c * (D + 1) + D];
  const float s = 1.0f / (1.0f + expf(-gate));
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < D; ++i) {
    const float gi = g[c * D + i];
    d_raw[c * (D + 1) + i] = gi * s;
    dot += gi * raw[c * (D + 1) + i];
  }
  d_raw[c * (D + 1) + D] = dot * s * (1.0f - s);
}

inline int blocks(size_t n) { return static_cast<int>((n + kThreads - 1) / kThreads); }

int check(const MmfPfTrainArgs* a) {
  if (!a) return MMF_EINVAL;
  if (a->T < 0 || a->N < 1 || a->M < 1 || a->n_meas < 1 || a->n_meas > MMF_LOOP_MAX_MEAS) return MMF_EINVAL;
  if (a->d != 2 && a->d != 3) return MMF_EINVAL;
  if (!a->dyn.packed || !a->dyn_bias || !a->noise || !a->scale_tril || !a->states || !a->logw || !a->estimates ||
      !a->loglik || !a->ll_steps)
    return MMF_EINVAL;
  for (int k = 0; k < a->n_meas; ++k)
    if (!a->meas[k].packed || !a->meas_bias[k]) return MMF_EINVAL;
  return 0;
}

}  // namespace

extern "C" int mmf_pf_train_forward(const MmfPfTrainArgs* a, void* stream) {
  int rc = check(a);
  if (rc) return rc;
  hipStream_t hs = static_cast<hipStream_t>(stream);
  const size_t row = static_cast<size_t>(a->N), R = row * a->M;
  const int K = a->n_meas;
  for (int t = 0; t < a->T; ++t) {
    const float* x = a->states + t * R * a->d;
    float* xn = a->states + (t + 1) * R * a->d;
    rc = mmf_pf_dynamics(a->dyn.packed, a->n_res_dyn, a->precision, x, a->dyn_bias + t * row * MMF_UNITS,
                         a->noise + t * R * a->d, a->scale_tril, xn, a->range_flag, a->N, a->M, a->d, stream);
    if (rc) return rc;
    float* ll = a->ll_steps + static_cast<size_t>(t) * K * R;  // (K, R): kept for the backward's softmax over modalities
    {  // every modality's own log-likelihood: one launch, blockIdx.y = modality
      const float *pk[MMF_LOOP_MAX_MEAS], *bias[MMF_LOOP_MAX_MEAS], *lw[MMF_LOOP_MAX_MEAS];
      float* out[MMF_LOOP_MAX_MEAS];
      for (int k = 0; k < K; ++k) {
        pk[k] = a->meas[k].packed;
        bias[k] = a->meas_bias[k] + t * row * MMF_UNITS;
        lw[k] = a->meas_logw[k] ? a->meas_logw[k] + t * row * a->logw_stride : nullptr;
        out[k] = ll + k * R;
      }
      rc = mmf_pf_measure_multi(pk, K, a->n_res_meas, a->precision, xn, bias, lw, a->logw_stride, out, a->range_flag, a->N, a->M,
                                a->d, hs);
      if (rc) return rc;
    }
    const float* loglik = ll;
    if (K > 1) {
      combine_fwd_kernel<<<blocks(R), kThreads, 0, hs>>>(ll, a->loglik, K, R);
      MMF_CHECK_LAUNCH();
      loglik = a->loglik;
    }
    rc = mmf_pf_reweight_resample(loglik, a->logw + t * R, xn, nullptr, a->estimates + t * row * a->d, nullptr,
                                  a->logw + (t + 1) * R, nullptr, a->N, a->M, a->M, a->d, 0, stream);
    if (rc) return rc;
  }
  return 0;
}

extern "C" int mmf_pf_train_backward(const MmfPfTrainArgs* a, void* stream) {
  int rc = check(a);
  if (rc) return rc;
  if (!a->g_estimates || !a->stash || (!a->fused && (!a->mask || !a->raw)) || !a->dz || !a->d_raw || !a->g_states_a || !a->g_states_b || !a->g_logw_a ||
      !a->g_logw_b || !a->d_tmp || !a->d_states0 || !a->d_logw0 || a->chunk_traj < 1 || a->n_splits < 1 || a->n_slices < 1)
    return MMF_EINVAL;
  hipStream_t hs = static_cast<hipStream_t>(stream);
  const int N = a->N, M = a->M, d = a->d, K = a->n_meas, S = a->n_splits, SL = a->n_slices;
  const size_t row = static_cast<size_t>(N), R = row * M;
  const int NLd = 3 + 2 * a->n_res_dyn, NLm = 3 + 2 * a->n_res_meas;
  const int NLmax = NLd > NLm ? NLd : NLm;
  // fused (ABI 37): recompute + backward + weight gradients of a network call as ONE kernel (particle_net_fused.hip);
  // stash / dz / dz_scale then hold only the three (C, 64) row slots of the narrow reductions
  const size_t Cmax = static_cast<size_t>(a->chunk_traj < N ? a->chunk_traj : N) * M;
  const bool fused = a->fused != 0;
  if (fused && (a->precision != MMF_PREC_F16X3 || !a->fused_act || !a->fused_g_act || a->n_splits > 256)) return MMF_EINVAL;
  if (!fused && a->precision != MMF_PREC_F32) return MMF_EINVAL;  // the three passes multiply exact fp32 products
  if (!a->dz_scale) return MMF_EINVAL;
  // stash / dz are f16 arrays (ABI 39: always) + fp32 row scales of dz
  const size_t esz = 2;
  auto stash_of = [&](int set) { return reinterpret_cast<char*>(a->stash) + 0 * esz; };
  auto dz_of = [&](int set) { return reinterpret_cast<char*>(a->dz) + 0 * esz; };
  auto scale_of = [&](int set) { return a->dz_scale + 0; };
  auto net_fwd = [&](const MmfTrainNet& net, int n_res, int kind, const float* xs, const float* bias, char* stash, uint32_t* mask,
                     float* raw, int Nc, hipStream_t s) {
    return mmf_internal_train_forward_h(net.packed_f32, n_res, kind, xs, bias, stash, mask, raw, Nc, M, d, s);
  };
  auto net_bwd = [&](const MmfTrainNet& net, int n_res, int kind, const uint32_t* mask, const float* d_out, char* dz, float* sc,
                     float* d_states, int Ci, hipStream_t s) {
    return mmf_internal_train_backward_h(net.packed_t, net.head_w, n_res, kind, mask, d_out, dz, sc, d_states, Ci, d, s);
  };
  auto net_wgrads = [&](const MmfTrainNet& net, const char* dz, const float* sc, const char* stash, int n_layers, int Ci, int acc,
                        hipStream_t s) { return mmf_internal_weight_grads_h(dz, sc, stash, net.pw, net.pb, n_layers, Ci, S, acc, s); };
  // the narrow reductions read dz of the first layer (slot NL) and of the join (slot 2), and the head's input (stash slot NL)
  auto net_sgrads = [&](const MmfTrainNet& net, const char* dz, const float* sc, const char* stash, int NL, size_t C, const float* xs,
                        const float* d_out, size_t slot0, int Nc, int n_out, hipStream_t s) {
    const size_t oL = static_cast<size_t>(NL) * C * MMF_UNITS * esz, o2 = 2 * C * MMF_UNITS * esz;
    float *pf = net.p_first + slot0 * MMF_UNITS * 4, *ph = net.p_head + slot0 * 4 * MMF_UNITS, *pd = net.p_dout + slot0 * 4,
          *pt = net.p_traj + slot0 * MMF_UNITS;
    return mmf_internal_small_grads_h(dz + oL, sc + static_cast<size_t>(NL) * C, dz + o2, sc + 2 * C, stash + oL, xs, d_out, pf, ph,
                                      pd, pt, Nc, M, d, n_out, SL, s);
  };
  // fused: one launch (dynamics: three) per network call, then the narrow reductions on the rows it left
  // row slots of one fused network call (set k of the caller's scratch when the step's networks share a launch)
  const bool merged = fused && K > 1 && a->fused_sets >= K;
  auto fused_args = [&](const MmfTrainNet& net, int set, int n_res, int kind, const float* xs, const float* bias, const float* d_out,
                        const float* g_next, float* d_raw, float* d_states, const float* d_states_base, size_t C, int Nc) {
    char* stash = reinterpret_cast<char*>(a->stash) + static_cast<size_t>(set) * Cmax * MMF_UNITS * 2;
    char* dz = reinterpret_cast<char*>(a->dz) + static_cast<size_t>(set) * 2 * Cmax * MMF_UNITS * 2;
    float* sc = a->dz_scale + static_cast<size_t>(set) * 2 * Cmax;
    MmfTrainFusedArgs f{};
    f.packed_dual = net.packed_dual; f.n_res = n_res; f.kind = kind; f.d = d; f.N = Nc; f.M = M; f.n_slots = S;
    f.states = xs; f.traj_bias = bias; f.d_out = d_out; f.g_next = g_next; f.d_raw = d_raw; f.act = a->fused_act;
    f.g_act = a->fused_g_act; f.d_states = d_states; f.d_states_base = d_states_base;
    f.dz_first_h = dz; f.sc_first = sc; f.dz_join_h = dz + C * MMF_UNITS * 2; f.sc_join = sc + C; f.h_last_h = stash;
    f.pw = net.pw; f.pb = net.pb;
    return f;
  };
  auto fused_small_grads = [&](const MmfTrainNet& net, const MmfTrainFusedArgs& f, const float* d_out_used, size_t slot0, int Nc, int n_out) {
    float *pf = net.p_first + slot0 * MMF_UNITS * 4, *ph = net.p_head + slot0 * 4 * MMF_UNITS, *pd = net.p_dout + slot0 * 4,
          *pt = net.p_traj + slot0 * MMF_UNITS;
    return mmf_internal_small_grads_h(f.dz_first_h, f.sc_first, f.dz_join_h, f.sc_join, f.h_last_h, f.states, d_out_used, pf, ph,
                                      pd, pt, Nc, M, d, n_out, SL, stream);
  };
  // fused: one launch (dynamics: three) per network call, then the narrow reductions on the rows it left
  auto net_fused = [&](const MmfTrainNet& net, int n_res, int kind, const float* xs, const float* bias, const float* d_out,
                       const float* g_next, float* d_raw, float* d_states, const float* d_states_base, int NL, size_t C,
                       size_t slot0, int Nc, int n_out) {
    if (!net.packed_dual) return static_cast<int>(MMF_EINVAL);
    (void)NL;
    const MmfTrainFusedArgs f = fused_args(net, 0, n_res, kind, xs, bias, d_out, g_next, d_raw, d_states, d_states_base, C, Nc);
    int r = mmf_particle_net_train_fused(&f, stream);
    if (r) return r;
    return fused_small_grads(net, f, kind == 1 ? d_out : d_raw, slot0, Nc, n_out);
  };
  auto mask_of = [&](int set) { return a->mask + 0; };
  auto raw_of = [&](int set) { return a->raw + 0; };
  auto tmp_of = [&](int set) { return a->d_tmp + 0; };
  float* d_raw_dyn = a->d_raw + 0;
  float* g_next = a->g_states_a;   // dL/d states[t+1] arriving from step t+1 (none at the last step)
  float* g_tot = a->g_states_b;
  float* g_lw = a->g_logw_a;       // dL/d logw[t+1] arriving from step t+1
  float* d_a = a->g_logw_b;
  bool first_wgrad_dyn = true;
  bool first_wgrad_meas[MMF_LOOP_MAX_MEAS];
  for (int k = 0; k < K; ++k) first_wgrad_meas[k] = true;

  for (int t = a->T - 1; t >= 0; --t) {
    const bool last = t == a->T - 1;
    const float* x = a->states + t * R * d;          // input of step t's dynamics
    const float* xn = a->states + (t + 1) * R * d;   // its output: what was measured and averaged
    // K1 (mode 0) backward: d_a = dL/d(logw[t] + loglik), g_tot = dL/d states[t+1] through the estimate
    rc = mmf_pf_reweight_backward(a->logw + (t + 1) * R, xn, a->g_estimates + t * row * d, last ? nullptr : g_lw, d_a, g_tot,
                                  N, M, d, stream);
    if (rc) return rc;
    if (!last) {
      add_kernel<<<blocks(R * d), kThreads, 0, hs>>>(g_tot, g_next, R * d);
      MMF_CHECK_LAUNCH();
    }
    const float* ll = a->ll_steps + static_cast<size_t>(t) * K * R;
    for (int n0 = 0; n0 < N; n0 += a->chunk_traj) {
      const int Nc = (N - n0 < a->chunk_traj) ? N - n0 : a->chunk_traj;
      const size_t r0 = static_cast<size_t>(n0) * M, C = static_cast<size_t>(Nc) * M;
      const int Ci = static_cast<int>(C);
      const size_t slot0 = (static_cast<size_t>(t) * N + n0) * SL;  // first slice slot of this chunk in the per-step partials
      // ---- measurement networks: d ll_k = d_a softmax_k, then each network's backward on recomputed activations
      if (K > 1) {
        combine_bwd_kernel<<<blocks(C), kThreads, 0, hs>>>(ll, d_a, a->d_raw, K, R, r0, C);
        MMF_CHECK_LAUNCH();
      }
      if (merged) {
        // the step's measurement networks differentiate independently: ONE launch (blockIdx.y = network).  Network 0 adds
        // its d states to the running gradient in its own store, the others leave theirs to be added in network order --
        // the sums and their order are those of one launch per network
        MmfTrainFusedArgs f[MMF_LOOP_MAX_MEAS];
        for (int k = 0; k < K; ++k) {
          if (!a->meas[k].packed_dual) return MMF_EINVAL;
          float* ds = k == 0 ? g_tot + r0 * d : a->d_tmp + static_cast<size_t>(k) * Cmax * d;
          f[k] = fused_args(a->meas[k], k, a->n_res_meas, 1, xn + r0 * d, a->meas_bias[k] + (t * row + n0) * MMF_UNITS, a->d_raw + k * C,
                            nullptr, nullptr, ds, k == 0 ? ds : nullptr, C, Nc);
        }
        rc = mmf_particle_net_train_fused_multi(f, K, stream);
        if (rc) return rc;
        for (int k = 0; k < K; ++k) {
          if (k > 0) {
            add_kernel<<<blocks(C * d), kThreads, 0, hs>>>(g_tot + r0 * d, a->d_tmp + static_cast<size_t>(k) * Cmax * d, C * d);
            MMF_CHECK_LAUNCH();
          }
          rc = fused_small_grads(a->meas[k], f[k], a->d_raw + k * C, slot0, Nc, 1);
          if (rc) return rc;
        }
      }
      for (int k = 0; k < K && !merged; ++k) {
        const MmfTrainNet& net = a->meas[k];
        const float* d_out = K > 1 ? a->d_raw + k * C : d_a + r0;
        hipStream_t sk = hs;
        if (fused) {
          // the network's d states is added to the running gradient in the kernel's own store (same sum, same order)
          rc = net_fused(net, a->n_res_meas, 1, xn + r0 * d, a->meas_bias[k] + (t * row + n0) * MMF_UNITS, d_out, nullptr, nullptr,
                         g_tot + r0 * d, g_tot + r0 * d, NLm, C, slot0, Nc, 1);
          if (rc) return rc;
          continue;
        }
        char *stash = stash_of(k), *dz = dz_of(k);
        float* sc = scale_of(k);
        rc = net_fwd(net, a->n_res_meas, 1, xn + r0 * d, a->meas_bias[k] + (t * row + n0) * MMF_UNITS, stash, mask_of(k), raw_of(k), Nc, sk);
        if (rc) return rc;
        rc = net_bwd(net, a->n_res_meas, 1, mask_of(k), d_out, dz, sc, tmp_of(k), Ci, sk);
        if (rc) return rc;
        rc = net_wgrads(net, dz, sc, stash, NLm + 1, Ci, first_wgrad_meas[k] ? 0 : 1, sk);
        if (rc) return rc;
        first_wgrad_meas[k] = false;
        rc = net_sgrads(net, dz, sc, stash, NLm, C, xn + r0 * d, d_out, slot0, Nc, 1, sk);
        if (rc) return rc;
        add_kernel<<<blocks(C * d), kThreads, 0, hs>>>(g_tot + r0 * d, tmp_of(k), C * d);
        MMF_CHECK_LAUNCH();
      }
      // ---- dynamics network: x' = x + dir sigmoid(gate) + L eps
      if (fused) {
        // the sigmoid-gate epilogue's backward runs inside the trunk kernel (d_raw_dyn is its output)
        // dL/d states[t] = direct path (g_tot) + through the network: written by the encoder-backward launch itself
        rc = net_fused(a->dyn, a->n_res_dyn, 0, x + r0 * d, a->dyn_bias + (t * row + n0) * MMF_UNITS, nullptr, g_tot + r0 * d, d_raw_dyn,
                       g_next + r0 * d, g_tot + r0 * d, NLd, C, slot0, Nc, d + 1);
        if (rc) return rc;
      } else {
        const MmfTrainNet& net = a->dyn;
        hipStream_t sd = hs;
        char *stash = stash_of(K), *dz = dz_of(K);
        float* sc = scale_of(K);
        rc = net_fwd(net, a->n_res_dyn, 0, x + r0 * d, a->dyn_bias + (t * row + n0) * MMF_UNITS, stash, mask_of(K), raw_of(K), Nc, sd);
        if (rc) return rc;
        if (d == 2) dyn_epilogue_bwd_kernel<2><<<blocks(C), kThreads, 0, hs>>>(raw_of(K), g_tot + r0 * d, d_raw_dyn, C);
        else dyn_epilogue_bwd_kernel<3><<<blocks(C), kThreads, 0, hs>>>(raw_of(K), g_tot + r0 * d, d_raw_dyn, C);
        MMF_CHECK_LAUNCH();
        rc = net_bwd(net, a->n_res_dyn, 0, mask_of(K), d_raw_dyn, dz, sc, tmp_of(K), Ci, hs);
        if (rc) return rc;
        rc = net_wgrads(net, dz, sc, stash, NLd + 1, Ci, first_wgrad_dyn ? 0 : 1, hs);
        if (rc) return rc;
        first_wgrad_dyn = false;
        rc = net_sgrads(net, dz, sc, stash, NLd, C, x + r0 * d, d_raw_dyn, slot0, Nc, d + 1, hs);
        if (rc) return rc;
        // dL/d states[t] = direct path + through the network
        sum2_kernel<<<blocks(C * d), kThreads, 0, hs>>>(g_tot + r0 * d, tmp_of(K), g_next + r0 * d, C * d);
        MMF_CHECK_LAUNCH();
      }
    }
    // dL/d logw[t] = d_a: becomes the incoming log-weight gradient of step t - 1
    float* s = g_lw; g_lw = d_a; d_a = s;
  }
  if (a->T > 0) {
    hipError_t e = hipMemcpyAsync(a->d_states0, g_next, R * d * sizeof(float), hipMemcpyDeviceToDevice, hs);
    if (e != hipSuccess) return static_cast<int>(e);
    e = hipMemcpyAsync(a->d_logw0, g_lw, R * sizeof(float), hipMemcpyDeviceToDevice, hs);
    if (e != hipSuccess) return static_cast<int>(e);
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// What is left after the recursion: the partial sums of one network -- a weight-gradient slot per workgroup (pw, pb), the
// narrow reductions per (step, trajectory, slice) (p_first, p_head, p_dout, p_traj) -- become the network's parameter
// gradients in the order and layout of its nn.Module parameters, the gradient of its hoisted per-trajectory term and of
// its modality log-weight column.  Round 4 left this to ~20 torch reductions / copies per network and step.  Every sum runs
// in ascending index order: reproducible run to run.
namespace {

constexpr int kFinChunks = 32;
constexpr int kFinSmall = 2 * MMF_UNITS * 4 + 4;  // first (64 x 4) | head (4 x 64) | d out (4)

// stage 1: the long sums (T * N * SL terms each), cut into kFinChunks runs
__global__ __launch_bounds__(256) void train_finalize_partial_kernel(MmfPfTrainFinalizeArgs a, float* __restrict__ partial) {
  const int terms = a.T * a.N * a.SL;
  const int per = (terms + kFinChunks - 1) / kFinChunks;
  const int t0 = blockIdx.x * per, t1 = min(terms, t0 + per);
  for (int e = threadIdx.x; e < kFinSmall; e += blockDim.x) {
    float s = 0.f;
    if (e < 256) {
      for (int t = t0; t < t1; ++t) s = __fadd_rn(s, a.p_first[static_cast<size_t>(t) * 256 + e]);
    } else if (e < 512) {
      for (int t = t0; t < t1; ++t) s = __fadd_rn(s, a.p_head[static_cast<size_t>(t) * 256 + (e - 256)]);
    } else {
      for (int t = t0; t < t1; ++t) s = __fadd_rn(s, a.p_dout[static_cast<size_t>(t) * 4 + (e - 512)]);
    }
    partial[blockIdx.x * kFinSmall + e] = s;
  }
}

__device__ __forceinline__ float sum_strided(const float* p, int n, size_t stride) {
  float s = 0.f;
  for (int k = 0; k < n; ++k) s = __fadd_rn(s, p[k * stride]);
  return s;
}

// stage 2: one thread per output value
__global__ __launch_bounds__(256) void train_finalize_kernel(MmfPfTrainFinalizeArgs a, const float* __restrict__ partial) {
  const int U = MMF_UNITS, UU = U * U;
  const int NL = 3 + 2 * a.n_res;
  // flat layout of grads = the parameters' order: w_in (U x d) | b_in (U) | enc block1 w, b | enc block2 w, b |
  // join w (U x join_in) | per residual block: block1 w, b, block2 w, b | head w (n_out x U) | head b (n_out)
  const int o_win = 0, o_bin = o_win + U * a.d, o_enc = o_bin + U, o_join = o_enc + 2 * (UU + U);
  const int o_res = o_join + U * a.join_in, o_head = o_res + a.n_res * 2 * (UU + U), o_hb = o_head + a.n_out * U;
  const int n_param = o_hb + a.n_out;
  const int TN = a.T * a.N;
  const int n_traj = TN * U;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n_param) {
    float v;
    if (e < o_bin) {                       // first-layer weights
      const int u = e / a.d, c = e % a.d;
      v = sum_strided(partial + u * 4 + c, kFinChunks, kFinSmall);
    } else if (e < o_enc) {                // first-layer bias: column d of the narrow reduction (fused) or slot NL of pb
      const int u = e - o_bin;
      v = a.fused ? sum_strided(partial + u * 4 + a.d, kFinChunks, kFinSmall)
                  : sum_strided(a.pb + static_cast<size_t>(NL) * a.S * U + u, a.S, U);
    } else if (e < o_join || (e >= o_res && e < o_head)) {   // a 64 x 64 layer's weights or bias
      int r = e < o_join ? e - o_enc : e - o_res;
      const int blk = r / (UU + U);
      r -= blk * (UU + U);
      const int layer = (e < o_join ? 0 : 3) + blk;
      v = r < UU ? sum_strided(a.pw + static_cast<size_t>(layer) * a.S * UU + r, a.S, UU)
                 : sum_strided(a.pb + static_cast<size_t>(layer) * a.S * U + (r - UU), a.S, U);
    } else if (e < o_res) {                // join layer: only the state columns belong to this network
      const int u = (e - o_join) / a.join_in, c = (e - o_join) % a.join_in - a.join_state_off;
      v = (c >= 0 && c < U) ? sum_strided(a.pw + static_cast<size_t>(2) * a.S * UU + u * U + c, a.S, UU) : 0.f;
    } else if (e < o_hb) {                 // head weights
      v = sum_strided(partial + 256 + (e - o_head), kFinChunks, kFinSmall);
    } else {
      v = sum_strided(partial + 512 + (e - o_hb), kFinChunks, kFinSmall);
    }
    a.grads[e] = v;
  } else if (e < n_param + n_traj) {       // gradient of the hoisted per-trajectory term
    const int j = e - n_param, tn = j / U, u = j % U;
    a.bias_grad[j] = sum_strided(a.p_traj + (static_cast<size_t>(tn) * a.SL) * U + u, a.SL, U);
  } else if (e < n_param + n_traj + TN) {  // gradient of the modality log-weight column
    const int tn = e - n_param - n_traj;
    if (a.d_beta) a.d_beta[static_cast<size_t>(tn) * a.beta_stride + a.beta_col] =
        sum_strided(a.p_dout + (static_cast<size_t>(tn) * a.SL) * 4, a.SL, 4);
  }
}

}  // namespace

extern "C" int mmf_pf_train_finalize(const MmfPfTrainFinalizeArgs* a, void* stream) {
  if (!a || !a->pw || !a->pb || !a->p_first || !a->p_head || !a->p_dout || !a->p_traj || !a->grads || !a->bias_grad || !a->scratch)
    return MMF_EINVAL;
  if (a->d < 1 || a->d > 3 || a->n_out < 1 || a->n_out > 4 || a->n_res < 0 || a->S < 1 || a->SL < 1 || a->T < 1 || a->N < 1)
    return MMF_EINVAL;
  if (a->join_state_off < 0 || a->join_state_off + MMF_UNITS > a->join_in) return MMF_EINVAL;
  auto s = static_cast<hipStream_t>(stream);
  train_finalize_partial_kernel<<<kFinChunks, 256, 0, s>>>(*a, a->scratch);
  MMF_CHECK_LAUNCH();
  const int U = MMF_UNITS;
  const int n_param = U * a->d + U + (2 + 2 * a->n_res) * (U * U + U) + U * a->join_in + a->n_out * U + a->n_out;
  const int n = n_param + a->T * a->N * U + a->T * a->N;
  train_finalize_kernel<<<(n + 255) / 256, 256, 0, s>>>(*a, a->scratch);
  MMF_CHECK_LAUNCH();
  return 0;
}
