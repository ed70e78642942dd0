/*
 * mmf.h -- C ABI of libmmf_hip.so, the MI355X (gfx950) filter hot path.
 *
 * Drop-in boundary for brentyi/multimodalfilter's batched filter recursions.  The
 * reference is pure Python; the interface each entry point replaces is the sequence of
 * stock torch ops at the cited lines (reference paths are relative to
 * /root/reference/).  The reference-side binding is a ctypes
 * stub; multimodalfilter_amd/_abi.py is that stub in this repo.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer valid on the current HIP device unless marked
 *     "host"; arrays are contiguous fp32 unless stated; N = trajectories, M = particles
 *     per trajectory, d = state_dim (1..4), rows R = N*M
 *   - `stream` is a hipStream_t passed as void*; every call only enqueues work
 *   - return value: 0 on success, a hipError_t (>0) from the runtime, or a negative
 *     MMF_E* code for argument errors; no allocation, no global state, re-entrant per
 *     stream; the caller owns every buffer
 *   - nothing here has a CPU fallback: without a GPU the launch fails with a hipError_t
 */
#ifndef MMF_H
#define MMF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMF_ABI_VERSION 40

#define MMF_EINVAL (-1)      /* bad argument (null pointer, d out of range, ...) */
#define MMF_ETOOLARGE (-2)   /* size beyond what the kernel supports (see each call) */

#define MMF_UNITS 64         /* hidden width of every per-particle layer (layers.py: units=64) */
#define MMF_MAX_RES 3        /* residual blocks after the join layer (LDS: 3 -> 149.5 KiB) */
#define MMF_MAX_STATE_DIM 4

/* arithmetic of the 64x64 layers of the per-particle networks (K2) */
#define MMF_PREC_F32 0    /* v_mfma_f32_32x32x2_f32: exact fp32 products                        */
#define MMF_PREC_BF16 2   /* mmf_image_encoder only: operands rounded to ONE bf16, one v_mfma_*_bf16 per
                           * product, f32 accumulate (BASELINE config 5's "bf16 measurement CNN on MFMA")   */
#define MMF_PREC_F16X3 1  /* operands split x = hi + lo (2 x f16, exact to 2^-22), products
                             hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16, fp32 accumulate   */
#define MMF_PREC_F16X3_DUAL 3 /* mmf_pack_particle_net only (ABI 37): the f16x3 halves of every 64x64 layer as ONE
                             row-major, XOR-swizzled LDS image [out row][hi 64 | lo 64] that serves BOTH the forward
                             product (row reads, ds_read_b128) and the transposed product of the backward
                             (ds_read_b64_tr_b16): the fused training kernel keeps one copy of the weights in LDS.
                             Same blob size and section offsets as the other precisions */

int mmf_version(void);

/* ---------------------------------------------------------------- K1: reweight + resample
 * Replaces the post-measurement half of torchfilter.ParticleFilter.forward (external
 * dependency; call sites crossmodal/eval_helpers.py:139-142):
 *     logw += loglik; logw -= logsumexp(logw); estimate = sum exp(logw) x;
 *     indices ~ resample(logw); states = gather(states, indices); logw = -log(M_out)
 * as ONE kernel, one workgroup per trajectory: DPP wave reductions / scans, an integer
 * (fixed-point, 2^-24 of the row max) CDF, then systematic (mode 1: search-free -- every run of
 * particles announces the first output it owns, ancestors = prefix maximum of the announcements)
 * or multinomial (mode 2: binary search in LDS) selection.  The integer CDF makes the indices
 * independent of the scan order: bit-exact against oracle/resample.py.
 *
 *  loglik      (N, M)          measurement log-likelihoods
 *  logw_in     (N, M)          current log-weights; modes 1/2: null = uniform -log M (nothing is read)
 *  states_in   (N, M, d)
 *  u           mode 1: (N) uniforms in [0,1); mode 2: (N, M_out); mode 0: ignored (may be null)
 *  estimate    (N, d)          weighted-mean state estimate (always written)
 *  states_out  (N, M_out, d)   mode 1/2: resampled particles; must NOT alias states_in.
 *                              mode 0: may be null or alias states_in (no copy when it does)
 *  logw_out    (N, M_out)      mode 0: normalised log-weights (may alias logw_in);
 *                              mode 1/2: -log(M_out) (may alias logw_in when M_out == M); null = not
 *                              written (the survivors' weights are uniform by definition)
 *  indices_out (N, M_out) int32 ancestor indices, or null
 *  mode        0 none, 1 systematic, 2 multinomial
 * Limits: d <= 4; M, M_out <= 65536; mode 0 stages 4 B/particle in LDS (M <= 40,800),
 *         modes 1/2 keep the 8-byte CDF there (M <= 20,400: mmf_pf_reweight_resample_lds_bytes(M, mode)
 *         <= 160 KiB); larger -> MMF_ETOOLARGE.  Plain systematic resampling of M, M_out <= 20,000 takes
 *         the search-free kernel, everything else the CDF-search kernel.
 */
int mmf_pf_reweight_resample(const float* loglik, const float* logw_in, const float* states_in,
                             const float* u, float* estimate, float* states_out,
                             float* logw_out, int32_t* indices_out, int N, int M, int M_out,
                             int d, int mode, void* stream);

/* K1 with torchfilter's ``soft_resample_alpha`` option (upstream ParticleFilter._resample,
 * an option the reference leaves at its default 1.0): ancestors are drawn from the mixture
 * alpha * w + (1 - alpha) / M -- in fixed point, q'_i = ((A q_i << 8) + (2^24 - A) floor((Q << 8) / M)) >> 32
 * with A = floor(alpha 2^24) -- and the survivors carry the importance weights w / mixture, normalised,
 * instead of -log(M_out).  Same arguments as mmf_pf_reweight_resample; mode 1 or 2; 0 < alpha <= 1
 * (alpha == 1 is the plain resampler).
 */
int mmf_pf_reweight_resample_soft(const float* loglik, const float* logw_in, const float* states_in,
                                  const float* u, float* estimate, float* states_out,
                                  float* logw_out, int32_t* indices_out, int N, int M, int M_out,
                                  int d, int mode, float alpha, void* stream);

/* Belief initialisation (replaces torchfilter's ParticleFilter.initialize_beliefs; call site
 * eval_helpers.py:125-131): states[n][m] = mean[n] + chol(covariance[n]) eps[n][m], logw = -log M.
 *  mean (N, d), covariance (N, d, d), eps (N, M, d) standard normal -> states (N, M, d), logw (N, M)
 *  not_pd   device int32 or null: OR-ed with 1 when a covariance is not positive definite
 */
int mmf_pf_init_particles(const float* mean, const float* covariance, const float* eps, float* states,
                          float* logw, int32_t* not_pd, int N, int M, int d, void* stream);

/* K6: backward of the no-resample step (mode 0 of mmf_pf_reweight_resample):
 *   a = logw_in + loglik, logw_out = a - logsumexp_m(a), estimate = sum_m exp(logw_out) x_m.
 *  logw_out (N, M) as returned by the forward, states (N, M, d), g_estimate (N, d),
 *  g_logw_out (N, M) or null -> d_a (N, M) (= d loglik = d logw_in), d_states (N, M, d)
 */
int mmf_pf_reweight_backward(const float* logw_out, const float* states, const float* g_estimate,
                             const float* g_logw_out, float* d_a, float* d_states, int N, int M, int d,
                             void* stream);

/* Dynamic LDS bytes K1 will request for (M, mode) -- for occupancy planning / tests. */
size_t mmf_pf_reweight_resample_lds_bytes(int M, int mode);

/* ABI 39.  K1 for few trajectories ("for N < 256 split M across a WG-cluster"): with `enabled` != 0, plain
 * systematic resampling (mode 1) of N <= CUs / 2 trajectories of M = M_out <= 4096 particles (M a multiple of 512, d = 2 or 3)
 * gives every trajectory a CLUSTER of workgroups that meet twice through L2 (csrc/pf_resample_cluster.inc).  Same bits as
 * the one-workgroup kernel -- ancestors, gathered particles and estimates (tests/test_gpu_kernels.py).  OFF by default: measured
 * no faster (11.6 against 11.2 us per launch at 32 x 4096, profiles/r06/k1_cluster_ab.txt -- K1 at these sizes is one
 * thread's instruction chain plus memory round trips, which more workgroups do not shorten). */
void mmf_pf_set_resample_cluster(int enabled);
int mmf_pf_get_resample_cluster(void);

/* ---------------------------------------------------------------- K2: per-particle networks
 * The reference evaluates its dynamics and measurement MLPs as ~20 stock nn.Linear launches
 * over R = N*M rows, materialising (R, 64) activations after every layer and repeating the
 * control / observation features per particle (crossmodal/door_models/dynamics.py:102-134,
 * door_models/pf.py:63-107, push mirrors).  Both networks share one shape:
 *
 *   enc  : Linear(d -> 64), ReLU, ResLinear(64)                (layers.py:11-24)
 *   join : Linear(64*(1+k) -> 64) on cat(per-trajectory features, enc)  [+ ReLU for the
 *          measurement model]; the per-trajectory half is hoisted: the caller passes
 *          traj_bias (N, 64) = W[:, traj cols] @ traj_features + b
 *   res  : n_res x ResLinear(64)                               (3 dynamics, 2 measurement)
 *   head : Linear(64 -> n_out)                                 (d+1 dynamics, 1 measurement)
 *
 * The kernels keep every activation in registers: a wave owns 64 particles as the N
 * (lane) dimension of v_mfma_f32_32x32x2_f32 tiles, the accumulator tile of one layer is
 * the B operand of the next with no data movement, and a network's weights (<= 149 KiB,
 * fragment-ordered by mmf_pack_particle_net) live in LDS for the whole launch.
 */
typedef struct MmfParticleNetDesc {
  int32_t d_in;             /* state_dim: 1..4                                            */
  int32_t n_res;            /* residual blocks after the join layer: 0..MMF_MAX_RES       */
  int32_t relu_after_join;  /* 0 dynamics (dynamics.py:29-35), 1 measurement (pf.py:54-55) */
  int32_t n_out;            /* head width: 1..MMF_MAX_STATE_DIM+1                         */
  int32_t join_in;          /* row length of w_join (128, 192 or 256)                     */
  int32_t join_state_off;   /* first column of the state-feature block inside w_join      */
  const float* w_in;        /* (64, d_in)   enc Linear                                    */
  const float* b_in;        /* (64)                                                       */
  const float* w_enc[2];    /* (64, 64) x2  enc ResLinear block1, block2                  */
  const float* b_enc[2];    /* (64) x2                                                    */
  const float* w_join;      /* (64, join_in); only columns [off, off+64) are packed       */
  const float* w_res[2 * MMF_MAX_RES]; /* (64,64): block1, block2 of each residual block  */
  const float* b_res[2 * MMF_MAX_RES];
  const float* w_head;      /* (n_out, 64)                                                */
  const float* b_head;      /* (n_out)                                                    */
} MmfParticleNetDesc;       /* host struct holding device pointers (torch Linear layout)  */

/* Number of floats of a packed network blob with `n_res` residual blocks. */
size_t mmf_particle_net_floats(int n_res);

/* Re-order a network's weights into MFMA-fragment order (device -> device) for `precision`
 * (MMF_PREC_*); the blob size does not depend on it, the blob must be used with the same value. */
int mmf_pack_particle_net(const MmfParticleNetDesc* desc /* host */, float* packed, int precision,
                          void* stream);

/* x' = x + dir(x) * sigmoid(gate(x)) + L eps      (dynamics.py:102-134 + the reparameterised
 * MultivariateNormal(loc, scale_tril).rsample() of torchfilter's PF step)
 *  packed      blob from mmf_pack_particle_net (n_out must be d+1)
 *  states_in   (N*M, d)
 *  traj_bias   (N, 64)      hoisted control half of shared_layers[0] (+ its bias)
 *  noise       (N*M, d) standard normal, or null (EKF predict / open-loop rollouts)
 *  scale_tril  (d, d) row-major lower-triangular, shared by all rows (ignored if noise null)
 *  states_out  (N*M, d)     may alias states_in
 *  range_flag  int32 on the device or null; MMF_PREC_F16X3 ORs 1 into it when an activation
 *              exceeded the f16-split range (|x| >= 65504: the result is finite but wrong --
 *              re-run with MMF_PREC_F32); never written otherwise
 */
int mmf_pf_dynamics(const float* packed, int n_res, int precision, const float* states_in,
                    const float* traj_bias, const float* noise, const float* scale_tril,
                    float* states_out, int32_t* range_flag, int N, int M, int d, void* stream);

/* One modality's log-likelihood and its crossmodal combination
 * (pf.py:63-107 + base_models/crossmodal_pf.py:106-139):
 *      ll = net(x) + (modality_logw ? modality_logw[n * logw_stride] : 0)
 *      combine 0: loglik = ll          combine 1: loglik = log(exp(loglik) + exp(ll))
 * Calling it once per enabled modality with combine = 0, 1, 1, ... gives
 * logsumexp_k(log beta_k + ll_k).
 *  states (N*M, d); traj_bias (N, 64); loglik (N*M) in/out
 */
int mmf_pf_measure(const float* packed, int n_res, int precision, const float* states, const float* traj_bias,
                   const float* modality_logw, int logw_stride, float* loglik, int combine,
                   int32_t* range_flag, int N, int M, int d, void* stream);

/* `count` modalities' own log-likelihoods (combine 0 each) of the SAME particles in one launch (blockIdx.y = modality):
 * the training forward keeps every modality's log-likelihood for the backward's softmax over modalities
 * (base_models/crossmodal_pf.py:106-139), and at the reference's training size a launch is one tile per wave.
 *  packed, traj_bias, modality_logw (entries may be null), loglik: HOST arrays of `count` device pointers */
int mmf_pf_measure_multi(const float* const* packed, int count, int n_res, int precision, const float* states,
                         const float* const* traj_bias, const float* const* modality_logw, int logw_stride,
                         float* const* loglik, int32_t* range_flag, int N, int M, int d, void* stream);

/* Forward-mode Jacobian of the dynamics network (replaces torchfilter's default autograd
 * DynamicsModel.jacobian: batch replicated d times + one autograd.grad):
 *  states_in (N, d), traj_bias (N, 64) -> states_out (N, d), jac (N, d, d), jac[n][i][j] = d x'_i / d x_j
 *  precision: of `packed` (mmf_pack_particle_net); f16x3: the tangent columns are operands like any
 *  other (signed: their splits mask the sign for the range tracking), range_flag as in mmf_pf_dynamics
 */
int mmf_dynamics_jacobian(const float* packed, int n_res, int precision, const float* states_in,
                          const float* traj_bias, float* states_out, float* jac, int32_t* range_flag,
                          int N, int d, void* stream);

/* K independent Jacobian problems of one shape in ONE launch (the sub-filters of a fused EKF):
 * states_in / states_out (K, N, d), jac (K, N, d, d); packed[k], traj_bias[k] per problem.
 */
int mmf_dynamics_jacobian_multi(const float* const* packed, int n_res, int precision,
                                const float* states_in, const float* const* traj_bias,
                                float* states_out, float* jac, int32_t* range_flag, int K, int N,
                                int d, void* stream);

/* ---------------------------------------------------------------- K6: training through the per-particle networks
 * Replaces autograd through DoorDynamicsModel*.forward / DoorMeasurementModel.forward over N*M
 * rows (door_models/dynamics.py:102-134, door_models/pf.py:63-107; callers train_helpers.py via
 * torchfilter.train.*).  Exact fp32 (MMF_PREC_F32 blobs).
 *
 * forward:  out (R, NOUT) = head(network(states, traj_bias)) including the head bias, BEFORE the
 *           gate / noise / log-weight epilogue of mmf_pf_dynamics / mmf_pf_measure (NOUT = d + 1
 *           for kind 0 = dynamics, 1 for kind 1 = measurement), and
 *           stash (NL + 1, R, 64), NL = 3 + 2 n_res: [l] = input of 64x64 layer l, [NL] = head input.
 * backward: packed_t is a blob packed from the TRANSPOSED 64x64 layer weights (the state part of
 *           the join layer as a (64, 64) matrix; biases / first layer / head are ignored);
 *           head_w (NOUT, 64).  Writes dz (NL + 1, R, 64): [l] = gradient w.r.t. layer l's
 *           pre-activation output, [NL] = w.r.t. the first (d -> 64) layer's pre-activation,
 *           and d_states (R, d) = dz[NL] W_in (W_in is read from packed_t's first-layer section,
 *           which must hold the untransposed (64, d) weight).
 * Every reduction over particles is then a GEMM / column sum over the two stashes:
 *   dW_l = dz[l]^T stash[l], db_l = sum_rows dz[l], d traj_bias[n] = sum_m dz[2][n, m],
 *   dW_in = dz[NL]^T [states, 1], dW_head = d_out^T stash[NL].
 */
/* `mask` (NL + 1, R, 2) uint32: the sign bits of the stashed activations (bit 16 t + r of word h = feature
 * 32 t + (r & 3) + 8 (r >> 2) + 4 h), written by the forward, read by the backward's ReLU masks -- 8 B per
 * particle and layer where round 2 re-read the 256-B activation. */
int mmf_particle_net_train_forward(const float* packed, int n_res, int kind, const float* states,
                                   const float* traj_bias, float* stash, uint32_t* mask, float* out, int N, int M,
                                   int d, void* stream);
int mmf_particle_net_train_backward(const float* packed_t, const float* head_w, int n_res, int kind,
                                    const uint32_t* mask, const float* d_out, float* dz, float* d_states,
                                    int R, int d, void* stream);

/* dW_l = dz[l]^T stash[l] and db_l = column sums of dz[l] for l < n_layers, as per-slice partial
 * sums: partial_w (n_layers, n_splits, 64, 64), partial_b (n_layers, n_splits, 64); the caller
 * adds the n_splits slices (no float atomics).  dz, stash: (>= n_layers, R, 64).
 */
int mmf_particle_net_weight_grads(const float* dz, const float* stash, float* partial_w,
                                  float* partial_b, int n_layers, int R, int n_splits, void* stream);

/* The same, ADDING into the caller's partial sums when accumulate != 0 (each (layer, split) block of the
 * partials belongs to one workgroup: no atomics) -- the native training loop sums over time steps in place. */
int mmf_particle_net_weight_grads_acc(const float* dz, const float* stash, float* partial_w, float* partial_b,
                                      int n_layers, int R, int n_splits, int accumulate, void* stream);

/* The narrow reductions of the same backward in one pass over the rows (as GEMMs they are the
 * library's worst shapes, as torch reductions five passes over (R, 64) tensors):
 *   first layer   dW0[c][i]  = sum_r dz_first[r][c] states[r][i]          (64, d)
 *   head          dWh[o][c]  = sum_r d_out[r][o] h_last[r][c]             (n_out, 64),  dbh[o] = sum_r d_out[r][o]
 *   join layer    d traj_bias[n][c] = sum_m dz_join[n M + m][c]           (N, 64)
 * as per-slice partials over `n_slices` slices of every trajectory's M rows: p_first (N n_slices, 64, 4),
 * p_head (N n_slices, 4, 64), p_dout (N n_slices, 4), p_traj (N n_slices, 64); the caller adds the
 * slices.  dz_first, dz_join, h_last: (N M, 64); states (N M, d); d_out (N M, n_out); d, n_out <= 4.
 */
int mmf_particle_net_small_grads(const float* dz_first, const float* dz_join, const float* h_last,
                                 const float* states, const float* d_out, float* p_first, float* p_head,
                                 float* p_dout, float* p_traj, int N, int M, int d, int n_out,
                                 int n_slices, void* stream);

/* ---------------------------------------------------------------- K4: image encoder
 * Replaces observation_image_layers (crossmodal/door_models/layers.py:43-63;
 * push_models/layers.py:91-104, default variant): Conv 1->32 k5, ReLU, ResConv 32 k3,
 * Conv 32->16 k3, ReLU, Conv 16->8 k3, Flatten, Linear 8192->64, ReLU, ResLinear 64 --
 * as implicit-GEMM convolutions on v_mfma_f32_16x16x4_f32 with bias / skip / ReLU fused,
 * one launch per layer for ALL encoders of a step (they share the image, not the weights).
 */
typedef struct MmfImageEncoderDesc {
  const float* conv_w[5];   /* torch layouts: (32,1,5,5) (32,32,3,3) (32,32,3,3) (16,32,3,3) (8,16,3,3) */
  const float* conv_b[5];
  const float* fc_w;        /* (64, 8192) */
  const float* fc_b;        /* (64) */
  const float* res_w[2];    /* ResLinear(64): block1, block2 (64,64) */
  const float* res_b[2];
  int32_t variant;          /* MMF_ENCODER_DEFAULT, or MMF_ENCODER_SPANNING_POOL (the push virtual   */
                            /* sensor's stack, push_models/layers.py:43-65,77-90): conv_w[4] is     */
                            /* (2,16,3,3), then a full-height and a full-width average pool (2-wide) */
                            /* give 64 values and fc_w is (64, 64)                                   */
} MmfImageEncoderDesc;      /* host struct holding device pointers */
#define MMF_ENCODER_DEFAULT 0
#define MMF_ENCODER_SPANNING_POOL 1

size_t mmf_image_encoder_floats(void);                 /* floats of one packed encoder blob */
size_t mmf_image_encoder_workspace_bytes(int n_images, int n_nets);
int mmf_pack_image_encoder(const MmfImageEncoderDesc* desc /* host */, float* packed, void* stream);

/*  packed     host array of n_nets (1..4) device pointers to packed encoder blobs
 *  images     (N, 32, 32)        shared by every encoder
 *  feat       (n_nets, N, 64)    out
 *  workspace  >= mmf_image_encoder_workspace_bytes(N, n_nets) bytes of device memory
 *  range_flag int32 on the device or null: MMF_PREC_F16X3 ORs 1 into it when an activation
 *             left the f16-split range (see mmf_pf_dynamics)
 *  variant    MMF_ENCODER_*: the architecture every blob of this call was packed for
 *  precision  MMF_PREC_F32: every layer on the f32 MFMA, one launch per layer.  MMF_PREC_F16X3: split-f16
 *             products; stem + conv 32->32 and conv 32->32 + skip + conv 32->16 run as two fused,
 *             persistent kernels with the activations in LDS (csrc/image_encoder_fused.inc).
 *             MMF_PREC_BF16: the same two fused kernels with single bf16 products (94 % of the
 *             MACs); conv 16->8 and the linear tail stay f16x3.
  * f16x3 products (the default mode, and conv 16->8 + the linear layer of the bf16 mode): weights are
 * held as f16 fragments of 256 w (the split's "lo" half stays out of the f16 subnormals); |w| < 255.
 */
int mmf_image_encoder(const float* const* packed, int n_nets, const float* images, float* feat,
                      void* workspace, int32_t* range_flag, int precision, int variant, int N,
                      void* stream);

/* K6 for the image encoder's convolution stack (door_models/layers.py:43-58) -- what torch
 * autograd / MIOpen do under torchfilter.train.* (crossmodal/train_helpers.py:76-162).  One
 * encoder per call, exact fp32, MMF_ENCODER_DEFAULT only.
 *  mmf_image_convs_train_forward   images (N,32,32) -> a1 (N,32,32,32) stem, h (N,32,..) ResConv hidden,
 *                                  a2 (N,32,..) ResConv out, a3 (N,16,..), a4 (N,8,32,32) = input of the
 *                                  flatten + linear; `packed` = mmf_pack_image_encoder's blob.  precision
 *                                  MMF_PREC_F32: exact fp32 products; MMF_PREC_BF16 (BASELINE config 5's
 *                                  "bf16 measurement CNN"): the two 32->32 convolutions with bf16 products,
 *                                  conv 32->16 / 16->8 f16x3, stem fp32; MMF_PREC_F16X3: f16x3 throughout.
 *                                  The backward differentiates the exact network through the saved
 *                                  (rounded-forward) activations
 *  mmf_image_convs_train_backward  g_a4 -> pre-activation gradients g3 (N,16,..), g2, gh, g1 (N,32,..):
 *                                  the forward conv kernel on transposed + flipped weights
 *                                  (`packed_bwd` = mmf_pack_image_convs_backward), ReLU masks fused
 *  mmf_conv_weight_grads           dW[tap][co][ci] partials of one 3x3 layer from its output gradient g
 *                                  (N,co,32,32) and input act (N,ci,32,32): (co,ci) in {(32,32),(16,32),(8,16)};
 *                                  partial (n_blocks, 9, 32, 32) and the bias gradient's partial_b
 *                                  (n_blocks, 32) = sums of g per channel, both summed over dim 0 by the
 *                                  caller (one slot per workgroup; a workgroup walks half-images, so
 *                                  n_blocks <= 2N, 256 fills the chip).
 *                                  (co,ci) = (32,1): the 5x5 stem, act = images (N,32,32); the head of each
 *                                  partial slot holds dW1 as [co 32][tap 32] (25 taps live).
 *                                  dw (co, ci, k, k) / db (co), when given: the slots summed in ascending order into
 *                                  nn.Conv2d's own layout by a second launch (round 5; null: the caller sums)
 */
size_t mmf_image_convs_backward_floats(void);
int mmf_pack_image_convs_backward(const MmfImageEncoderDesc* desc, float* packed_bwd, void* stream);
int mmf_image_convs_train_forward(const float* packed, const float* images, float* a1, float* h,
                                  float* a2, float* a3, float* a4, int32_t* range_flag, int precision,
                                  int N, void* stream);
int mmf_image_convs_train_backward(const float* packed_bwd, const float* a1, const float* h,
                                   const float* a2, const float* a3, const float* g_a4, float* g1,
                                   float* gh, float* g2, float* g3, int N, void* stream);
int mmf_conv_weight_grads(const float* g, const float* act, float* partial, float* partial_b, int N, int co,
                          int ci, int n_blocks, float* dw, float* db, void* stream);

/* ABI 39.  mmf_image_convs_train_backward with the three layers that have 32 output channels (97 % of the MACs) on
 * v_mfma_f32_32x32x16_f16, three products per product: every gradient tensor is scaled by the power of two that brings its
 * largest magnitude to [2^14, 2^15) before its operand split and the result scaled back; masks, skip path and outputs as the
 * exact-fp32 form.  conv 16->8's gradient stays exact.  scratch (4 floats, device): max |g3|, max |g2|, max |gh|, max |g_a4| on
 * return -- reduced on the device (each launch leaves the next one's) -- the `g_absmax` words of mmf_conv_weight_grads_h. */
int mmf_image_convs_train_backward_h(const float* packed_bwd, const float* a1, const float* h,
                                     const float* a2, const float* a3, const float* g_a4, float* g1,
                                     float* gh, float* g2, float* g3, float* scratch, int N, void* stream);

/* ABI 39.  The same gradients of a 3x3 layer (co, ci) = (32, 32) | (16, 32) | (8, 16) on v_mfma_f32_32x32x16_f16 with the
 * three-product f16 split of both operands (csrc/image_encoder_train_h.inc): 54 MFMAs of 32 cycles per image row instead of
 * 144 of 64.  g_absmax: DEVICE scalar, the largest |g| (the caller reduces it first: no host read) -- g is multiplied by the
 * power of two that brings it to [2^14, 2^15) and the sums are multiplied back, so gradients of any magnitude keep their
 * leading bits; db is the exact fp32 sum.  range_flag (or null): OR-ed with 1 if an activation left the f16 range. */
int mmf_conv_weight_grads_h(const float* g, const float* act, const float* g_absmax, float* partial, float* partial_b,
                            int32_t* range_flag, int N, int co, int ci, int n_blocks, float* dw, float* db, void* stream);

/* ---------------------------------------------------------------- particle-filter step loop
 * Replaces the Python loop of torchfilter's Filter.forward_loop (call site
 * crossmodal/eval_helpers.py:139-142) for the fused models: one call enqueues the kernels of
 * all T steps (mmf_pf_dynamics, mmf_pf_measure per modality, mmf_pf_reweight_resample) on
 * `stream`.  Per-trajectory terms and randomness are indexed by step: row block t of every
 * (T*N, ...) array belongs to step t.
 */
#define MMF_LOOP_MAX_MEAS 4
typedef struct MmfPfLoopArgs {
  int32_t T, N, M, d;
  int32_t n_meas;            /* measurement networks (modalities), 1..MMF_LOOP_MAX_MEAS       */
  int32_t resample_mode;     /* 0 none, 1 systematic, 2 multinomial                           */
  int32_t precision;         /* MMF_PREC_*                                                    */
  int32_t n_res_dyn, n_res_meas;
  int32_t logw_stride;       /* row stride of the modality log-weight arrays                  */
  const float* dyn_packed;   /* packed dynamics network                                       */
  const float* dyn_bias;     /* (T*N, 64)                                                     */
  const float* meas_packed[MMF_LOOP_MAX_MEAS];
  const float* meas_bias[MMF_LOOP_MAX_MEAS];   /* (T*N, 64) each                              */
  const float* meas_logw[MMF_LOOP_MAX_MEAS];   /* first element of modality k's log-weight    */
                                               /* column in a (T*N, logw_stride) array, or null */
  const float* noise;        /* (T, N, M, d) standard normal                                  */
  const float* scale_tril;   /* (d, d)                                                        */
  const float* uniforms;     /* (T, N) systematic / (T, N, M) multinomial / null              */
  float* states_a;           /* (N, M, d) belief on entry                                     */
  float* states_b;           /* (N, M, d) scratch                                             */
  float* logw_a;             /* (N, M) log-weights on entry                                   */
  float* logw_b;             /* (N, M) scratch                                                */
  float* loglik;             /* (N, M) scratch                                                */
  float* estimates;          /* (T, N, d) out                                                 */
  int32_t* range_flag;       /* device int32 or null (see mmf_pf_dynamics)                    */
  int32_t* final_location;   /* HOST int32 out or null: bit 0 belief states in states_b,      */
                             /* bit 1 log-weights in logw_b                                   */
  void* const* events;       /* HOST array of hipEvent_t or null: recorded around every launch */
                             /* of the sampled steps, [sample][dynamics, measure.., resample][start,end] */
  int32_t event_stride;      /* steps t with t % stride == stride / 2 are sampled (<= 1: every  */
                             /* step); `events` holds 2*(2+n_meas) entries per sampled step    */
  float* loglik_steps;       /* (T, N, M) or null: step t's fused log-likelihoods are kept here  */
                             /* instead of the shared `loglik` scratch (parity certificates)     */
  int32_t* indices_steps;    /* (T, N, M) int32 or null: ancestors drawn at every resampling step */
  uint64_t noise_seed;       /* noise_mode 2: key of the counter-based generator (include/mmf_philox.h)       */
  uint32_t noise_step0;      /* noise_mode 2: step t of the loop draws counter step noise_step0 + t          */
  uint32_t noise_traj0;      /* noise_mode 2: global index of this shard's first trajectory                  */
  int32_t noise_mode;        /* 0: `noise` tensor; 2: generated inside the dynamics kernel (`noise` may be null) */
  float soft_alpha;          /* 0 or 1: plain resampling; 0 < alpha < 1 (resample_mode != 0): torchfilter's soft      */
                             /* resampling (mmf_pf_reweight_resample_soft) -- the survivors carry importance weights, */
                             /* so every step reads and writes the log-weights                                        */
  int32_t estimate_argmax;   /* != 0: estimates = the particle with the largest pre-resampling weight                 */
                             /* (mmf_pf_argmax_estimate; torchfilter's estimation_method = "argmax")                  */
  float* estimate_scratch;   /* (N, d), needed with estimate_argmax: K1's weighted mean lands here instead            */
  int32_t persistent;        /* != 0: the whole loop as ONE persistent launch (small problems: mmf_pf_persistent_plan > 0; */
                             /* plain systematic resampling, weighted-average estimates, no events / per-step records).    */
                             /* Same results, bit for bit, as the launch-per-step loop.                                    */
  int32_t n_sync_words;      /* 4-byte words of sync_words (>= mmf_pf_persistent_sync_words(N, M, d, n_meas))              */
  uint32_t* sync_words;      /* persistent: device workspace of the in-launch hand-offs -- tagged 8-byte granules of the   */
                             /* particle rows and log-likelihoods --, zeroed by the call; an allocation of its own         */
} MmfPfLoopArgs;             /* host struct holding device pointers                           */

int mmf_pf_forward_loop(const MmfPfLoopArgs* args /* host */, void* stream);

/* The persistent form of the step loop (MmfPfLoopArgs.persistent): at the sizes the reference itself runs (32
 * trajectories x 300 particles: door_models/pf.py:24-27, eval_helpers.py:125-142) a step is bound by the fixed cost
 * of its four launches; one launch whose workgroups keep ONE network's weights in LDS for all T steps and hand
 * particles over through L2 (tagged 8-byte granules: the data is the flag; no grid barrier) removes it.
 *   mmf_pf_persistent_plan: number of workgroups it would use (0: not eligible -- too many tiles per wave, or
 *   M > 2048; MMF_EINVAL: invalid sizes), and optionally the workgroups per network role, K1 workgroups and tiles
 *   per trajectory.                                                                                               */
int mmf_pf_persistent_plan(int N, int M, int n_meas, int* G_out, int* GK_out, int* tiles_per_trajectory_out);
size_t mmf_pf_persistent_sync_words(int N, int M, int d, int n_meas);

/* estimation_method = "argmax" of torchfilter's ParticleFilter: per trajectory the particle with
 * the largest pre-resampling log-weight logw_in + loglik (logw_in null: uniform), first index on ties (torch.argmax).
 *   loglik (N, M), logw_in (N, M) | null, states (N, M, d) -> estimate (N, d)                                  */
int mmf_pf_argmax_estimate(const float* loglik, const float* logw_in, const float* states, float* estimate,
                           int N, int M, int d, void* stream);

/* Counter-based process noise (include/mmf_philox.h: Philox4x32-10 + a fixed-order Box-Muller, a pure
 * function of (seed, step, global trajectory index, particle)): replaces the
 * MultivariateNormal(...).rsample() torchfilter draws inside every step without a
 * (T, N, M, d) tensor.  mmf_pf_dynamics_philox = mmf_pf_dynamics with the noise generated in the kernel's
 * epilogue; mmf_philox_normals materialises the same draws (N, M, d) (initial particles, step-by-step use,
 * tests); mmf_philox_uniforms the resampling uniforms (T, N) of steps step0 .. step0 + T - 1.
 */
int mmf_pf_dynamics_philox(const float* packed, int n_res, int precision, const float* states_in,
                           const float* traj_bias, unsigned long long seed, unsigned step, unsigned traj0,
                           const float* scale_tril, float* states_out, int* range_flag, int N, int M, int d,
                           void* stream);
int mmf_philox_normals(unsigned long long seed, unsigned step, unsigned traj0, float* out, int N, int M, int d,
                       void* stream);
int mmf_philox_uniforms(unsigned long long seed, unsigned step0, unsigned traj0, float* out, int T, int N,
                        void* stream);


/* Open-loop rollout of the dynamics model: replaces torchfilter's DynamicsModel.forward_loop (external
 * dependency; call sites crossmodal/eval_helpers.py:135-137, scripts/door_task/eval_dynamics.py:36-38):
 *     x_t = f(x_{t-1}, u_t),  t = 1 .. T      (no process noise: the mean prediction)
 * as T launches of the K2 dynamics kernel enqueued by one C call, each reading the previous step's row
 * of the output.
 *  packed     packed dynamics network (mmf_pack_particle_net)
 *  x0         (N, d) initial states
 *  traj_bias  (T, N, 64) hoisted control term of every step (one K7 launch over the T*N controls)
 *  out        (T, N, d) predicted states
 */
int mmf_dynamics_forward_loop(const float* packed, int n_res, int precision, const float* x0,
                              const float* traj_bias, float* out, int32_t* range_flag, int T, int N, int d,
                              void* stream);

/* ---------------------------------------------------------------- K6, fused: one network call of the training backward
 * Recompute (the forward pass's f16x3 arithmetic), backward data path and weight / bias gradients of ONE per-particle
 * network over N * M rows in one kernel (the dynamics network: three launches -- encoder forward, trunk, encoder
 * backward), replacing mmf_particle_net_train_forward + _backward + _weight_grads of the reference's training step
 * (/root/reference/crossmodal/train_helpers.py:124-162 over door_models/dynamics.py:102-134, door_models/pf.py:63-107).
 *  packed_dual  mmf_pack_particle_net(.., MMF_PREC_F16X3_DUAL)
 *  kind         0 dynamics (n_res 3): g_next (N M, d) = dL / d x' in, d_raw (N M, d + 1) = dL / d (dir, gate) out,
 *               act / g_act scratch (N M, 64) fp32;  1 measurement (n_res 2): d_out (N M) = dL / d log-likelihood
 *  d_states     out (N M, d): the gradient THROUGH the network (the dynamics' direct path x' = x + .. is the caller's),
 *               plus d_states_base when that is given
 *  dz_first_h, sc_first, dz_join_h, sc_join, h_last_h: out, the compact (N M, 64) f16 rows + (N M) fp32 row scales the
 *               narrow reductions read (first-layer / join pre-activation gradients, head input)
 *  pw, pb       (NL, n_slots, 64, 64) / (NL, n_slots, 64) ACCUMULATED in place, slot = workgroup (grid =
 *               min(n_slots, 256, ceil(N M / 128))): zero them before the first call, sum over slots after the last
 */
typedef struct MmfTrainFusedArgs {
  const float* packed_dual;
  int32_t n_res, kind, d, N, M, n_slots;
  const float* states;
  const float* traj_bias;
  const float* d_out;
  const float* g_next;
  float* d_raw;
  float* act;
  float* g_act;
  float* d_states;
  void* dz_first_h;
  float* sc_first;
  void* dz_join_h;
  float* sc_join;
  void* h_last_h;
  float* pw;
  float* pb;
  const float* d_states_base; /* null, or (N M, d) values ADDED to what is written to d_states (may be d_states itself: the
                                 caller's running gradient is updated in place, no separate add pass) */
} MmfTrainFusedArgs;

int mmf_particle_net_train_fused(const MmfTrainFusedArgs* args /* host */, void* stream);
/* The same for several MEASUREMENT networks of one step in ONE launch (blockIdx.y = network; they differentiate
 * independently of each other): same d, rows, depth and n_slots, each with its own outputs and row slots.  At the
 * reference's training size (960 rows) a call is one tile per wave -- its latency -- whatever runs beside it. */
int mmf_particle_net_train_fused_multi(const MmfTrainFusedArgs* nets /* host, n */, int n, void* stream);

/* ---------------------------------------------------------------- K6: the particle filter's training recursion
 * One C call for the forward recursion of a train-mode (no resampling) particle filter over T steps and
 * one for its backward -- the caller /root/reference/crossmodal/train_helpers.py:124-162
 * (torchfilter.train.train_filter: forward_loop over a subsequence, MSE, backward) at any size, including
 * the reference's own 32 x 30 particles x 16 steps where per-launch host work dominates.
 *
 * forward (per step t):  x_t = f(x_{t-1}) + L eps_t  (K2 dynamics);  ll = logsumexp_k(ll_k(x_t) + beta_k)
 *   (K2 measurement);  logw_t = logw_{t-1} + ll - logsumexp,  est_t = sum_m exp(logw_t) x_t  (K1 mode 0).
 *   Only the particle sets states (T+1, N, M, d) and log-weights logw (T+1, N, M) are kept.
 * backward (t = T-1 .. 0): the K6 kernels on RECOMPUTED activations -- the step's stashes are rebuilt by
 *   mmf_particle_net_train_forward from states[t] / states[t+1] for `chunk_traj` trajectories at a time,
 *   into ONE reused pair of buffers (stash, dz) + the ReLU sign bits; weight-gradient partials accumulate in
 *   place over steps and chunks.  Peak memory: the chunk buffers + (T+1) particle sets, whatever T.
 *   (Chunks small enough to keep stash + dz inside the 256 MiB Infinity Cache were measured and lose to
 *   large ones -- DESIGN.md, K6: the caller picks chunk_traj for memory, not for cache residency.)
 *
 * Buffers (device, fp32, caller-owned).  K = n_meas networks (modalities), NLd = 3 + 2 n_res_dyn, NLm likewise.
 *  in:   dyn_* / meas_*[k]: packed      forward blob (mmf_pack_particle_net, `precision` for the forward pass),
 *                           packed_f32  exact-f32 blob (recompute), packed_t  blob of the transposed layers,
 *                           head_w (n_out, 64)
 *        dyn_bias (T, N, 64), meas_bias[k] (T, N, 64), meas_logw[k] = &beta[0][0][k] of a (T, N, logw_stride)
 *        array or null, noise (T, N, M, d), scale_tril (d, d), g_estimates (T, N, d)  [backward]
 *  io:   states (T+1, N, M, d), logw (T+1, N, M): slot 0 = initial belief (caller), slot t+1 = after step t
 *  out:  estimates (T, N, d)  [forward]
 *        backward: d_states0 (N, M, d), d_logw0 (N, M);
 *        per network: pw (NL+1, n_splits, 64, 64), pb (NL+1, n_splits, 64) weight / bias partials summed over
 *        steps; p_first (T, N n_slices, 64, 4), p_head (T, N n_slices, 4, 64), p_dout (T, N n_slices, 4),
 *        p_traj (T, N n_slices, 64): the narrow reductions per step (the caller adds slices / steps;
 *        d traj_bias[t][n] = sum_s p_traj[t][n][s], d beta_k[t][n] = sum_s p_dout_k[t][n][s][0])
 *  kept:  ll_steps (T, K, N, M) per-modality log-likelihoods ll_k = raw_k + b_k + beta_k of every step
 *  scratch: stash, dz (max(NLd, NLm) + 1, chunk_traj M, 64); raw (chunk_traj M, 8); d_raw (chunk_traj M, 8);
 *        loglik (N, M); g_states_a, g_states_b (N, M, d); g_logw_a, g_logw_b (N, M); d_tmp (chunk_traj M, d)
 */
typedef struct MmfTrainNet {
  const float* packed;      /* forward pass blob                                   */
  const float* packed_f32;  /* recompute blob (MMF_PREC_F32)                        */
  const float* packed_t;    /* transposed layers (backward data path)              */
  const float* head_w;      /* (n_out, 64)                                         */
  float* pw;                /* (NL+1, n_splits, 64, 64)                             */
  float* pb;                /* (NL+1, n_splits, 64)                                 */
  float* p_first;           /* (T, N n_slices, 64, 4)                               */
  float* p_head;            /* (T, N n_slices, 4, 64)                               */
  float* p_dout;            /* (T, N n_slices, 4)                                   */
  float* p_traj;            /* (T, N n_slices, 64)                                  */
  const float* packed_dual; /* ABI 37, fused = 1: MMF_PREC_F16X3_DUAL blob            */
} MmfTrainNet;

typedef struct MmfPfTrainArgs {
  int32_t T, N, M, d, n_meas, n_res_dyn, n_res_meas, logw_stride, precision;
  int32_t chunk_traj, n_splits, n_slices;
  MmfTrainNet dyn;
  MmfTrainNet meas[MMF_LOOP_MAX_MEAS];
  const float* dyn_bias;
  const float* meas_bias[MMF_LOOP_MAX_MEAS];
  const float* meas_logw[MMF_LOOP_MAX_MEAS];
  const float* noise;
  const float* scale_tril;
  const float* g_estimates;
  float* states;
  float* logw;
  float* estimates;
  float* d_states0;
  float* d_logw0;
  float* stash;
  uint32_t* mask;            /* scratch (max(NLd, NLm) + 1, chunk_traj M, 2): ReLU sign bits of the recomputed activations */
  float* dz;
  float* raw;
  float* d_raw;
  float* loglik;
  float* ll_steps;           /* (T, K, N, M): every step's per-modality log-likelihoods (forward -> backward) */
  float* g_states_a;
  float* g_states_b;
  float* g_logw_a;
  float* g_logw_b;
  float* d_tmp;
  int32_t* range_flag;
  float* dz_scale;           /* scratch: fp32 row scales of `dz` -- (max(NLd, NLm) + 1, chunk_traj M); fused: (sets, 2, chunk_traj M).
                                ABI 39: `stash` and `dz` are ALWAYS f16 arrays (activations as f16, pre-activation gradients as f16
                                relative to the largest magnitude of their 32-row tile, kept here; the weight-gradient products
                                of the f16 values are exact: f16 MFMA, fp32 accumulate).  The fp32 buffers of ABI 34 (`compact = 0`)
                                and the three-pass f16x3 recompute / backward of ABI 36 (`recompute_f16x3`, `backward_f16x3`) are
                                gone: superseded by `fused`, timings kept in profiles/r05/bench_train_fused_ab.txt */
  int32_t fused;             /* ABI 37.  1 (needs precision = MMF_PREC_F16X3): every network's recompute (in the forward pass's
                                own three-product f16 arithmetic, on `packed`), backward data path and weight gradients run as ONE
                                kernel per network call (mmf_particle_net_train_fused: layer inputs and pre-activation gradients
                                never reach HBM); each MmfTrainNet then carries `packed_dual`, and pw / pb are (NL, n_splits, 64, 64)
                                / (NL, n_splits, 64) partials, ZEROED by the caller, one slot per workgroup (n_splits <= 256 = the
                                largest grid).  stash / dz / dz_scale then only hold the three rows-by-64 slots the narrow
                                reductions read (layer slots 2 and NL).  fused_act, fused_g_act: scratch (chunk_traj M, 64) fp32.
                                0: three passes per network call with EXACT fp32 products (recompute on `packed_f32`, transposed
                                layers `packed_t` an MMF_PREC_F32 blob) over the f16 buffers -- the f32 engine mode's training path
                                and the cross-check of the fused kernel (MMF_TRAIN_FUSED=0) */
  float* fused_act;
  float* fused_g_act;
  int32_t fused_sets;        /* ABI 38.  >= n_meas > 1: stash / dz / dz_scale / d_tmp hold one set of row slots per
                                measurement network -- (sets, C, 64), (sets, 2, C, 64), (sets, 2, C), (sets, C, d) with
                                C = chunk_traj M -- and a step's measurement networks run as ONE launch
                                (mmf_particle_net_train_fused_multi).  0 / 1: one launch per network */
} MmfPfTrainArgs;

int mmf_pf_train_forward(const MmfPfTrainArgs* args /* host */, void* stream);
int mmf_pf_train_backward(const MmfPfTrainArgs* args /* host */, void* stream);

/* After mmf_pf_train_backward: one network's partial sums -> its parameter gradients in the order and layout of the
 * nn.Module parameters the reference's optimiser walks (train_helpers.py:124-162: torch autograd produces them per
 * parameter), plus the gradient of its hoisted per-trajectory term and of its modality log-weight column.
 *  grads      flat: w_in (64, d) | b_in (64) | encoder block1 w (64, 64), b | block2 w, b | join w (64, join_in: columns
 *             outside [join_state_off, +64) are zero) | per residual block: block1 w, b, block2 w, b | head w (n_out, 64) |
 *             head b (n_out)
 *  bias_grad  (T N, 64);  d_beta (T N, beta_stride) or null: column beta_col is written
 *  pw (NL + 1, S, 64, 64), pb (NL + 1, S, 64), p_first (T, N SL, 64, 4), p_head (T, N SL, 4, 64), p_dout (T, N SL, 4),
 *  p_traj (T, N SL, 64): MmfTrainNet's buffers (NL = 3 + 2 n_res); fused: the first-layer bias is column d of p_first
 *  scratch    32 x 516 floats */
typedef struct MmfPfTrainFinalizeArgs {
  int32_t T, N, SL, S, n_res, d, n_out, join_in, join_state_off, fused, beta_stride, beta_col;
  const float *pw, *pb, *p_first, *p_head, *p_dout, *p_traj;
  float *grads, *bias_grad, *d_beta, *scratch;
} MmfPfTrainFinalizeArgs;
int mmf_pf_train_finalize(const MmfPfTrainFinalizeArgs* args /* host */, void* stream);

/* ---------------------------------------------------------------- K7: per-trajectory MLP programs
 * The N-row networks around the filters (vector encoders layers.py:11-40,66-95; PF weight
 * model crossmodal_pf.py:74-106; virtual sensor kf.py:81-126; EKF weight model
 * crossmodal_kf.py:134-167; hoisted join-layer halves) as ONE launch per model: a list of
 * instructions interpreted per 16 rows by a workgroup.  Vectors (<= 128 wide) live in MMF_TRAJ_SLOTS
 * LDS slots.
 *   LOAD        slot[dst][0:out_dim] = io[io][row*io_stride + io_off + 0:out_dim]
 *   LINEAR      slot[dst] = act(W cat(slot[src[s]][src_off[s] : src_off[s]+src_dim[s]] ..) + b (+ slot[res]));  W is stored
 *               in `weights` at w_off as the A fragments of v_mfma_f32_16x16x4_f32: per source s, per group g of 16
 *               input columns (zero-padded), per tile mt of 16 outputs (out_pad = out_dim <= 64 ? 64 : 128):
 *               [g][mt][lane = 16 q + i][ks] = W[16 mt + i][first column of s + 16 g + 4 ks + q]; the bias (128
 *               floats, zero-padded) at b_off
 *   STORE       io[io][row*io_stride + io_off + 0:out_dim] = act(slot[src[0]])
 *   STORE_DIAG  io[io][row*io_stride + io_off + 0:d*d] = diag(act(slot[src[0]][0:d])), d = out_dim
 * Round 5 -- the reverse mode of a program is ANOTHER program over the same slot file (train_helpers.py:124-162: the
 * reference differentiates these networks with torch autograd, a GEMM + a reduction + an element-wise launch per
 * nn.Linear), built by trajprog.py from the forward list; it needs four more instructions, and LOAD / LINEAR write at
 * slot[dst][dst_off + ..] (LOAD applies `act`):
 *   MASK        slot[dst][dst_off + c] = io[..][c] > 0 ? slot[dst][dst_off + c] : 0     (backward of a ReLU, from the stashed output)
 *   ADD         slot[dst][dst_off + c] += slot[src[0]][src_off[0] + c]
 *   ZERO        slot[dst][dst_off + c] = 0
 *   LOAD_ADD    slot[dst][dst_off + c] += io[..][c]                     c < out_dim throughout
 * The transposed layers are LINEARs over a second blob; parameter gradients are mmf_traj_weight_grads over the rows
 * the two programs stashed.
 */
#define MMF_TRAJ_MAX_IO 8
#define MMF_TRAJ_SLOTS 8
#define MMF_TRAJ_LOAD 0
#define MMF_TRAJ_LINEAR 1
#define MMF_TRAJ_STORE 2
#define MMF_TRAJ_STORE_DIAG 3
#define MMF_TRAJ_MASK 4
#define MMF_TRAJ_ADD 5
#define MMF_TRAJ_ZERO 6
#define MMF_TRAJ_LOAD_ADD 7
#define MMF_TRAJ_ACT_NONE 0
#define MMF_TRAJ_ACT_RELU 1
#define MMF_TRAJ_ACT_SIGMOID 2
#define MMF_TRAJ_ACT_SQRT_SQ_PLUS 3   /* sqrt(x*x + fparam): kf.py:117-126 */

typedef struct MmfTrajInstr {
  int32_t op, dst;
  int32_t src[4];       /* source slots, -1 = unused */
  int32_t src_off[4];   /* first feature read from each source slot (multiple of 4) */
  int32_t src_dim[4];
  int32_t out_dim;
  int32_t w_off, b_off; /* float offsets into `weights`; b_off = -1: no bias */
  int32_t res;          /* residual slot or -1 */
  int32_t act;
  int32_t io, io_stride, io_off;
  float fparam;
  int32_t dst_off;      /* first feature written in slot[dst] (multiple of 4; LINEAR reads its residual there too) */
} MmfTrajInstr;

/*  prog     (n_instr) MmfTrajInstr on the DEVICE
 *  weights  device blob the instructions index
 *  io       HOST array of MMF_TRAJ_MAX_IO device pointers (inputs and outputs; unused = null)
 *  R        rows
 *  n_slots  LDS vector slots the program uses (1 + its largest slot index, <= MMF_TRAJ_SLOTS)
 *  vec_width  64 when no vector of the program is wider, else 128: the launch sizes its LDS for
 *           n_slots x 16 rows x vec_width (4 .. 66 KiB per workgroup of four waves, one 16-row task at a time)
 */
int mmf_traj_program(const MmfTrajInstr* prog, int n_instr, const float* weights,
                     float* const* io, int R, int n_slots, int vec_width, void* stream);

/* The weight blob of a program, gathered on the device from the parameters where they lie (an optimiser updates them in
 * place, so the descriptor table is built once per program): per part
 *   MMF_TRAJ_PACK_LAYER       fragments of W[0:rows][col0 : col0 + dim] (W row-major with row stride ld) for one source
 *   MMF_TRAJ_PACK_TRANSPOSED  fragments of the transposed block: output o < rows is column col0 + o of W, input
 *                             k < dim its row k -- the layer the reverse program multiplies by
 *   MMF_TRAJ_PACK_BIAS        128 floats, the first `rows` from src
 * written at blob + dst_off in the LINEAR layout above (out_pad = 64 | 128). */
#define MMF_TRAJ_PACK_LAYER 0
#define MMF_TRAJ_PACK_TRANSPOSED 1
#define MMF_TRAJ_PACK_BIAS 2
typedef struct MmfTrajPackDesc {
  uint64_t src;          /* device address of the fp32 parameter */
  int32_t kind, rows, ld, col0, dim, out_pad;
  int32_t dst_off, reserved;
} MmfTrajPackDesc;
int mmf_traj_pack(const MmfTrajPackDesc* desc /* device */, int n_desc, float* blob, void* stream);

/* Parameter gradients of a program's LINEARs from the rows its forward (every LOADed / LINEAR output: `stash`) and
 * its reverse program (every pre-activation gradient: `dz`) wrote: per descriptor
 *   grads[grad_off + o * grad_ld + k] = sum_row dz[row][dz_col + o] * stash[row][x_col + k]   o < out_dim, k < x_dim
 *   grads[bias_off + o]               = sum_row dz[row][dz_col + o]                            (bias_off >= 0)
 * on v_mfma_f32_16x16x4_f32 with rows as the contraction, in a fixed order (row slices in ascending order: no atomics).
 * Replaces autograd's Linear backward (one GEMM and one column reduction per nn.Linear).
 *  desc      (n_desc) on the DEVICE;  stash (R, stash_ld), dz (R, dz_ld) fp32
 *  grads     flat output, every descriptor's block written (not accumulated)
 *  partials  (n_slices, n_grads) scratch when n_slices > 1 (rows are cut into n_slices equal runs, <= 64), else null
 */
typedef struct MmfTrajGradDesc {
  int32_t x_col, x_dim, dz_col, out_dim;
  int32_t grad_off, grad_ld, bias_off, reserved;
} MmfTrajGradDesc;
int mmf_traj_weight_grads(const MmfTrajGradDesc* desc, int n_desc, const float* stash, int stash_ld, const float* dz,
                          int dz_ld, float* grads, int n_grads, float* partials, int n_slices, int R, void* stream);

/* The 8192 -> 64 linear layer behind the convolutions of a training step (door_models/layers.py:59-60, the
 * nn.Linear(8 * 32 * 32, units) of the image encoder), forward and both backward products in exact fp32 on
 * v_mfma_f32_16x16x4_f32, fixed summation order; replaces the library GEMMs of torch's Linear forward / backward.
 *  x (R, K) fp32, w (64, K) row-major = nn.Linear.weight, b (64) or null, K % 512 == 0
 *   forward:  y (R, 64) = x w^T + b; partial: scratch (K / 512, R, 64) -- the K chunks meet in ascending order
 *   backward: dx (R, K) = g w (null: skipped);  dw (64, K) = g^T x;  db (64) = column sums of g (null: skipped) */
int mmf_fc64_train_forward(const float* x, const float* w, const float* b, float* y, float* partial, int R, int K,
                           void* stream);
int mmf_fc64_train_backward(const float* g, const float* x, const float* w, float* dx, float* dw, float* db, int R,
                            int K, void* stream);

/* ---------------------------------------------------------------- K3: EKF algebra + fusion
 * Replaces torchfilter's EKF predict/update (A S A^T + L L^T; K = S-(S- + R)^-1;
 * mu = mu- + K(z - mu-); S = (I-K)S-) for K sub-filters and the reference's
 * fusion of their beliefs: crossmodal (base_models/crossmodal_kf.py:153-167 via
 * utility.py:4-11) or unimodal information form (base_models/unimodal_kf.py:204-242).
 * One trajectory per lane, all d x d algebra in registers.
 *
 *  A        (K, N, d, d)  dynamics Jacobians
 *  mu_pred  (K, N, d)     predicted means
 *  q_tril   (K, d, d)     dynamics noise scale_tril per sub-filter (Q = L L^T)
 *  z        (K, N, d)     virtual-sensor observations
 *  r_tril   (K, N, d, d)  virtual-sensor scale matrices (R = T T^T; need not be triangular)
 *  fuse_w   (K, N, d)     crossmodal weights (fusion 1) or null
 *  mu       (K, N, d)     out: corrected sub-filter means
 *  Sigma    (K, N, d, d)  in: previous covariances; out: corrected covariances
 *  mu_f     (N, d), Sigma_f (N, d, d)  fused belief (fusion != 0), else may be null
 *  fusion   0 none, 1 crossmodal, 2 unimodal
 *  feedback 0: sub-filters keep their own beliefs (the reference's effective behaviour,
 *           SURVEY.md appendix C Q1); 1: fused belief is written back into every sub-filter
 */
int mmf_ekf_step(const float* A, const float* mu_pred, const float* q_tril, const float* z,
                 const float* r_tril, const float* fuse_w, float* mu, float* Sigma,
                 float* mu_f, float* Sigma_f, int N, int d, int K, int fusion, int feedback,
                 void* stream);

/* mmf_ekf_step whose write-back is gated by a DEVICE word: `feedback` applies only where *feedback_gate != 0
 * (null = always).  Carries the reference's batch-global blackout test -- DoorCrossmodalKalmanFilter.forward
 * takes the branch WITHOUT write-back as soon as any frame of the batch is blacked out
 * (/root/reference/crossmodal/door_models/crossmodal_kf.py:59-62) -- into the
 * native step loop without a host read per step. */
int mmf_ekf_step_gated(const float* A, const float* mu_pred, const float* q_tril, const float* z,
                       const float* r_tril, const float* fuse_w, float* mu, float* Sigma,
                       float* mu_f, float* Sigma_f, int N, int d, int K, int fusion, int feedback,
                       const int32_t* feedback_gate, void* stream);

/* K6: reverse mode of mmf_ekf_step without fusion (fusion = 0; the fusions of the K sub-filters
 * are a handful of element-wise torch ops on (K, N, d) and keep their autograd form).  Inputs as
 * the forward call's, Sigma_in = the covariances BEFORE the step; g_mu (K, N, d) / g_Sigma
 * (K, N, d, d): gradients of the corrected beliefs (either may be null = zero).  Outputs (any may
 * be null): g_A, g_mu_pred, g_z, g_r_tril, g_Sigma_in, shaped like their forward operands.
 * Used by the "hip" training backend for every EKF the reference trains end to end
 * (crossmodal/train_helpers.py:124-162; base_models/crossmodal_kf.py:88-151).
 */
int mmf_ekf_step_backward(const float* A, const float* mu_pred, const float* q_tril, const float* z,
                          const float* r_tril, const float* Sigma_in, const float* g_mu,
                          const float* g_Sigma, float* g_A, float* g_mu_pred, float* g_z,
                          float* g_r_tril, float* g_Sigma_in, int N, int d, int K, void* stream);

/* ---------------------------------------------------------------- unscented transform (UKF)
 * torchfilter's UnscentedKalmanFilter / VirtualSensorUnscentedKalmanFilter (absent, un-pinned
 * dependency of the reference, cf. crossmodal/door_models/kf.py:14-28 for the EKF sibling):
 * sigma points of N Gaussian beliefs, and the weighted moments of the propagated points.
 *  mu (N, d), Sigma (N, d, d); scale = sqrt(d + lambda);
 *  points (N, 2d+1, d): [mu, mu + scale L[:, i], mu - scale L[:, i]] with L = chol(Sigma)
 *  not_pd: int32 on the device or null, OR-ed with 1 when a covariance is not positive definite
 */
int mmf_ukf_sigma_points(const float* mu, const float* Sigma, float scale, float* points,
                         int32_t* not_pd, int N, int d, void* stream);
/*  points (N, 2d+1, d) propagated sigma points; wm0 / wc0: mean / covariance weight of point 0,
 *  wi: weight of every other point; q_tril (d, d): dynamics noise (Q = L L^T)
 *  -> mu_pred (N, d), Sigma_pred (N, d, d) = sum_i wc_i (X_i - mu)(X_i - mu)^T + Q
 */
int mmf_ukf_moments(const float* points, float wm0, float wc0, float wi, const float* q_tril,
                    float* mu_pred, float* Sigma_pred, int N, int d, void* stream);

/* R11: fusion of K virtual sensors before a single EKF (crossmodal_kf.py:291-359 mode 1;
 * unimodal_kf.py:56-115 mode 2, quirk Q5 preserved: see csrc/ekf.hip).
 *  z (K, N, d), tril (K, N, d, d), w (K, N, d) (mode 1) -> z_out (N, d), tril_out (N, d, d)
 */
int mmf_fuse_virtual_sensors(const float* z, const float* tril, const float* w, float* z_out,
                             float* tril_out, int N, int d, int K, int mode, void* stream);

/* All T steps of a fused EKF in one call (host loop; replaces torchfilter's Filter.forward_loop
 * over crossmodal_kf.py:88-151 / unimodal_kf.py:162-250): per step, mmf_dynamics_jacobian for
 * each of the K sub-filters at its current belief mean, then mmf_ekf_step.  Step-indexed
 * arrays are laid out (T, K, N, ...) so that step t's block is what mmf_ekf_step takes.
 */
typedef struct MmfEkfLoopArgs {
  int32_t T, N, d, K;
  int32_t fusion, feedback;  /* as mmf_ekf_step                                              */
  int32_t n_res_dyn;
  int32_t precision;         /* of dyn_packed (the Jacobian launches)                          */
  int32_t* range_flag;       /* f16x3: OR-ed with 1 when an operand leaves the f16 range, or null */
  const float* dyn_packed[MMF_LOOP_MAX_MEAS];  /* blobs of the dynamics networks (mmf_pack_particle_net) */
  const float* dyn_bias[MMF_LOOP_MAX_MEAS];    /* (T*N, 64) hoisted control terms               */
  const float* q_tril;       /* (K, d, d)                                                      */
  const float* z;            /* (T, K, N, d)      virtual-sensor observations                  */
  const float* r_tril;       /* (T, K, N, d, d)   virtual-sensor scale matrices                */
  const float* fuse_w;       /* (T, K, N, d) crossmodal weights (fusion 1) or null             */
  float* mu;                 /* (K, N, d)     sub-filter means, in/out                          */
  float* Sigma;              /* (K, N, d, d)  sub-filter covariances, in/out                    */
  float* mu_pred;            /* (K, N, d)     scratch                                           */
  float* A;                  /* (K, N, d, d)  scratch                                           */
  float* Sigma_f;            /* (N, d, d)     fused covariance of the last step (fusion != 0)   */
  float* estimates;          /* (T, N, d) out: fused mean, or sub-filter 0's mean for fusion 0  */
  const int32_t* feedback_gate; /* (T) device words or null: step t writes the fused belief back (feedback) only where
                                gate[t] != 0 -- the reference's batch-global blackout branch
                                (door_models/crossmodal_kf.py:59-62) decided on the device, no host read per step  */
  int32_t persistent;        /* ABI 40. != 0: the whole loop as ONE persistent launch (mmf_ekf_persistent_plan > 0);
                                mu_pred and A are not touched; same bits as the loop of launches                  */
  int32_t n_sync_words;      /* 4-byte words of sync_words (>= mmf_ekf_persistent_sync_words(N, K, d))            */
  uint32_t* sync_words;      /* persistent: device workspace of the in-launch hand-offs between the K sub-filters'
                                workgroups (tagged 8-byte granules), zeroed by the call; range_flag bit 2 = "a
                                hand-off timed out: discard this loop and run it as launches"                     */
} MmfEkfLoopArgs;            /* host struct holding device pointers                            */

int mmf_ekf_forward_loop(const MmfEkfLoopArgs* args /* host */, void* stream);

/* The persistent form of the EKF step loop (MmfEkfLoopArgs.persistent; csrc/ekf_persistent.inc): a wave owns 8
 * trajectories of ONE sub-filter for all T steps -- belief in registers, the dynamics network's weights in LDS, per step
 * the forward-mode Jacobian tile and the d x d algebra; K > 1 sub-filters meet once per step and trajectory through L2.
 * Replaces the 2 T launches of mmf_ekf_forward_loop (crossmodal_kf.py:88-151, unimodal_kf.py:162-250).
 *   mmf_ekf_persistent_plan: number of workgroups it would use (0: not eligible -- more than two rounds of tiles per wave,
 *   or a device with too few CUs to keep every workgroup resident).  d must be 2 or 3, n_res_dyn 3. */
int mmf_ekf_persistent_plan(int N, int K);
size_t mmf_ekf_persistent_sync_words(int N, int K, int d);

#ifdef __cplusplus
}
#endif
#endif /* MMF_H */
