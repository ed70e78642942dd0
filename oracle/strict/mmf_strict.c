/*
 * mmf_strict.c -- CPU restatement of the ARITHMETIC of the exact-fp32 ("strict", MMF_PREC_F32) mode
 * of the HIP filter path, operation for operation.  TEST INFRASTRUCTURE (see oracle/__init__.py):
 * only tests/, __graft_entry__.smoke() and bench.py's checker legs may load it.
 *
 * Why it exists: `north_star` asks for bit-exact resample indices under a fixed seed on the whole
 * filter (the caller /root/reference/crossmodal/eval_helpers.py:125-160).  K1's resampler is integer
 * and already order-independent, but its input -- the log-likelihoods -- comes out of ~100 k fp32
 * multiply-adds per particle, and a different summation order moves the last ulp.  In the strict mode
 * every dense layer of the engine is a k-ordered chain of fused multiply-adds (v_mfma_f32_32x32x2_f32 /
 * 16x16x4_f32 are bit-for-bit `fmaf` chains, MI355X_MICROARCH.md "Matrix cores"; K7 issues explicit
 * `fma`), every transcendental is one of include/mmf_detmath.h, every reduction a fixed tree.  This
 * file walks the same chains in the same order with C's fmaf, so engine and checker agree in EVERY bit
 * and the free-running filters draw identical ancestors for any number of steps.
 *
 * What each function follows (reference algorithm | kernel whose operation order it restates):
 *   strict_linear        nn.Linear (+ReLU / +skip) of the N-row networks, door_models/layers.py:11-40,66-95,
 *                        crossmodal_pf.py:74-106                     | csrc/traj_program.hip LINEAR
 *   strict_conv          nn.Conv2d of the image encoder, door_models/layers.py:43-58 | image_encoder.hip conv_kernel
 *   strict_fc_tail       Flatten + Linear 8192->64 + ReLU + ResLinear, layers.py:59-63 | fc_partial_kernel + fc_tail_kernel
 *   strict_particle_net  per-particle dynamics / measurement MLP, door_models/dynamics.py:102-134,
 *                        door_models/pf.py:63-107                     | particle_net.hip (PREC F32)
 *   strict_measure_epilogue / strict_dynamics_epilogue  base_models/crossmodal_pf.py:106-139, dynamics.py:60-66
 *   strict_estimate      torchfilter's weighted-mean estimate  | pf_resample.hip pass 2
 * The torch oracle (oracle/models.py, pinned to the reference by tests/golden) and this file agree to
 * ~1e-6 (tests/test_strict_cpu.py); the HIP engine and this file agree exactly (tests/test_gpu_strict.py).
 *
 * Build: gcc -O3 -mavx2 -mfma -ffp-contract=off -fopenmp -shared -fPIC (oracle/strict/__init__.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/mmf_detmath.h"
#include "../../include/mmf_philox.h"

#define UNITS 64
#define PB 16 /* particles evaluated together (the inner, vectorised loop) */

/* K2's ReLU: max((int)bits, 0) -- negative floats and -0.0 become +0.0 */
static inline float relu_bits(float v) {
  int32_t i = mmf_det_to_bits(v);
  return mmf_det_from_bits(i < 0 ? 0 : i);
}

/* ------------------------------------------------------------------ exported scalar functions */
void strict_det_exp_nonpos(const float* x, float* y, long n) {
  for (long i = 0; i < n; ++i) y[i] = mmf_det_exp_nonpos(x[i]);
}
void strict_det_log(const float* x, float* y, long n) {
  for (long i = 0; i < n; ++i) y[i] = mmf_det_log(x[i]);
}
void strict_det_sigmoid(const float* x, float* y, long n) {
  for (long i = 0; i < n; ++i) y[i] = mmf_det_sigmoid(x[i]);
}
void strict_det_logaddexp(const float* a, const float* b, float* y, long n) {
  for (long i = 0; i < n; ++i) y[i] = mmf_det_logaddexp(a[i], b[i]);
}

/* ------------------------------------------------------------------ K7 LINEAR
 * y[r][o] = act(chain + res[r][o]),  chain = b[o] (or 0), then for k = 0 .. in_dim-1 in order:
 * chain = fma(W[o][c0 + k], x[r][k], chain).   act: 0 none, 1 ReLU (fmaxf), 2 sigmoid, 3 sqrt(v*v + fparam) */
void strict_linear(const float* x, long R, int in_dim, const float* W, int w_stride, int c0, const float* b,
                   const float* res, int out_dim, int act, float fparam, float* y) {
#pragma omp parallel for schedule(static)
  for (long r = 0; r < R; ++r) {
    const float* xr = x + r * in_dim;
    for (int o = 0; o < out_dim; ++o) {
      float acc = b ? b[o] : 0.f;
      const float* w = W + (long)o * w_stride + c0;
      for (int k = 0; k < in_dim; ++k) acc = fmaf(w[k], xr[k], acc);
      float v = acc + (res ? res[r * out_dim + o] : 0.f);
      if (act == 1) v = fmaxf(v, 0.f);
      else if (act == 2) v = mmf_det_sigmoid(v);
      else if (act == 3) v = sqrtf(fmaf(v, v, fparam));
      y[r * out_dim + o] = v;
    }
  }
}

/* ------------------------------------------------------------------ K4 convolution (f32 path)
 * out[n][co][y][x] = bias[co], then for tap = (ky, kx) row-major, for ci = 0 .. cin-1:
 *   fma(w[co][ci][ky][kx], in[n][ci][y + ky - h][x + kx - h] (0 outside), acc)
 * (conv_kernel: k-step s = tap * cin/4 + cg, lane quarter q -> ci = 4 cg + q; cin == 1: s = tap / 4,
 * q = tap % 4 -- in both cases the chain runs tap-major, then channel), then + skip, then ReLU. */
void strict_conv(const float* in, const float* w, const float* b, const float* skip, long N, int cin, int cout,
                 int ks, int relu, float* out) {
  const int h = ks / 2, P = 32 + 2 * h;
#pragma omp parallel
  {
    float* pad = (float*)malloc(sizeof(float) * (size_t)cin * P * P);
#pragma omp for schedule(dynamic, 1)
    for (long n = 0; n < N; ++n) {
      memset(pad, 0, sizeof(float) * (size_t)cin * P * P);
      for (int ci = 0; ci < cin; ++ci)
        for (int y = 0; y < 32; ++y)
          memcpy(pad + ((size_t)ci * P + y + h) * P + h, in + (((size_t)n * cin + ci) * 32 + y) * 32, 32 * sizeof(float));
      for (int co = 0; co < cout; ++co)
        for (int y = 0; y < 32; ++y) {
          float acc[32];
          for (int x = 0; x < 32; ++x) acc[x] = b[co];
          for (int ky = 0; ky < ks; ++ky)
            for (int kx = 0; kx < ks; ++kx)
              for (int ci = 0; ci < cin; ++ci) {
                const float wv = w[(((size_t)co * cin + ci) * ks + ky) * ks + kx];
                const float* row = pad + ((size_t)ci * P + y + ky) * P + kx;
#pragma omp simd
                for (int x = 0; x < 32; ++x) acc[x] = fmaf(wv, row[x], acc[x]);
              }
          float* o = out + (((size_t)n * cout + co) * 32 + y) * 32;
          const float* s = skip ? skip + (((size_t)n * cout + co) * 32 + y) * 32 : NULL;
          for (int x = 0; x < 32; ++x) {
            float v = acc[x];
            if (s) v = v + s[x];
            if (relu) v = fmaxf(v, 0.f);
            o[x] = v;
          }
        }
    }
    free(pad);
  }
}

/* ------------------------------------------------------------------ K4 linear tail (f32 path)
 * fc_partial_kernel: 16 K-slices of 512; within a slice two accumulators: for k0 = 0, 16, ..: acc0 takes
 * k0 + 4q + 0 (q = 0..3), acc1 k0 + 4q + 1, acc0 k0 + 4q + 2, acc1 k0 + 4q + 3; partial = acc0 + acc1.
 * fc_tail_kernel: h = relu(bias + partial[0] + .. + partial[15]); t = relu(chain(b1, W1, h));
 * feat = relu(chain(b2 + h, W2, t)), chains in natural k order. */
void strict_fc_tail(const float* act, const float* fcw, const float* fcb, const float* w1, const float* b1,
                    const float* w2, const float* b2, long N, float* feat) {
#pragma omp parallel for schedule(static)
  for (long n = 0; n < N; ++n) {
    const float* a = act + (size_t)n * 8192;
    float hvec[UNITS], tvec[UNITS];
    for (int o = 0; o < UNITS; ++o) {
      const float* w = fcw + (size_t)o * 8192;
      float hsum = fcb[o];
      for (int s = 0; s < 16; ++s) {
        float acc0 = 0.f, acc1 = 0.f;
        for (int k0 = 0; k0 < 512; k0 += 16) {
          const int kb = s * 512 + k0;
          for (int q = 0; q < 4; ++q) acc0 = fmaf(w[kb + 4 * q + 0], a[kb + 4 * q + 0], acc0);
          for (int q = 0; q < 4; ++q) acc1 = fmaf(w[kb + 4 * q + 1], a[kb + 4 * q + 1], acc1);
          for (int q = 0; q < 4; ++q) acc0 = fmaf(w[kb + 4 * q + 2], a[kb + 4 * q + 2], acc0);
          for (int q = 0; q < 4; ++q) acc1 = fmaf(w[kb + 4 * q + 3], a[kb + 4 * q + 3], acc1);
        }
        hsum = hsum + (acc0 + acc1);
      }
      hvec[o] = fmaxf(hsum, 0.f);
    }
    for (int o = 0; o < UNITS; ++o) {
      float t = b1[o];
      for (int k = 0; k < UNITS; ++k) t = fmaf(w1[o * UNITS + k], hvec[k], t);
      tvec[o] = fmaxf(t, 0.f);
    }
    for (int o = 0; o < UNITS; ++o) {
      float y = b2[o] + hvec[o];
      for (int k = 0; k < UNITS; ++k) y = fmaf(w2[o * UNITS + k], tvec[k], y);
      feat[n * UNITS + o] = fmaxf(y, 0.f);
    }
  }
}

/* ------------------------------------------------------------------ K2 per-particle network (f32 path)
 * particle_net.hip: k-step s (0..31) of a 64x64 layer feeds features kmap(s, 0) then kmap(s, 1),
 *   kmap(s, h) = 32 (s >> 4) + ((s & 15) & 3) + 8 ((s & 15) >> 2) + 4 h,
 * each as one fma onto the accumulator, which starts at the bias (first layer of a block), at
 * skip + bias (second layer) or at the per-trajectory hoisted term (join layer).  First layer: [x; 1]
 * against [w_in | b_in] in natural order from 0.  Head: two 32-feature chains from 0 (rows 32t + 8g + 4h + e
 * for h = 0 and h = 1), added. */
typedef struct StrictNet {
  int32_t d_in, n_res, relu_after_join, n_out, join_in, join_state_off;
  const float* w_in;       /* (64, d_in) */
  const float* b_in;       /* (64) */
  const float* w_enc[2];   /* (64, 64) */
  const float* b_enc[2];
  const float* w_join;     /* (64, join_in) */
  const float* w_res[6];
  const float* b_res[6];
  const float* w_head;     /* (n_out, 64) */
} StrictNet;

static int kmap(int s, int h) { return 32 * (s >> 4) + ((s & 15) & 3) + 8 * ((s & 15) >> 2) + 4 * h; }

/* out[f][p] = init[f][p] then the 64-term chain; W row stride `ws`, column offset `c0` */
static void layer64(const float* W, int ws, int c0, const int* order, float (*in)[PB], float (*acc)[PB]) {
  for (int f = 0; f < UNITS; ++f) {
    const float* w = W + (size_t)f * ws + c0;
    float a[PB];
    for (int p = 0; p < PB; ++p) a[p] = acc[f][p];
    for (int i = 0; i < UNITS; ++i) {
      const int k = order[i];
      const float wv = w[k];
#pragma omp simd
      for (int p = 0; p < PB; ++p) a[p] = fmaf(wv, in[k][p], a[p]);
    }
    for (int p = 0; p < PB; ++p) acc[f][p] = a[p];
  }
}

/* raw head outputs (before the head bias): out (R, n_out) */
void strict_particle_net(const StrictNet* net, const float* states, const float* traj_bias, long R, int M,
                         float* out) {
  int order[UNITS];
  for (int s = 0; s < 32; ++s) { order[2 * s] = kmap(s, 0); order[2 * s + 1] = kmap(s, 1); }
  const int D = net->d_in;
  const int ks0 = (D + 2) / 2;  /* first layer: 2 * ks0 columns [x_0 .. x_{D-1}, 1, 0..] */
#pragma omp parallel for schedule(static)
  for (long blk = 0; blk < (R + PB - 1) / PB; ++blk) {
    float X[UNITS][PB], H[UNITS][PB];
    long rows[PB];
    for (int p = 0; p < PB; ++p) { long r = blk * PB + p; rows[p] = r < R ? r : R - 1; }
    /* encoder layer 0 */
    for (int f = 0; f < UNITS; ++f)
      for (int p = 0; p < PB; ++p) {
        float acc = 0.f;
        for (int c = 0; c < 2 * ks0; ++c) {
          const float w = c < D ? net->w_in[f * D + c] : (c == D ? net->b_in[f] : 0.f);
          const float x = c < D ? states[rows[p] * D + c] : (c == D ? 1.f : 0.f);
          acc = fmaf(w, x, acc);
        }
        X[f][p] = relu_bits(acc);
      }
    /* encoder residual block */
    for (int f = 0; f < UNITS; ++f) for (int p = 0; p < PB; ++p) H[f][p] = net->b_enc[0][f];
    layer64(net->w_enc[0], UNITS, 0, order, X, H);
    for (int f = 0; f < UNITS; ++f) for (int p = 0; p < PB; ++p) H[f][p] = relu_bits(H[f][p]);
    for (int f = 0; f < UNITS; ++f) for (int p = 0; p < PB; ++p) X[f][p] = X[f][p] + net->b_enc[1][f];
    layer64(net->w_enc[1], UNITS, 0, order, H, X);
    for (int f = 0; f < UNITS; ++f) for (int p = 0; p < PB; ++p) X[f][p] = relu_bits(X[f][p]);
    /* join layer: accumulator starts at the per-trajectory hoisted term */
    for (int f = 0; f < UNITS; ++f)
      for (int p = 0; p < PB; ++p) H[f][p] = traj_bias[(rows[p] / M) * UNITS + f];
    layer64(net->w_join, net->join_in, net->join_state_off, order, X, H);
    if (net->relu_after_join)
      for (int f = 0; f < UNITS; ++f) for (int p = 0; p < PB; ++p) H[f][p] = relu_bits(H[f][p]);
    /* trunk: activations live in H, X is scratch */
    for (int i = 0; i < net->n_res; ++i) {
      for (int f = 0; f < UNITS; ++f) for (int p = 0; p < PB; ++p) X[f][p] = net->b_res[2 * i][f];
      layer64(net->w_res[2 * i], UNITS, 0, order, H, X);
      for (int f = 0; f < UNITS; ++f) for (int p = 0; p < PB; ++p) X[f][p] = relu_bits(X[f][p]);
      for (int f = 0; f < UNITS; ++f) for (int p = 0; p < PB; ++p) H[f][p] = H[f][p] + net->b_res[2 * i + 1][f];
      layer64(net->w_res[2 * i + 1], UNITS, 0, order, X, H);
      for (int f = 0; f < UNITS; ++f) for (int p = 0; p < PB; ++p) H[f][p] = relu_bits(H[f][p]);
    }
    /* head */
    for (int o = 0; o < net->n_out; ++o)
      for (int p = 0; p < PB; ++p) {
        float part[2] = {0.f, 0.f};
        for (int h = 0; h < 2; ++h)
          for (int t = 0; t < 2; ++t)
            for (int g = 0; g < 4; ++g)
              for (int e = 0; e < 4; ++e) {
                const int row = 32 * t + 8 * g + 4 * h + e;
                part[h] = fmaf(net->w_head[o * UNITS + row], H[row][p], part[h]);
              }
        const long r = blk * PB + p;
        if (r < R) out[r * net->n_out + o] = part[0] + part[1];
      }
  }
}

/* measurement epilogue: ll = raw + b_head (+ modality log-weight); combine: logaddexp with the running value */
void strict_measure_epilogue(const float* raw, float b_head, const float* mod_logw, int logw_stride, long R, int M,
                             int combine, float* loglik) {
#pragma omp parallel for schedule(static)
  for (long r = 0; r < R; ++r) {
    float ll = raw[r] + b_head;
    if (mod_logw) ll = ll + mod_logw[(r / M) * logw_stride];
    if (combine) ll = mmf_det_logaddexp(loglik[r], ll);
    loglik[r] = ll;
  }
}

/* dynamics epilogue: x' = x + (dir + b) * sigmoid(gate + b) + L eps  (each term one fma, k ascending) */
void strict_dynamics_epilogue(const float* raw, const float* b_head, const float* states, const float* noise,
                              const float* tril, long R, int D, float* out) {
#pragma omp parallel for schedule(static)
  for (long r = 0; r < R; ++r) {
    const float gate = raw[r * (D + 1) + D] + b_head[D];
    const float sg = mmf_det_sigmoid(gate);
    for (int i = 0; i < D; ++i) {
      float v = fmaf(raw[r * (D + 1) + i] + b_head[i], sg, states[r * D + i]);
      if (noise)
        for (int k = 0; k < D; ++k) v = fmaf(tril[i * D + k], noise[r * D + k], v);
      out[r * D + i] = v;
    }
  }
}

/* ------------------------------------------------------------------ K1 weighted-mean estimate
 * pf_resample.hip pass 2: thread `tid` of a `block`-thread workgroup owns particles base + 4 tid + j
 * (j = 0..3) of every chunk base = 0, 4 block, ..; it accumulates S += e and acc_c = fma(e, x_c, acc_c) in
 * that order; a wave's 64 partials are summed by the DPP tree of mmf::wave_sum (= xor butterfly with offsets 1, 2, .., 32), the waves'
 * totals sequentially; estimate_c = acc_c / S.   e (N, M) = the fp32 weights detexp(x - max). */
void strict_estimate(const float* e, const float* states, long N, int M, int D, int block, float* estimate) {
#pragma omp parallel for schedule(static)
  for (long n = 0; n < N; ++n) {
    const float* en = e + n * M;
    const float* xn = states + n * (long)M * D;
    float* part = (float*)malloc(sizeof(float) * (size_t)block * (D + 1));
    for (int tid = 0; tid < block; ++tid) {
      float S = 0.f, acc[4] = {0.f, 0.f, 0.f, 0.f};
      for (int base = 0; base < M; base += 4 * block)
        for (int j = 0; j < 4; ++j) {
          const int i = base + 4 * tid + j;
          if (i < M) {
            S = S + en[i];
            for (int c = 0; c < D; ++c) acc[c] = fmaf(en[i], xn[(long)i * D + c], acc[c]);
          }
        }
      part[tid * (D + 1)] = S;
      for (int c = 0; c < D; ++c) part[tid * (D + 1) + 1 + c] = acc[c];
    }
    float tot[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int w = 0; w < block / 64; ++w)
      for (int c = 0; c <= D; ++c) {
        float v[64];
        for (int l = 0; l < 64; ++l) v[l] = part[(w * 64 + l) * (D + 1) + c];
        for (int o = 1; o < 64; o <<= 1) {  /* mmf::wave_sum: DPP tree = xor butterfly 1, 2, 4, .., 32, read at lane 63 */
          float t[64];
          for (int l = 0; l < 64; ++l) t[l] = v[l] + v[l ^ o];
          memcpy(v, t, sizeof(v));
        }
        tot[c] = tot[c] + v[63];
      }
    for (int c = 0; c < D; ++c) estimate[n * D + c] = tot[1 + c] / tot[0];
    free(part);
  }
}

/* ------------------------------------------------------------------ counter-based noise (include/mmf_philox.h)
 * the draws the dynamics kernel generates in its epilogue (mmf_pf_dynamics_philox), materialised */
void strict_philox_normals(uint64_t seed, uint32_t step, uint32_t traj0, long N, int M, int d, float* out) {
#pragma omp parallel for schedule(static)
  for (long r = 0; r < N * M; ++r) {
    float z[4];
    mmf_philox_normal4(seed, step, traj0 + (uint32_t)(r / M), (uint32_t)(r % M), z);
    for (int i = 0; i < d; ++i) out[r * d + i] = z[i];
  }
}

void strict_philox_uniforms(uint64_t seed, uint32_t step0, uint32_t traj0, int T, int N, float* out) {
  for (int i = 0; i < T * N; ++i) out[i] = mmf_philox_uniform(seed, step0 + (uint32_t)(i / N), traj0 + (uint32_t)(i % N));
}

void strict_philox_raw(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t* out) {
  mmf_philox4x32_10(k0, k1, c0, c1, c2, c3, out);
}
