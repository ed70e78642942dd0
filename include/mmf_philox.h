/*
 * mmf_philox.h -- counter-based process noise, defined ONCE for the HIP kernels and the CPU checker.
 *
 * torchfilter draws `MultivariateNormal(...).rsample()` inside every particle-filter step (external
 * dependency of the reference; call site /root/reference/crossmodal/eval_helpers.py:139-142)
 * from torch's global generator.  Here a standard-normal draw is a pure function of
 *     (seed, time step t, trajectory n, particle m)
 * -- Philox4x32-10 (Salmon et al., SC'11; the generator behind curand / torch.cuda) keyed by the seed, with
 * the counter (m, n, t, stream), followed by a Box-Muller transform built from the deterministic
 * functions of mmf_detmath.h and two fixed polynomials -- so the dynamics kernel generates its noise in
 * its epilogue (no (T, N, M, d) tensor in HBM, 12 B per particle-step less traffic), results do not depend
 * on how trajectories are sharded over GPUs, and oracle/strict reproduces every draw bit for bit.
 *
 * One call yields 4 normals (state_dim <= 4): u32 x[4] -> u1 = ((x0 >> 9) + 0.5) 2^-23, u2 = (x1 >> 8) 2^-24
 * -> r = sqrt(-2 log u1), (z0, z1) = r (cos, sin)(2 pi u2); likewise (z2, z3) from (x2, x3).
 * IEEE binary32 add / multiply / fma / sqrt / integer operations only, in a fixed order.
 */
#ifndef MMF_PHILOX_H
#define MMF_PHILOX_H

#include "mmf_detmath.h"

#define MMF_PHILOX_STREAM_NOISE 0u     /* per-particle process noise            */
#define MMF_PHILOX_STREAM_UNIFORM 1u   /* per-trajectory resampling uniforms    */

#if defined(__HIPCC__)
#define MMF_PHILOX_MULHI(a, b) __umulhi((a), (b))
#define MMF_PHILOX_SQRT(x) __builtin_sqrtf(x)
#else
#define MMF_PHILOX_MULHI(a, b) ((uint32_t)(((uint64_t)(a) * (uint64_t)(b)) >> 32))
#define MMF_PHILOX_SQRT(x) sqrtf(x)
#endif

MMF_DET_FN void mmf_philox4x32_10(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                  uint32_t out[4]) {
  for (int round = 0; round < 10; ++round) {
    const uint32_t hi0 = MMF_PHILOX_MULHI(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = MMF_PHILOX_MULHI(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* (cos, sin)(2 pi u) for u = j 2^-24, j in [0, 2^24): octant reduction on the integer, then Taylor
 * polynomials of sin / cos on |x| <= pi/4 with explicit fma (abs error < 1.5e-7) */
MMF_DET_FN void mmf_det_sincos2pi(uint32_t j, float* c, float* s) {
#if defined(__HIPCC__)
#pragma clang fp contract(off)
#endif
  const uint32_t oct = (j + (1u << 20)) >> 21;                       /* nearest multiple of 1/8 turn: 0..8 */
  const int32_t rem = (int32_t)j - (int32_t)(oct << 21);            /* |rem| <= 2^20  (1/16 turn)          */
  const float x = (float)rem * 3.7450704e-07f;                      /* 2 pi 2^-24, |x| <= pi/8 .. pi/4     */
  const float x2 = x * x;
  float ps = -1.9841270e-04f;                                       /* sin x = x + x^3 (-1/6 + x^2 (1/120 - x^2/5040)) */
  ps = MMF_DET_FMAF(ps, x2, 8.3333333e-03f);
  ps = MMF_DET_FMAF(ps, x2, -1.6666667e-01f);
  const float sx = MMF_DET_FMAF(ps * x2, x, x);
  float pc = 2.4801587e-05f;                                        /* cos x = 1 + x^2 (-1/2 + x^2 (1/24 + x^2 (-1/720 + x^2/40320))) */
  pc = MMF_DET_FMAF(pc, x2, -1.3888889e-03f);
  pc = MMF_DET_FMAF(pc, x2, 4.1666667e-02f);
  pc = MMF_DET_FMAF(pc, x2, -0.5f);
  const float cx = MMF_DET_FMAF(pc, x2, 1.0f);
  /* rotate by oct * 45 degrees: (cos, sin)(a + b) with (cos b, sin b) from an exact table of 8 directions */
  const float h = 0.70710678118654752440f;
  const uint32_t o = oct & 7u;
  const float cb = (o == 0) ? 1.f : (o == 1) ? h : (o == 2) ? 0.f : (o == 3) ? -h : (o == 4) ? -1.f : (o == 5) ? -h : (o == 6) ? 0.f : h;
  const float sb = (o == 0) ? 0.f : (o == 1) ? h : (o == 2) ? 1.f : (o == 3) ? h : (o == 4) ? 0.f : (o == 5) ? -h : (o == 6) ? -1.f : -h;
  const float t0 = cx * cb, t1 = sx * cb;
  *c = MMF_DET_FMAF(-sx, sb, t0);
  *s = MMF_DET_FMAF(cx, sb, t1);
}

/* 4 standard normals of (seed, stream | t, n, m): key = seed, counter = (m, n, t, stream) */
MMF_DET_FN void mmf_philox_normal4(uint64_t seed, uint32_t t, uint32_t n, uint32_t m, float z[4]) {
#if defined(__HIPCC__)
#pragma clang fp contract(off)
#endif
  uint32_t x[4];
  mmf_philox4x32_10((uint32_t)seed, (uint32_t)(seed >> 32), m, n, t, MMF_PHILOX_STREAM_NOISE, x);
  for (int p = 0; p < 2; ++p) {
    const float u1 = ((float)(x[2 * p] >> 9) + 0.5f) * 1.1920929e-07f;   /* (k + 1/2) 2^-23 in (0, 1): exact in fp32 */
    const float r = MMF_PHILOX_SQRT(-2.0f * mmf_det_log(u1));
    float c, s;
    mmf_det_sincos2pi(x[2 * p + 1] >> 8, &c, &s);
    z[2 * p] = r * c;
    z[2 * p + 1] = r * s;
  }
}

/* uniform in [0, 1) with 24 bits for trajectory n at step t (systematic resampling) */
MMF_DET_FN float mmf_philox_uniform(uint64_t seed, uint32_t t, uint32_t n) {
  uint32_t x[4];
  mmf_philox4x32_10((uint32_t)seed, (uint32_t)(seed >> 32), 0u, n, t, MMF_PHILOX_STREAM_UNIFORM, x);
  return (float)(x[0] >> 8) * 5.9604645e-08f;
}

#endif /* MMF_PHILOX_H */
