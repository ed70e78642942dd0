/*
 * mmf_detmath.h -- the transcendental functions of the bit-reproducible ("strict", MMF_PREC_F32)
 * arithmetic mode, defined ONCE for both sides of the parity certificate.
 *
 * libm's / the GPU's expf, logf differ by an ulp between implementations, so in the exact-fp32 mode
 * every transcendental on the filter path (the sigmoid gate of the dynamics model,
 * /root/reference/crossmodal/door_models/dynamics.py:60-66; the logsumexp over modalities,
 * /root/reference/crossmodal/base_models/crossmodal_pf.py:136-141; torchfilter's weight
 * normalisation) is one of the functions below: IEEE-754 binary32 operations only -- add, multiply,
 * fused multiply-add, divide, round-to-nearest-even to integer, integer bit manipulation -- in a
 * fixed order.  The same text compiles into the HIP kernels (csrc/) and into the CPU checker
 * (oracle/strict/mmf_strict.c), and any IEEE-conforming compiler gives the same bits, PROVIDED the
 * translation unit does not contract a*b+c on its own (hipcc: the pragma below; gcc:
 * -ffp-contract=off) -- every fused operation here is an explicit fmaf.
 *
 * Accuracy (against fp64, tests/test_strict_cpu.py): exp 1.5e-7 (1 + |x|) relative on [-87, 0] (the
 * argument x log2(e) is rounded once), log 2.5e-7 absolute on [1, 8] and relative elsewhere, sigmoid
 * 3e-7 absolute, logaddexp 1.5e-6 absolute.
 */
#ifndef MMF_DETMATH_H
#define MMF_DETMATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define MMF_DET_FN __device__ __forceinline__
#define MMF_DET_FMAF(a, b, c) __builtin_fmaf((a), (b), (c))
#define MMF_DET_RINT(x) __builtin_rintf(x)
#define MMF_DET_FMAX(a, b) __builtin_fmaxf((a), (b))
#else
#include <math.h>
#include <string.h>
#define MMF_DET_FN static inline
#define MMF_DET_FMAF(a, b, c) fmaf((a), (b), (c))
#define MMF_DET_RINT(x) rintf(x)
#define MMF_DET_FMAX(a, b) fmaxf((a), (b))
#endif

MMF_DET_FN float mmf_det_from_bits(int32_t i) {
#if defined(__HIPCC__)
  return __builtin_bit_cast(float, i);
#else
  float f;
  memcpy(&f, &i, 4);
  return f;
#endif
}

MMF_DET_FN int32_t mmf_det_to_bits(float f) {
#if defined(__HIPCC__)
  return __builtin_bit_cast(int32_t, f);
#else
  int32_t i;
  memcpy(&i, &f, 4);
  return i;
#endif
}

/* exp(x) for x <= 0 (also -inf -> 2^-126 * p(0), NaN -> that too): degree-6 Taylor polynomial of
 * 2^f on |f| <= 1/2, separate (UN-fused) multiplies and adds in Horner order -- the resampler's
 * weight function, identical to oracle/resample.py::detexp (numpy has no fma). */
MMF_DET_FN float mmf_det_exp_nonpos(float x) {
#if defined(__HIPCC__)
#pragma clang fp contract(off)
#endif
  float t = x * 1.4426950408889634f;
  t = MMF_DET_FMAX(t, -126.0f);
  const float n = MMF_DET_RINT(t); /* round half to even (v_rndne_f32) */
  const float f = t - n;
  float p = 0.00015403530393381608f;
  p = p * f;
  p = p + 0.0013333558146428443f;
  p = p * f;
  p = p + 0.009618129107628477f;
  p = p * f;
  p = p + 0.05550410866482158f;
  p = p * f;
  p = p + 0.2402265069591007f;
  p = p * f;
  p = p + 0.6931471805599453f;
  p = p * f;
  p = p + 1.0f;
  const float scale = mmf_det_from_bits(((int32_t)n + 127) << 23);
  return p * scale;
}

/* log(x) for finite x >= 2^-126 (callers pass sums of exponentials in [1, K]): x = m 2^e with
 * m in [sqrt(1/2), sqrt(2)), log m by the degree-9 polynomial in r = m - 1 of the Cephes logf
 * (every product feeding a sum is an explicit fmaf), e ln 2 added in two parts. */
MMF_DET_FN float mmf_det_log(float x) {
#if defined(__HIPCC__)
#pragma clang fp contract(off)
#endif
  int32_t ix = mmf_det_to_bits(x);
  int32_t e = ((ix >> 23) & 0xff) - 126;                   /* x = m 2^e, m in [1/2, 1) */
  float m = mmf_det_from_bits((ix & 0x007fffff) | 0x3f000000);
  if (m < 0.70710678118654752440f) {
    e -= 1;
    m = m + m;
  }
  const float r = m - 1.0f;                                /* exact (Sterbenz) */
  const float z = r * r;
  float p = 7.0376836292e-2f;
  p = MMF_DET_FMAF(p, r, -1.1514610310e-1f);
  p = MMF_DET_FMAF(p, r, 1.1676998740e-1f);
  p = MMF_DET_FMAF(p, r, -1.2420140846e-1f);
  p = MMF_DET_FMAF(p, r, 1.4249322787e-1f);
  p = MMF_DET_FMAF(p, r, -1.6668057665e-1f);
  p = MMF_DET_FMAF(p, r, 2.0000714765e-1f);
  p = MMF_DET_FMAF(p, r, -2.4999993993e-1f);
  p = MMF_DET_FMAF(p, r, 3.3333331174e-1f);
  const float fe = (float)e;
  float y = (p * r) * z;
  y = MMF_DET_FMAF(fe, -2.12194440e-4f, y);
  y = MMF_DET_FMAF(z, -0.5f, y);
  const float s = r + y;
  return MMF_DET_FMAF(fe, 0.693359375f, s);
}

/* 1 / (1 + exp(-g)) without evaluating exp of a positive argument */
MMF_DET_FN float mmf_det_sigmoid(float g) {
#if defined(__HIPCC__)
#pragma clang fp contract(off)
#endif
  const float a = g < 0.f ? g : -g;                        /* -|g| */
  const float zexp = mmf_det_exp_nonpos(a);
  const float den = 1.0f + zexp;
  return g < 0.f ? zexp / den : 1.0f / den;
}

/* log(exp(a) + exp(b)); -inf when both are -inf */
MMF_DET_FN float mmf_det_logaddexp(float a, float b) {
#if defined(__HIPCC__)
#pragma clang fp contract(off)
#endif
  const float m = MMF_DET_FMAX(a, b);
  if (!(m > -3.0e38f)) return m;                           /* -inf (or NaN) */
  const float s = mmf_det_exp_nonpos(a - m) + mmf_det_exp_nonpos(b - m);
  return m + mmf_det_log(s);
}

#endif /* MMF_DETMATH_H */
