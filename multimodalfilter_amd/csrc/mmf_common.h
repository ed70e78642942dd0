// Shared device helpers for the gfx950 filter kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mmf.h"

#define MMF_WAVE 64

#define MMF_CHECK_LAUNCH()                          \
  do {                                              \
    hipError_t _e = hipGetLastError();              \
    if (_e != hipSuccess) return static_cast<int>(_e); \
  } while (0)

namespace mmf {

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__device__ __forceinline__ unsigned long long shfl_up_u64(unsigned long long v, int o) {
  unsigned lo = __shfl_up(static_cast<unsigned>(v), o);
  unsigned hi = __shfl_up(static_cast<unsigned>(v >> 32), o);
  return (static_cast<unsigned long long>(hi) << 32) | lo;
}

// Inclusive scan across the 64 lanes of a wave.
__device__ __forceinline__ unsigned long long wave_inclusive_scan(unsigned long long v, int lane) {
#pragma unroll
  for (int o = 1; o < MMF_WAVE; o <<= 1) {
    unsigned long long t = shfl_up_u64(v, o);
    if (lane >= o) v += t;
  }
  return v;
}

// Deterministic fp32 exp for x <= 0: the exact operation sequence of
// oracle/resample.py::detexp (separate, un-fused multiplies and adds, Horner order), so the
// fixed-point weights -- and therefore the resampled indices -- match the oracle bit for bit.
__device__ __forceinline__ float detexp(float x) {
#pragma clang fp contract(off)
  const float LOG2E = 1.4426950408889634f;
  float t = x * LOG2E;
  t = fmaxf(t, -126.0f);  // fmaxf(NaN, c) = c; -inf -> -126
  float n = rintf(t);     // v_rndne_f32: round half to even
  float f = t - n;
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
  float scale = __int_as_float((static_cast<int>(n) + 127) << 23);
  return p * scale;
}

}  // namespace mmf
