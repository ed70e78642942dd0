// Micro-benchmark: do f16 MFMAs of one wave overlap with the split-style VALU work of the
// other wave on the same SIMD?  512 threads = 8 waves = 2 per SIMD; waves 0-3 run `a`, 4-7 run `b`.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void mfma_work(f32x16& acc, half8 a, half8 b, int n) {
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  }
}
__device__ __forceinline__ void valu_work(float (&x)[16], int n) {
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      float x0 = x[2 * p], x1 = x[2 * p + 1];
      auto h = __builtin_amdgcn_cvt_pkrtz(x0, x1);
      f32x2 xs = {x0, x1}, hf = {(float)h[0], (float)h[1]};
      f32x2 r = xs - hf;
      auto l = __builtin_amdgcn_cvt_pkrtz(r[0], r[1]);
      x[2 * p] = __int_as_float(max(__float_as_int(r[0] + (float)l[0]), 0)) + 1.0f;
      x[2 * p + 1] = __int_as_float(max(__float_as_int(r[1] + (float)l[1]), 0)) + 1.0f;
    }
  }
}
// mode: 0 = all waves MFMA, 1 = all waves VALU, 2 = waves 0-3 MFMA + waves 4-7 VALU, 3 = only waves 0-3 MFMA (4-7 idle), 4 = only 4-7 VALU
__global__ __launch_bounds__(512) void k(float* out, int mode, int n) {
  const int wave = threadIdx.x >> 6;
  f32x16 acc = {};
  half8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(1.0f + i * 0.5f); }
  float x[16];
  for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 0.37f + i;
  const bool lower = wave < 4;
  bool do_m = mode == 0 || ((mode == 2 || mode == 3) && lower);
  bool do_v = mode == 1 || ((mode == 2 || mode == 4) && !lower);
  if (do_m) mfma_work(acc, a, b, n);
  if (do_v) valu_work(x, n);
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i] + x[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int n = 2000;
  for (int mode = 0; mode < 5; ++mode) {
    k<<<256, 512>>>(out, mode, n);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<<<256, 512>>>(out, mode, n);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("mode %d: %.3f ms  (per iter %.1f ns)\n", mode, ms, ms * 1e6 / n);
  }
  return 0;
}
