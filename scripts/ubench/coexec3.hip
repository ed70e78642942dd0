// Intra-wave overlap: one stream "1 MFMA + K independent VALU" -- does the VALU hide in the MFMA shadow?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KV, int THREADS>
__global__ __launch_bounds__(THREADS) void k(float* out, int n) {
  f32x16 acc = {};
  half8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(1.0f + i * 0.5f); }
  float x[16];
  for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 0.37f + i;
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
#pragma unroll
      for (int v = 0; v < KV; ++v) {  // split-style ops on independent registers
        const int p = (m * KV + v) % 8;
        float x0 = x[2 * p], x1 = x[2 * p + 1];
        if (v % 3 == 0) { auto h = __builtin_amdgcn_cvt_pkrtz(x0, x1); x[2 * p] = __builtin_bit_cast(float, h) + x1; }
        else if (v % 3 == 1) { x[2 * p + 1] = x1 - x0; }
        else { x[2 * p] = fmaf(x0, 1.0001f, x1); }
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, KV, 0);
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i] + x[i];
  out[blockIdx.x * THREADS + threadIdx.x] = s;
}
template <int KV, int THREADS> void run(float* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int n = 2000; float t;
  k<KV, THREADS><<<256, THREADS>>>(out, n); hipDeviceSynchronize();
  hipEventRecord(e0); k<KV, THREADS><<<256, THREADS>>>(out, n); hipEventRecord(e1); hipEventSynchronize(e1);
  hipEventElapsedTime(&t, e0, e1);
  printf("waves/SIMD %d  VALU per MFMA %d : %.3f ms  (%.1f ns per MFMA per wave)\n", THREADS / 256, KV, t, t * 1e6 / (n * 8));
}
int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  run<0, 256>(out); run<2, 256>(out); run<4, 256>(out); run<6, 256>(out); run<8, 256>(out); run<12, 256>(out);
  run<0, 512>(out); run<4, 512>(out); run<6, 512>(out); run<8, 512>(out);
  return 0;
}
