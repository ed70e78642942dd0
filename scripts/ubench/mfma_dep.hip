// Dependent vs independent accumulator chains of v_mfma_f32_32x32x16_f16 (one wave per SIMD and two).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int ITER = 2048;
#define M(acc) "v_mfma_f32_32x32x16_f16 " acc ", %4, %5, " acc "\n"
template <int CHAIN>  // CHAIN consecutive MFMAs on one accumulator before moving to the next (of 4)
__global__ void kern(float* out) {
  f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
  half8 A, B;
  for (int i = 0; i < 8; ++i) { A[i] = (_Float16)(0.001f * threadIdx.x + i); B[i] = (_Float16)(1.f + 0.5f * i); }
  for (int i = 0; i < ITER; ++i) {
    if constexpr (CHAIN == 1)
      asm volatile(M("%0") M("%1") M("%2") M("%3") M("%0") M("%1") M("%2") M("%3") M("%0") M("%1") M("%2") M("%3")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(A), "v"(B));
    if constexpr (CHAIN == 3)
      asm volatile(M("%0") M("%0") M("%0") M("%1") M("%1") M("%1") M("%2") M("%2") M("%2") M("%3") M("%3") M("%3")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(A), "v"(B));
    if constexpr (CHAIN == 2)  // K2 today at CT=2: 3 on acc0, 3 on acc1, alternating
      asm volatile(M("%0") M("%0") M("%0") M("%1") M("%1") M("%1") M("%0") M("%0") M("%0") M("%1") M("%1") M("%1")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(A), "v"(B));
    if constexpr (CHAIN == 12)
      asm volatile(M("%0") M("%0") M("%0") M("%0") M("%0") M("%0") M("%0") M("%0") M("%0") M("%0") M("%0") M("%0")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(A), "v"(B));
    if constexpr (CHAIN == 6)  // alternate two accumulators every MFMA
      asm volatile(M("%0") M("%1") M("%0") M("%1") M("%0") M("%1") M("%0") M("%1") M("%0") M("%1") M("%0") M("%1")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(A), "v"(B));
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i] + a3[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CHAIN> void run(float* out, int threads, const char* what) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  kern<CHAIN><<<256, threads>>>(out); hipDeviceSynchronize();
  hipEventRecord(e0); kern<CHAIN><<<256, threads>>>(out); hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s %d wave(s)/SIMD: %.2f ns per MFMA per SIMD\n", what, threads / 256, ms * 1e6 / (ITER * 12.0) / (threads / 256));
}
int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  for (int threads : {256, 512}) {
    run<1>(out, threads, "rotate 4 accumulators");
    run<6>(out, threads, "alternate 2 accumulators");
    run<3>(out, threads, "3 in a row per accumulator, 4 accumulators");
    run<2>(out, threads, "3 in a row per accumulator, 2 accumulators");
    run<12>(out, threads, "one accumulator");
  }
  return 0;
}
