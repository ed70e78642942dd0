// How many VALU instructions of K2's operand-split mix hide in the shadow of one wave's own
// v_mfma_f32_32x32x16_f16 stream?  One wave per SIMD (256 threads per CU) and two waves per
// SIMD (512); every wave runs ITER x 4 groups of {1 MFMA + K fillers}; the MFMAs rotate over 4
// independent accumulators; the fillers touch no MFMA register.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int ITER = 2048;

#define F0 "v_max_i32 %4, 0, %4\n"
#define F1 "v_cvt_pkrtz_f16_f32 %5, %12, %13\n"
#define F2 "v_fma_mix_f32 %6, %5, -1.0, %12 op_sel_hi:[1,0,0]\n"
#define F3 "v_fma_mix_f32 %7, %5, -1.0, %13 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
#define F4 "v_pk_max_i16 %8, %8, %5\n"
#define F5 "v_max_i32 %9, 0, %9\n"
#define F6 "v_cvt_pkrtz_f16_f32 %10, %12, %13\n"
#define F7 "v_max_i32 %11, 0, %11\n"

#define GROUP(acc, fill) "v_mfma_f32_32x32x16_f16 " acc ", %14, %15, " acc "\n" fill

template <int K>
__global__ void kern(float* out, long long* cyc) {
  f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
  half8 A, B;
  for (int i = 0; i < 8; ++i) { A[i] = (_Float16)(0.001f * threadIdx.x + i); B[i] = (_Float16)(1.f + 0.5f * i); }
  unsigned v0 = threadIdx.x, v1 = 1, v2 = 2, v3 = 3, v4 = 4, v5 = 5, v6 = 6, v7 = 7;
  float x0 = 1.5f * threadIdx.x, x1 = 0.25f * threadIdx.x;
  const long long t0 = wall_clock64();
  for (int i = 0; i < ITER; ++i) {
#define FILL                                                                                   \
  ((K > 0) ? F0 : "") ((K > 1) ? F1 : "")((K > 2) ? F2 : "")((K > 3) ? F3 : "")((K > 4) ? F4 : "") \
      ((K > 5) ? F5 : "")((K > 6) ? F6 : "")((K > 7) ? F7 : "")
    // string literals cannot be selected by a constant expression: spell the variants out
    if constexpr (K == 0)
      asm volatile(GROUP("%0", "") GROUP("%1", "") GROUP("%2", "") GROUP("%3", "")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7)
                   : "v"(x0), "v"(x1), "v"(A), "v"(B));
#define VARIANT(k, fill)                                                                        \
    if constexpr (K == k)                                                                       \
      asm volatile(GROUP("%0", fill) GROUP("%1", fill) GROUP("%2", fill) GROUP("%3", fill)      \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) \
                   : "v"(x0), "v"(x1), "v"(A), "v"(B));
    VARIANT(1, F0) VARIANT(2, F0 F1) VARIANT(3, F0 F1 F2) VARIANT(4, F0 F1 F2 F3) VARIANT(5, F0 F1 F2 F3 F4)
    VARIANT(6, F0 F1 F2 F3 F4 F5) VARIANT(7, F0 F1 F2 F3 F4 F5 F6) VARIANT(8, F0 F1 F2 F3 F4 F5 F6 F7)
  }
  const long long t1 = wall_clock64();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i] + a3[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int K>
void run(float* out, long long* cyc, int threads) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  kern<K><<<256, threads>>>(out, cyc); hipDeviceSynchronize();
  hipEventRecord(e0); kern<K><<<256, threads>>>(out, cyc); hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double per_mfma_ns = ms * 1e6 / (ITER * 4.0) / (threads / 256);
  printf("K=%d fillers/MFMA, %d wave(s)/SIMD: %.2f ns per MFMA per SIMD (%.1f cyc @2.4GHz)\n", K, threads / 256,
         per_mfma_ns, per_mfma_ns * 2.4);
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
  for (int threads : {256, 512}) {
    run<0>(out, cyc, threads); run<1>(out, cyc, threads); run<2>(out, cyc, threads); run<3>(out, cyc, threads);
    run<4>(out, cyc, threads); run<5>(out, cyc, threads); run<6>(out, cyc, threads); run<7>(out, cyc, threads);
    run<8>(out, cyc, threads);
  }
  return 0;
}
