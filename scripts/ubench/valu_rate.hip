// Issue rate of the VALU instructions K2's f16x3 operand split can be built from.
// One workgroup of 256 threads per CU (1 wave / SIMD) or 512 (2 waves / SIMD); every wave
// runs ITER x 32 independent instructions of one kind; cycles per instruction per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
constexpr int ITER = 4096;

#define KERNEL(name, body)                                                          \
  __global__ void name(unsigned* out, long long* cyc) {                             \
    unsigned v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7; \
    unsigned a = 0x3c003c00u + threadIdx.x, b = 0x40004000u;                       \
    long long t0 = wall_clock64();                                                  \
    for (int i = 0; i < ITER; ++i) {                                                \
      asm volatile(REP8(body) REP8(body) REP8(body) REP8(body)                      \
                   : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) \
                   : "v"(a), "v"(b));                                               \
    }                                                                               \
    long long t1 = wall_clock64();                                                  \
    out[blockIdx.x * blockDim.x + threadIdx.x] = v0 ^ v1 ^ v2 ^ v3 ^ v4 ^ v5 ^ v6 ^ v7; \
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;                        \
  }

// 8 instructions per body, each on its own destination register (no dependency stalls within 8)
KERNEL(k_max_i32, "v_max_i32 %0, %8, %0\n v_max_i32 %1, %8, %1\n v_max_i32 %2, %8, %2\n v_max_i32 %3, %8, %3\n v_max_i32 %4, %8, %4\n v_max_i32 %5, %8, %5\n v_max_i32 %6, %8, %6\n v_max_i32 %7, %8, %7\n")
KERNEL(k_max3_i32, "v_max3_i32 %0, %8, %9, %0\n v_max3_i32 %1, %8, %9, %1\n v_max3_i32 %2, %8, %9, %2\n v_max3_i32 %3, %8, %9, %3\n v_max3_i32 %4, %8, %9, %4\n v_max3_i32 %5, %8, %9, %5\n v_max3_i32 %6, %8, %9, %6\n v_max3_i32 %7, %8, %9, %7\n")
KERNEL(k_pk_max_i16, "v_pk_max_i16 %0, %8, %0\n v_pk_max_i16 %1, %8, %1\n v_pk_max_i16 %2, %8, %2\n v_pk_max_i16 %3, %8, %3\n v_pk_max_i16 %4, %8, %4\n v_pk_max_i16 %5, %8, %5\n v_pk_max_i16 %6, %8, %6\n v_pk_max_i16 %7, %8, %7\n")
KERNEL(k_cvt_pkrtz, "v_cvt_pkrtz_f16_f32 %0, %8, %9\n v_cvt_pkrtz_f16_f32 %1, %8, %9\n v_cvt_pkrtz_f16_f32 %2, %8, %9\n v_cvt_pkrtz_f16_f32 %3, %8, %9\n v_cvt_pkrtz_f16_f32 %4, %8, %9\n v_cvt_pkrtz_f16_f32 %5, %8, %9\n v_cvt_pkrtz_f16_f32 %6, %8, %9\n v_cvt_pkrtz_f16_f32 %7, %8, %9\n")
KERNEL(k_cvt_f32_f16, "v_cvt_f32_f16 %0, %8\n v_cvt_f32_f16 %1, %8\n v_cvt_f32_f16 %2, %8\n v_cvt_f32_f16 %3, %8\n v_cvt_f32_f16 %4, %8\n v_cvt_f32_f16 %5, %8\n v_cvt_f32_f16 %6, %8\n v_cvt_f32_f16 %7, %8\n")
KERNEL(k_cvt_f32_f16_sdwa, "v_cvt_f32_f16_sdwa %0, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_f16_sdwa %1, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_f16_sdwa %2, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_f16_sdwa %3, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_f16_sdwa %4, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_f16_sdwa %5, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_f16_sdwa %6, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_f16_sdwa %7, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n")
KERNEL(k_fma_mixlo, "v_fma_mixlo_f16 %0, %8, -1.0, %9 op_sel_hi:[1,0,0]\n v_fma_mixlo_f16 %1, %8, -1.0, %9 op_sel_hi:[1,0,0]\n v_fma_mixlo_f16 %2, %8, -1.0, %9 op_sel_hi:[1,0,0]\n v_fma_mixlo_f16 %3, %8, -1.0, %9 op_sel_hi:[1,0,0]\n v_fma_mixlo_f16 %4, %8, -1.0, %9 op_sel_hi:[1,0,0]\n v_fma_mixlo_f16 %5, %8, -1.0, %9 op_sel_hi:[1,0,0]\n v_fma_mixlo_f16 %6, %8, -1.0, %9 op_sel_hi:[1,0,0]\n v_fma_mixlo_f16 %7, %8, -1.0, %9 op_sel_hi:[1,0,0]\n")
KERNEL(k_fma_mix_f32, "v_fma_mix_f32 %0, %8, -1.0, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %8, -1.0, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %2, %8, -1.0, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %8, -1.0, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %4, %8, -1.0, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %5, %8, -1.0, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %6, %8, -1.0, %9 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %7, %8, -1.0, %9 op_sel_hi:[1,0,0]\n")
KERNEL(k_and, "v_and_b32 %0, %8, %0\n v_and_b32 %1, %8, %1\n v_and_b32 %2, %8, %2\n v_and_b32 %3, %8, %3\n v_and_b32 %4, %8, %4\n v_and_b32 %5, %8, %5\n v_and_b32 %6, %8, %6\n v_and_b32 %7, %8, %7\n")
KERNEL(k_sub_f32, "v_sub_f32 %0, %8, %0\n v_sub_f32 %1, %8, %1\n v_sub_f32 %2, %8, %2\n v_sub_f32 %3, %8, %3\n v_sub_f32 %4, %8, %4\n v_sub_f32 %5, %8, %5\n v_sub_f32 %6, %8, %6\n v_sub_f32 %7, %8, %7\n")
KERNEL(k_cvt_pk_rne, "v_cvt_pk_f16_f32 %0, %8, %9\n v_cvt_pk_f16_f32 %1, %8, %9\n v_cvt_pk_f16_f32 %2, %8, %9\n v_cvt_pk_f16_f32 %3, %8, %9\n v_cvt_pk_f16_f32 %4, %8, %9\n v_cvt_pk_f16_f32 %5, %8, %9\n v_cvt_pk_f16_f32 %6, %8, %9\n v_cvt_pk_f16_f32 %7, %8, %9\n")
KERNEL(k_pk_max_f16, "v_pk_max_f16 %0, %8, %0\n v_pk_max_f16 %1, %8, %1\n v_pk_max_f16 %2, %8, %2\n v_pk_max_f16 %3, %8, %3\n v_pk_max_f16 %4, %8, %4\n v_pk_max_f16 %5, %8, %5\n v_pk_max_f16 %6, %8, %6\n v_pk_max_f16 %7, %8, %7\n")

// 64-bit operand forms need register pairs: separate kernel shape
__global__ void k_pk_add_f32(unsigned* out, long long* cyc) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 v0 = {1.f * threadIdx.x, 2.f}, v1 = v0 + 1.f, v2 = v0 + 2.f, v3 = v0 + 3.f, v4 = v0 + 4.f, v5 = v0 + 5.f, v6 = v0 + 6.f, v7 = v0 + 7.f;
  f2 a = {0.5f, 0.25f};
  long long t0 = wall_clock64();
  for (int i = 0; i < ITER; ++i) {
    asm volatile(REP8("v_pk_add_f32 %0, %8, %0\n v_pk_add_f32 %1, %8, %1\n v_pk_add_f32 %2, %8, %2\n v_pk_add_f32 %3, %8, %3\n") REP8("v_pk_add_f32 %4, %8, %4\n v_pk_add_f32 %5, %8, %5\n v_pk_add_f32 %6, %8, %6\n v_pk_add_f32 %7, %8, %7\n")
                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(a));
  }
  long long t1 = wall_clock64();
  f2 s = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
  out[blockIdx.x * blockDim.x + threadIdx.x] = __float_as_uint(s[0] + s[1]);
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <typename K>
void run(const char* name, K kern, int threads, int per_iter) {
  unsigned* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  kern<<<256, threads>>>(out, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  kern<<<256, threads>>>(out, cyc);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double n = double(ITER) * per_iter;
  // wall_clock64 ticks at 100 MHz: ns per instruction per wave, from both clocks
  printf("%-20s %d waves/SIMD: %.3f ns/instr/wave (event) %.3f ns (wall_clock64)\n", name, threads / 256,
         ms * 1e6 / n, c * 10.0 / n);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int threads : {256, 512}) {
    run("v_max_i32", k_max_i32, threads, 32);
    run("v_max3_i32", k_max3_i32, threads, 32);
    run("v_pk_max_i16", k_pk_max_i16, threads, 32);
    run("v_pk_max_f16", k_pk_max_f16, threads, 32);
    run("v_and_b32", k_and, threads, 32);
    run("v_sub_f32", k_sub_f32, threads, 32);
    run("v_cvt_pkrtz_f16_f32", k_cvt_pkrtz, threads, 32);
    run("v_cvt_pk_f16_f32", k_cvt_pk_rne, threads, 32);
    run("v_cvt_f32_f16", k_cvt_f32_f16, threads, 32);
    run("v_cvt_f32_f16_sdwa", k_cvt_f32_f16_sdwa, threads, 32);
    run("v_fma_mixlo_f16", k_fma_mixlo, threads, 32);
    run("v_fma_mix_f32", k_fma_mix_f32, threads, 32);
    run("v_pk_add_f32", k_pk_add_f32, threads, 64);
  }
  return 0;
}
