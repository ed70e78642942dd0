// Which VALU instructions co-execute with another wave's f16 MFMAs on the same SIMD?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int OP>
__device__ __forceinline__ void valu_work(float (&x)[16], int n) {
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int rep = 0; rep < 4; ++rep)
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      float x0 = x[2 * p], x1 = x[2 * p + 1];
      if (OP == 0) { x0 = fmaf(x0, 1.0001f, 0.5f); x1 = fmaf(x1, 0.9999f, 0.25f); }            // v_fma_f32
      if (OP == 1) { auto h = __builtin_amdgcn_cvt_pkrtz(x0, x1); x0 = __builtin_bit_cast(float, h); x1 = x0; }  // cvt_pkrtz (+mov)
      if (OP == 2) { auto h = __builtin_bit_cast(__attribute__((ext_vector_type(2))) _Float16, x0); x0 = (float)h[0]; x1 = (float)h[1]; } // cvt_f32_f16 x2
      if (OP == 3) { f32x2 a = {x0, x1}, b = {1.5f, 2.5f}; f32x2 r = a - b; x0 = r[0]; x1 = r[1]; }  // v_pk_add_f32
      if (OP == 4) { x0 = __int_as_float(max(__float_as_int(x0), 3)); x1 = __int_as_float(max(__float_as_int(x1), 5)); }  // v_max_i32
      if (OP == 5) { x0 = __int_as_float(__float_as_int(x0) & 0xffffe000); x1 = __int_as_float(__float_as_int(x1) & 0xffffe000) ; } // v_and
      if (OP == 6) { x0 = x0 - 1.5f; x1 = x1 - 2.5f; }  // v_sub_f32
      x[2 * p] = x0; x[2 * p + 1] = x1;
    }
  }
}
__device__ __forceinline__ void mfma_work(f32x16& acc, half8 a, half8 b, int n) {
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  }
}
template <int OP>
__global__ __launch_bounds__(512) void k(float* out, int mode, int n) {
  const int wave = threadIdx.x >> 6;
  f32x16 acc = {};
  half8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(1.0f + i * 0.5f); }
  float x[16];
  for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 0.37f + i;
  const bool lower = wave < 4;
  if ((mode == 2 || mode == 3) && lower) mfma_work(acc, a, b, n);
  if ((mode == 2 || mode == 4) && !lower) valu_work<OP>(x, n);
  float s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i] + x[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int OP> void run(float* out, const char* name) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int n = 1000; float t[5] = {0};
  for (int mode = 2; mode < 5; ++mode) {
    k<OP><<<256, 512>>>(out, mode, n); hipDeviceSynchronize();
    hipEventRecord(e0); k<OP><<<256, 512>>>(out, mode, n); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&t[mode], e0, e1);
  }
  printf("%-14s mfma-only %.3f  valu-only %.3f  both %.3f ms  -> overlap %.0f%%\n", name, t[3], t[4], t[2],
         100.0 * (t[3] + t[4] - t[2]) / (t[3] < t[4] ? t[3] : t[4]));
}
int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  run<0>(out, "v_fma_f32"); run<1>(out, "cvt_pkrtz"); run<2>(out, "cvt_f32_f16"); run<3>(out, "v_pk_add_f32");
  run<4>(out, "v_max_i32"); run<5>(out, "v_and_b32"); run<6>(out, "v_sub_f32");
  return 0;
}
