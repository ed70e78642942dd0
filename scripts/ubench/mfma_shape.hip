// Which f16 MFMA shape holds the higher clock under a K2-like load?  Same FLOPs per loop body,
// weights re-read from LDS every step (ds_read_b128), operands random, 2 waves per SIMD on every CU:
//   A: 32x32x16  (16 MFMAs per layer-tile: 2 row tiles x 2 column tiles x 4 k-steps)
//   B: 16x16x32  (32 MFMAs: 4 x 4 x 2), same 64 x 64 x 64 product per wave
// plus VALU filler per MFMA step (FILL instructions on live registers) to mimic the operand split.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o scripts/ubench/mfma_shape scripts/ubench/mfma_shape.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

using half8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

// round 6: the shader clock the loop holds (s_memtime ticks per s_memrealtime tick of 100 MHz), stamped by one lane per workgroup
__device__ unsigned long long g_clk[256][4];

template <int SHAPE, int FILL>
__global__ __launch_bounds__(512) void k(const float* __restrict__ w, float* __restrict__ out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = threadIdx.x; i < 65536 / 16; i += 512) reinterpret_cast<float4*>(lds)[i] = reinterpret_cast<const float4*>(w)[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  // B operands: random-ish f16 values in registers
  half8 b[8];
  for (int i = 0; i < 8; ++i)
    for (int e = 0; e < 8; ++e) b[i][e] = static_cast<_Float16>(0.01f * ((lane * 131 + i * 17 + e * 7) % 97) - 0.4f);
  float filler[8];
  for (int i = 0; i < 8; ++i) filler[i] = 0.001f * (lane + i);
  if (SHAPE == 0) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const half8 a = *reinterpret_cast<const half8*>(lds + ((it * 16 + g) & 63) * 1024 + lane * 16);
        acc[g & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[g & 7], acc[g & 3], 0, 0, 0);
#pragma unroll
        for (int f = 0; f < FILL; ++f) filler[f & 7] = __builtin_fmaf(filler[f & 7], 1.0001f, 0.5f);
      }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 8; ++i) s += filler[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  } else {
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int g = 0; g < 32; ++g) {
        const half8 a = *reinterpret_cast<const half8*>(lds + ((it * 32 + g) & 63) * 1024 + lane * 16);
        acc[g & 15] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b[g & 7], acc[g & 15], 0, 0, 0);
#pragma unroll
        for (int f = 0; f < (FILL + (g & 1)) / 2; ++f) filler[f & 7] = __builtin_fmaf(filler[f & 7], 1.0001f, 0.5f);
      }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    for (int i = 0; i < 8; ++i) s += filler[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  }
  if (threadIdx.x == 0) {
    g_clk[blockIdx.x][0] = c0; g_clk[blockIdx.x][1] = __builtin_amdgcn_s_memtime();
    g_clk[blockIdx.x][2] = r0; g_clk[blockIdx.x][3] = __builtin_amdgcn_s_memrealtime();
  }
}

template <int SHAPE, int FILL>
void run(const float* w, float* out) {
  const int iters = 4000;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<SHAPE, FILL>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    k<SHAPE, FILL><<<256, 512, 65536>>>(w, out, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = 256.0 * 8 * iters * 16 * 2.0 * 32 * 32 * 16;  // per wave and iteration: 16 MFMAs of 32x32x16 (or 32 of 16x16x32)
    if (rep == 2) {
      static unsigned long long h[256][4];
      hipMemcpyFromSymbol(h, HIP_SYMBOL(g_clk), sizeof(h));
      std::vector<double> ghz;
      for (int b = 0; b < 256; ++b) ghz.push_back(0.1 * double(h[b][1] - h[b][0]) / double(h[b][3] - h[b][2]));
      std::sort(ghz.begin(), ghz.end());
      // matrix-pipe cycles the loop needs per SIMD: 2 waves x iters x 16 MFMAs x 32 cycles (or 32 x 16)
      const double need = 2.0 * iters * 16 * 32, have = ghz[128] * 1e9 * ms * 1e-3;
      printf("shape %s fill %d: %.3f ms  %.1f TFLOP/s; shader clock in the loop %.2f GHz (median of 256 workgroups, %.2f .. %.2f): "
             "the matrix pipe is busy %.0f %% of those cycles\n", SHAPE ? "16x16x32" : "32x32x16", FILL, ms, flop / ms / 1e9, ghz[128], ghz[0], ghz[255],
             100.0 * need / have);
    }
  }
}

int main() {
  std::vector<float> h(65536 / 4);
  for (auto& x : h) x = 0.01f * (rand() % 200 - 100);
  float *w, *out; hipMalloc(&w, 65536); hipMalloc(&out, 256 * 512 * 4);
  hipMemcpy(w, h.data(), 65536, hipMemcpyHostToDevice);
  run<0, 0>(w, out); run<1, 0>(w, out);
  run<0, 5>(w, out); run<1, 5>(w, out);
  run<0, 7>(w, out); run<1, 7>(w, out);
  return 0;
}
