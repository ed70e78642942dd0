// How fast can the split-K linear layer's activation stream be read?  Pattern A = fc_partial_f16x3_kernel's (lane (j, h) of a wave reads
// 32 B of image j's row per k-step: 64-byte pieces of 32 rows 32 KB apart); pattern B = the same bytes from an image-tiled layout
// [tile of 32 images][k / 8][image][8 floats] (a wave's two loads of a k-step cover 2 KB contiguously).  Grid, block and loop shape
// as the product kernel's (N / 128, 16 K slices, encoders) x 256 threads, 32 k-steps, loads unrolled by 4; values are only summed.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/fc_read_pattern scripts/ubench/fc_read_pattern.hip && ./scripts/ubench/fc_read_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int K = 8192, SPLIT = 16, KS = K / SPLIT / 16;
template <bool TILED>
__global__ __launch_bounds__(256) void rd(const float* __restrict__ act, float* __restrict__ out, int N) {
  const int split = blockIdx.y, net = blockIdx.z, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
  const int img0 = (blockIdx.x * 4 + wave) * 32;
  if (img0 >= N) return;
  const float* X;
  size_t stride;
  if (TILED) {
    X = act + (static_cast<size_t>(net) * N + img0) * K + (static_cast<size_t>(split * 64 + h) * 32 + j) * 8;
    stride = 2 * 32 * 8;  // two k-groups of 32 images x 8 floats per k-step
  } else {
    X = act + (static_cast<size_t>(net) * N + img0 + j) * K + split * (K / SPLIT) + 8 * h;
    stride = 16;
  }
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (int ks = 0; ks < KS; ++ks) {
    s += *reinterpret_cast<const f32x4*>(X + ks * stride);
    s += *reinterpret_cast<const f32x4*>(X + ks * stride + 4);
  }
  if (s[0] + s[1] + s[2] + s[3] == 12345.f) out[0] = 1.f;
}
int main() {
  const int N = 4096, nets = 2;
  float *act, *out;
  hipMalloc(&act, sizeof(float) * static_cast<size_t>(nets) * N * K);
  hipMalloc(&out, 4);
  hipMemset(act, 0, sizeof(float) * static_cast<size_t>(nets) * N * K);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 3; ++rep)
    for (int tiled = 0; tiled < 2; ++tiled) {
      float best = 1e9f;
      for (int it = 0; it < 10; ++it) {
        hipEventRecord(a);
        if (tiled) rd<true><<<dim3(N / 128, SPLIT, nets), 256>>>(act, out, N);
        else rd<false><<<dim3(N / 128, SPLIT, nets), 256>>>(act, out, N);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        best = ms < best ? ms : best;
      }
      printf("%s: %.1f us for %.0f MB = %.2f TB/s\n", tiled ? "B image-tiled layout " : "A one row per lane    ", 1e3 * best,
             4.0 * nets * N * K / 1e6, 4.0 * nets * N * K / (best * 1e-3) / 1e12);
    }
  return 0;
}
