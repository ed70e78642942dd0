#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* p) {
  unsigned a = threadIdx.x, b = 100 + threadIdx.x;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  p[threadIdx.x] = r[0];
  p[threadIdx.x + 64] = r[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 128 * 4);
  k<<<1, 64>>>(d);
  unsigned h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("a=lane, b=100+lane; r = permlane32_swap(a, b)\n");
  for (int l : {0, 1, 31, 32, 33, 63}) printf("lane %2d: r[0]=%3u r[1]=%3u\n", l, h[l], h[l + 64]);
  return 0;
}
