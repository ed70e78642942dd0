// One-way latency of a tagged-granule hand-off between two workgroups (the persistent loop's mechanism,
// csrc/pf_persistent.inc), by placement: partner = blockIdx + stride.  Workgroups are dispatched round-robin over the
// eight XCDs, so stride 8 pairs land on ONE XCD (one L2) and stride 1 pairs on neighbouring XCDs; the kernel reports
// each pair's XCC_ID so the assumption is checked, not trusted.  Scopes: agent (what the library uses) and workgroup-
// scope loads that only bypass nothing (would be wrong across XCDs; shown for the latency of an L2 hit).
//   hipcc --offload-arch=gfx950 -O3 -o handoff_latency handoff_latency.hip && ./handoff_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

using Granule = unsigned long long;

template <int SCOPE>
__device__ __forceinline__ Granule ld(const Granule* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, SCOPE);
}
template <int SCOPE>
__device__ __forceinline__ void st(Granule* p, Granule v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, SCOPE);
}

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}

// pair p = (a, b): a = first of the pair.  Round trips: a writes ping[k], b waits for it and writes pong[k], a waits.
template <int SCOPE>
__global__ void pingpong(Granule* ping, Granule* pong, int rounds, int stride, int pairs, unsigned* xcc, long long* cycles) {
  const int b = blockIdx.x;
  // block layout: groups of 2 * stride blocks; inside a group block j < stride is "a" of pair j, block j + stride its "b"
  const int group = b / (2 * stride), j = b % (2 * stride);
  const bool is_a = j < stride;
  const int pair = group * stride + (is_a ? j : j - stride);
  if (pair >= pairs) return;
  if (threadIdx.x == 0) xcc[b] = xcc_id();
  Granule* pi = ping + pair * 64 + threadIdx.x;  // 64 lanes = 64 granules = 512 B per hand-off (like a tile row block)
  Granule* po = pong + pair * 64 + threadIdx.x;
  const long long t0 = clock64();
  for (int k = 1; k <= rounds; ++k) {
    if (is_a) {
      st<SCOPE>(pi, (static_cast<Granule>(k) << 32) | threadIdx.x);
      while (!__all((ld<SCOPE>(po) >> 32) == static_cast<Granule>(k))) {}
    } else {
      while (!__all((ld<SCOPE>(pi) >> 32) == static_cast<Granule>(k))) {}
      st<SCOPE>(po, (static_cast<Granule>(k) << 32) | threadIdx.x);
    }
  }
  if (threadIdx.x == 0 && is_a) cycles[pair] = clock64() - t0;
}

template <int SCOPE>
void run(const char* name, int stride, int pairs, int rounds) {
  Granule *ping, *pong;
  unsigned* xcc;
  long long* cyc;
  const int blocks = ((pairs + stride - 1) / stride) * 2 * stride;
  hipMalloc(&ping, pairs * 64 * sizeof(Granule));
  hipMalloc(&pong, pairs * 64 * sizeof(Granule));
  hipMalloc(&xcc, blocks * sizeof(unsigned));
  hipMalloc(&cyc, pairs * sizeof(long long));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipMemset(ping, 0, pairs * 64 * sizeof(Granule));
    hipMemset(pong, 0, pairs * 64 * sizeof(Granule));
    hipEventRecord(e0);
    pingpong<SCOPE><<<blocks, 64>>>(ping, pong, rounds, stride, pairs, xcc, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  std::vector<unsigned> hx(blocks);
  hipMemcpy(hx.data(), xcc, blocks * sizeof(unsigned), hipMemcpyDeviceToHost);
  int same = 0;
  for (int p = 0; p < pairs; ++p) {
    const int group = p / stride, j = p % stride;
    same += hx[group * 2 * stride + j] == hx[group * 2 * stride + j + stride];
  }
  printf("%-34s stride %2d, %3d pairs (%3d on one XCD): %.3f us per one-way hand-off\n", name, stride, pairs, same,
         1e3 * best / (2.0 * rounds));
  hipFree(ping); hipFree(pong); hipFree(xcc); hipFree(cyc);
}

int main() {
  const int rounds = 2000;
  for (int pairs : {1, 32, 100}) {
    run<__HIP_MEMORY_SCOPE_AGENT>("agent scope (library)", 1, pairs, rounds);
    run<__HIP_MEMORY_SCOPE_AGENT>("agent scope (library)", 8, pairs, rounds);
    run<__HIP_MEMORY_SCOPE_AGENT>("agent scope (library)", 4, pairs, rounds);
    run<__HIP_MEMORY_SCOPE_SYSTEM>("system scope", 8, pairs, rounds);
  }
  return 0;
}
