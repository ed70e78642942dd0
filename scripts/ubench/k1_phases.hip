// Where does K1 (reweight + resample) spend its time?  Includes the product kernel source with
// MMF_K1_PHASE_CLOCKS: thread 0 of every workgroup stamps s_memtime (100 MHz) at entry, after the row
// maximum (pass 1: log-weights in, max reduction), after the CDF (pass 2: exp, sums, scan), and after the
// search + gather + store loop.  Read SHARES, not lengths (the stamps fence overlaps).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Imultimodalfilter_amd/csrc \
//         -o scripts/ubench/k1_phases scripts/ubench/k1_phases.hip
#define MMF_K1_PHASE_CLOCKS 1
#include "../../multimodalfilter_amd/csrc/pf_resample.hip"

#include <cstdio>
#include <cstdlib>
#include <vector>

static float* dev_random(size_t n, float scale, float off = 0.f) {
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = off + scale * (2.f * rand() / RAND_MAX - 1.f);
  float* d;
  hipMalloc(&d, n * 4);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  return d;
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 256, M = argc > 2 ? atoi(argv[2]) : 4096, d = 3;
  float* ll = dev_random(size_t(N) * M, 2.0f);
  float* x = dev_random(size_t(N) * M * d, 1.0f);
  float* u = dev_random(N, 0.49f, 0.5f);
  float *est, *xo;
  hipMalloc(&est, N * d * 4);
  hipMalloc(&xo, size_t(N) * M * d * 4);
  for (int rep = 0; rep < 4; ++rep) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    if (mmf_pf_reweight_resample(ll, nullptr, x, u, est, xo, nullptr, nullptr, N, M, M, d, 1, nullptr)) return 2;
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    static unsigned long long st[1024][8];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(g_k1_stamps), sizeof(st));
    double ph[3] = {0, 0, 0};
    unsigned long long first = ~0ull, last = 0;
    const int nb = N < 1024 ? N : 1024;
    for (int b = 0; b < nb; ++b) {
      ph[0] += (st[b][1] - st[b][0]) * 0.01 / nb;
      ph[1] += (st[b][2] - st[b][1]) * 0.01 / nb;
      ph[2] += (st[b][3] - st[b][2]) * 0.01 / nb;
      if (st[b][0] < first) first = st[b][0];
      if (st[b][3] > last) last = st[b][3];
    }
    printf("rep %d: %d x %d in %.2f us (events); per workgroup: load+max %.2f us, exp+sums+scan %.2f us, search+gather+store %.2f us; first start -> last end %.2f us\n",
           rep, N, M, ms * 1e3, ph[0], ph[1], ph[2], (last - first) * 0.01);
  }
  return 0;
}
