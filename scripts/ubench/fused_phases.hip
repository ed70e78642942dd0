// Where does the fused training kernel (csrc/particle_net_fused.hip) spend a tile?  Includes the product source
// with MMF_FUSED_PHASE_CLOCKS: wave 0 of workgroup 0 accumulates s_memtime (100 MHz) differences per phase.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Iinclude -Imultimodalfilter_amd/csrc \
//         -o scripts/ubench/fused_phases scripts/ubench/fused_phases.hip
// The blob is random bits (finite f16 halves): the arithmetic is garbage, the instruction stream is the product's.
#ifndef NO_PHASE_CLOCKS
#define MMF_FUSED_PHASE_CLOCKS 1
#endif
#include "../../multimodalfilter_amd/csrc/particle_net_fused.hip"

#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>

static float* dev_random(size_t n, float scale) {
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = scale * (2.f * rand() / RAND_MAX - 1.f);
  float* d;
  hipMalloc(&d, n * 4);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  return d;
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 32, M = argc > 2 ? atoi(argv[2]) : 8192, d = 3, kind = argc > 3 ? atoi(argv[3]) : 1;
  const int n_res_arg = argc > 4 ? atoi(argv[4]) : -1;  // experiment: a measurement network with fewer blocks (register pressure)
  const int n_res = n_res_arg >= 0 ? n_res_arg : (kind == 1 ? 2 : 3), NL = 3 + 2 * n_res;
  const size_t R = size_t(N) * M;
  // blob: small f16 halves (|w| < 0.1) packed two per float; the fp32 sections (W0, biases, head) small floats
  std::vector<float> hb(blob_floats(n_res));
  for (auto& v : hb) v = 0.05f * (2.f * rand() / RAND_MAX - 1.f);
  for (int i = off_layers(); i < off_bias(n_res); ++i) {
    const _Float16 a = static_cast<_Float16>(0.1f * (2.f * rand() / RAND_MAX - 1.f)), b = static_cast<_Float16>(0.1f * (2.f * rand() / RAND_MAX - 1.f));
    unsigned short ua, ub;
    memcpy(&ua, &a, 2); memcpy(&ub, &b, 2);
    const unsigned w = ua | (unsigned(ub) << 16);
    memcpy(&hb[i], &w, 4);
  }
  float* blob;
  hipMalloc(&blob, hb.size() * 4);
  hipMemcpy(blob, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
  MmfTrainFusedArgs c{};
  c.packed_dual = blob; c.n_res = n_res; c.kind = kind; c.d = d; c.N = N; c.M = M; c.n_slots = 256;
  c.states = dev_random(R * d, 1.f); c.traj_bias = dev_random(size_t(N) * 64, 1.f);
  c.d_out = dev_random(R, 1e-3f); c.g_next = dev_random(R * d, 1e-3f);
  float* scratch = dev_random(R * 64 * 6 + R * 16, 0.f);
  c.d_raw = scratch; c.act = scratch + R * 4; c.g_act = c.act + R * 64; c.d_states = c.g_act + R * 64;
  c.dz_first_h = c.d_states + R * 4; c.dz_join_h = (float*)c.dz_first_h + R * 32; c.h_last_h = (float*)c.dz_join_h + R * 32;
  c.sc_first = (float*)c.h_last_h + R * 32; c.sc_join = c.sc_first + R;
  hipMalloc(&c.pw, size_t(NL) * 256 * 4096 * 4); hipMemset(c.pw, 0, size_t(NL) * 256 * 4096 * 4);
  hipMalloc(&c.pb, size_t(NL) * 256 * 64 * 4); hipMemset(c.pb, 0, size_t(NL) * 256 * 64 * 4);
  for (int rep = 0; rep < 4; ++rep) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    int rc;
    if (n_res_arg >= 0 && kind == 1) {
      FusedArgs a{};
      a.blob = c.packed_dual; a.states = c.states; a.traj_bias = c.traj_bias; a.d_out = c.d_out; a.d_states = c.d_states;
      a.dz_first_h = static_cast<_Float16*>(c.dz_first_h); a.sc_first = c.sc_first; a.dz_join_h = static_cast<_Float16*>(c.dz_join_h);
      a.sc_join = c.sc_join; a.h_last_h = static_cast<_Float16*>(c.h_last_h); a.pw = c.pw; a.pb = c.pb; a.R = N * M; a.M = M; a.slots = 256;
      rc = n_res_arg == 1 ? launch_fused<3, 1, kMeasure, kFull>(a, nullptr) : n_res_arg == 0 ? launch_fused<3, 0, kMeasure, kFull>(a, nullptr)
                                                                                  : launch_fused<3, 2, kMeasure, kFull>(a, nullptr);
    } else {
      rc = mmf_particle_net_train_fused(&c, nullptr);
    }
    if (rc) { printf("rc %d\n", rc); return 2; }
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("rep %d: kind %d, %d x %d rows: %.1f us", rep, kind, N, M, ms * 1e3);
#ifdef MMF_FUSED_PHASE_CLOCKS
    unsigned long long ph[16];
    hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_fused_phases), sizeof(ph));
    static const char* names[10] = {"tile top", "forward", "between layers", "bwd prologue", "W^T product", "barrier A", "exchange+epilogue", "barrier B", "dW products", "first layer + d states"};
    double tot = 0;
    for (int i = 0; i < 10; ++i) tot += ph[i];
    printf("; wave 0 of workgroup 0 (last launch of the call), us [share]:");
    for (int i = 0; i < 10; ++i) printf(" %s %.1f [%.0f%%]", names[i], ph[i] * 0.01, 100.0 * ph[i] / tot);
#endif
    printf("\n");
  }
  return 0;
}
