// When do the workgroups of the resident K4 kernel (image_encoder_resident.inc) start and finish, by XCD?  Includes the product
// source with MMF_K4_WG_STAMPS: thread 0 of every workgroup stamps wall_clock64 (100 MHz) at entry and exit and its XCC_ID.
// The launches are persistent grids with the same number of images per workgroup; a spread of finishing times is what a
// static partition leaves on the table.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Iinclude -Imultimodalfilter_amd/csrc \
//         -o scripts/ubench/k4_wg_spread scripts/ubench/k4_wg_spread.hip && ./scripts/ubench/k4_wg_spread [images] [nets] [unused] [brief]
#define MMF_K4_WG_STAMPS 1
#include "../../multimodalfilter_amd/csrc/image_encoder.hip"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

static float* dev_random(size_t n, float scale) {
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = scale * (2.f * rand() / RAND_MAX - 1.f);
  float* d;
  (void)hipMalloc(&d, n * 4);
  (void)hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  return d;
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 4096, nets = argc > 2 ? atoi(argv[2]) : 2;
  const bool brief = argc > 4;
  MmfImageEncoderDesc d{};
  const size_t cw[5] = {32 * 25, 32 * 32 * 9, 32 * 32 * 9, 16 * 32 * 9, 8 * 16 * 9};
  const size_t cb[5] = {32, 32, 32, 16, 8};
  for (int i = 0; i < 5; ++i) { d.conv_w[i] = dev_random(cw[i], 0.06f); d.conv_b[i] = dev_random(cb[i], 0.05f); }
  d.fc_w = dev_random(64 * 8192, 0.01f); d.fc_b = dev_random(64, 0.05f);
  for (int i = 0; i < 2; ++i) { d.res_w[i] = dev_random(64 * 64, 0.1f); d.res_b[i] = dev_random(64, 0.05f); }
  float* blobs[4];
  for (int k = 0; k < nets; ++k) {
    (void)hipMalloc(&blobs[k], mmf_image_encoder_floats() * 4);
    if (mmf_pack_image_encoder(&d, blobs[k], nullptr)) return 1;
  }
  float* images = dev_random(size_t(N) * 1024, 1.f);
  float* feat; (void)hipMalloc(&feat, size_t(nets) * N * 64 * 4);
  void* ws; (void)hipMalloc(&ws, mmf_image_encoder_workspace_bytes(N, nets));
  for (int rep = 0; rep < 4; ++rep) {
    if (mmf_image_encoder(blobs, nets, images, feat, ws, nullptr, MMF_PREC_F16X3, MMF_ENCODER_DEFAULT, N, nullptr)) return 2;
    (void)hipDeviceSynchronize();
    if (rep < 2) continue;
    static long long st[2][2][2048];
    static int xcc[2][2048];
    (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(g_wg_stamp), sizeof(st));
    (void)hipMemcpyFromSymbol(xcc, HIP_SYMBOL(g_wg_xcc), sizeof(xcc));
    const char* names[1] = {"image_encoder_resident"};
    const int wgs[1] = {std::min(256 / nets, N) * nets};
    for (int k = 0; k < 1; ++k) {
      const int n = wgs[k];
      long long t0 = st[k][0][0], t1 = 0;
      for (int w = 0; w < n; ++w) { t0 = std::min(t0, st[k][0][w]); t1 = std::max(t1, st[k][1][w]); }
      const double dur = (t1 - t0) * 0.01;
      printf("rep %d %s: %d workgroups, first start -> last end %.1f us\n", rep, names[k], n, dur);
      if (brief) continue;
      if (rep == 3) {  // which workgroups are the late ones?  By dispatch half, by parity, and a sample of XCD 0's in id order
        double half[2] = {0, 0}, par[2] = {0, 0};
        for (int w = 0; w < n; ++w) {
          half[w >= n / 2] += (st[k][1][w] - t0) * 0.01 / (n / 2);
          par[w & 1] += (st[k][1][w] - t0) * 0.01 / (n / 2);
        }
        printf("   mean end: first half of the ids %.1f us, second half %.1f; even ids %.1f, odd ids %.1f\n", half[0], half[1], par[0], par[1]);
        printf("   XCD 0 ends in id order:");
        for (int w = 0; w < n; w += 8) printf(" %.0f", (st[k][1][w] - t0) * 0.01);
        printf("\n");
      }
      for (int x = 0; x < 8; ++x) {
        long long lo = 1LL << 62, hi = 0, slo = 1LL << 62, shi = 0;
        int cnt = 0;
        for (int w = 0; w < n; ++w)
          if (xcc[k][w] == x) {
            lo = std::min(lo, st[k][1][w]); hi = std::max(hi, st[k][1][w]);
            slo = std::min(slo, st[k][0][w]); shi = std::max(shi, st[k][0][w]);
            ++cnt;
          }
        if (cnt) printf("   XCD %d: %3d workgroups start %6.1f .. %6.1f us, end %6.1f .. %6.1f us (idle after its last: %4.1f %%)\n", x, cnt,
                        (slo - t0) * 0.01, (shi - t0) * 0.01, (lo - t0) * 0.01, (hi - t0) * 0.01, 100.0 * (t1 - hi) / (t1 - t0));
      }
    }
  }
  return 0;
}
