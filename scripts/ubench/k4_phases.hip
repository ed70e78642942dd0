// Where does a band of K4's f16x3 convolutions spend its time?  Includes the product kernel
// source with MMF_K4_PHASE_CLOCKS defined: workgroup (0,0)'s thread 0 accumulates wall_clock64
// (100 MHz) between phase boundaries, per 3x3 layer.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Imultimodalfilter_amd/csrc \
//         -o scripts/ubench/k4_phases scripts/ubench/k4_phases.hip
#define MMF_K4_PHASE_CLOCKS 1
#include "../../multimodalfilter_amd/csrc/image_encoder.hip"

#include <cstdio>
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
  const int N = argc > 1 ? atoi(argv[1]) : 2048, nets = 2;
  MmfImageEncoderDesc d{};
  const size_t cw[5] = {32 * 25, 32 * 32 * 9, 32 * 32 * 9, 16 * 32 * 9, 8 * 16 * 9};
  const size_t cb[5] = {32, 32, 32, 16, 8};
  for (int i = 0; i < 5; ++i) { d.conv_w[i] = dev_random(cw[i], 0.06f); d.conv_b[i] = dev_random(cb[i], 0.05f); }
  d.fc_w = dev_random(64 * 8192, 0.01f); d.fc_b = dev_random(64, 0.05f);
  for (int i = 0; i < 2; ++i) { d.res_w[i] = dev_random(64 * 64, 0.1f); d.res_b[i] = dev_random(64, 0.05f); }
  float* blobs[2];
  for (int k = 0; k < nets; ++k) {
    hipMalloc(&blobs[k], mmf_image_encoder_floats() * 4);
    if (mmf_pack_image_encoder(&d, blobs[k], nullptr)) return 1;
  }
  float* images = dev_random(size_t(N) * 1024, 1.f);
  float* feat; hipMalloc(&feat, size_t(nets) * N * 64 * 4);
  void* ws; hipMalloc(&ws, mmf_image_encoder_workspace_bytes(N, nets));
  for (int rep = 0; rep < 3; ++rep) {
    long long zero[4][8] = {};
    hipMemcpyToSymbol(HIP_SYMBOL(g_phase), zero, sizeof(zero));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    if (mmf_image_encoder(blobs, nets, images, feat, ws, nullptr, MMF_PREC_F16X3, MMF_ENCODER_DEFAULT, N, nullptr)) return 2;
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long ph[4][8];
    hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_phase), sizeof(ph));
    printf("rep %d: %d images x %d nets in %.3f ms\n", rep, N, nets, ms);
    const char* names[4] = {"32->32", "32->32+skip", "32->16", "16->8"};
    const int bands = 2 * N / (256 / nets);  // 16-row units per CU (two 8-row bands each, one per resident workgroup)
    for (int l = 0; l < 4; ++l)
      printf("  %-12s us/band: wait-prev %.2f commit %.2f barrier %.2f prefetch-issue %.2f mfma %.2f epilogue %.2f  (x%d bands)\n",
             names[l], ph[l][0] * 0.01 / bands, ph[l][1] * 0.01 / bands, ph[l][2] * 0.01 / bands,
             ph[l][3] * 0.01 / bands, ph[l][4] * 0.01 / bands, ph[l][5] * 0.01 / bands, bands);
  }
  return 0;
}
