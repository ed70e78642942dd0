// Where do the fused K4 kernels (image_encoder_fused.inc) spend their time?  Includes the product
// source with MMF_K4_PHASE_CLOCKS: thread 0 of workgroup (0,0) accumulates wall_clock64 (100 MHz)
// between phase boundaries.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Imultimodalfilter_amd/csrc \
//         -o scripts/ubench/k4_fused_phases scripts/ubench/k4_fused_phases.hip
#ifndef NO_PHASE_CLOCKS  // -DNO_PHASE_CLOCKS: kernel times only (the clocks serialise what they measure)
#define MMF_K4_PHASE_CLOCKS 1
#endif
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
  {  // per-kernel times (no clocks involved): the launchers of the fused path, one by one
    const size_t act = size_t(nets) * N * 32 * kImg * kImg;
    float* bufA = static_cast<float*>(ws); float* bufB = bufA + act; float* bufC = bufB + act;
    FusedArgs fa{}; Conv4Args c4{};
    for (int i = 0; i < nets; ++i) { fa.packed[i] = blobs[i]; c4.packed[i] = blobs[i]; }
    fa.images = images; fa.N = N; c4.N = N;
    const bool bf = getenv("BF16") != nullptr;
    hipEvent_t ev[4]; for (auto& e : ev) hipEventCreate(&e);
    float tA = 0, tB = 0, tC = 0; const int reps = 5;
    for (int rep = 0; rep < reps + 2; ++rep) {
      hipEventRecord(ev[0]);
      fa.out = bufA; launch_fused(fa, nets, 0, bf, fused_flavour(), nullptr);
      hipEventRecord(ev[1]);
      fa.bin = bufA; fa.out = bufB; launch_fused(fa, nets, 1, bf, fused_flavour(), nullptr);
      hipEventRecord(ev[2]);
      c4.din = reinterpret_cast<const unsigned char*>(bufB); c4.out = bufC; launch_conv4(c4, nets, nullptr);
      hipEventRecord(ev[3]); hipDeviceSynchronize();
      float a, b, c; hipEventElapsedTime(&a, ev[0], ev[1]); hipEventElapsedTime(&b, ev[1], ev[2]); hipEventElapsedTime(&c, ev[2], ev[3]);
      if (rep >= 2) { tA += a; tB += b; tC += c; }
    }
    printf("kernel times (us): stem_conv2a %.1f  conv2b_conv3 %.1f  conv4 %.1f\n", 1e3 * tA / reps, 1e3 * tB / reps, 1e3 * tC / reps);
  }
#ifdef MMF_K4_PHASE_CLOCKS
  if (getenv("PHASES"))
  for (int rep = 0; rep < 3; ++rep) {
    long long zero[4][8] = {};
    hipMemcpyToSymbol(HIP_SYMBOL(g_fphase), zero, sizeof(zero));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    if (mmf_image_encoder(blobs, nets, images, feat, ws, nullptr, MMF_PREC_F16X3, MMF_ENCODER_DEFAULT, N, nullptr)) return 2;
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long ph[4][8];
    hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_fphase), sizeof(ph));
    printf("rep %d: %d images x %d nets in %.3f ms\n", rep, N, nets, ms);
    const int imgs = (N + 256 / nets - 1) / (256 / nets);  // images per workgroup
    printf("  stem_conv2a  us/image (x%d): wait-top %.2f commit+barrier %.2f stem %.2f barrier %.2f conv %.2f store %.2f\n", imgs,
           ph[0][0] * 0.01 / imgs, ph[0][1] * 0.01 / imgs, ph[0][2] * 0.01 / imgs, ph[0][3] * 0.01 / imgs,
           ph[0][4] * 0.01 / imgs, ph[0][5] * 0.01 / imgs);
    printf("  conv2b_conv3 X wave us/image (x%d): setup %.2f conv2b+gathers %.2f stem+writeC %.2f barrier %.2f\n", imgs,
           ph[1][0] * 0.01 / imgs, ph[1][1] * 0.01 / imgs, ph[1][2] * 0.01 / imgs, ph[1][3] * 0.01 / imgs);
    printf("  conv2b_conv3 Y wave us/image (x%d): stores %.2f conv3 %.2f commit+prefetch+image %.2f barrier %.2f\n", imgs,
           ph[3][0] * 0.01 / imgs, ph[3][1] * 0.01 / imgs, ph[3][2] * 0.01 / imgs, ph[3][3] * 0.01 / imgs);
  }
#endif
  return 0;
}
