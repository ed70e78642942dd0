// What does v_mfma_f32_32x32x16_f16 compute, bit for bit?  D = C + sum_{k<16} A[i][k] B[k][j] with f16 inputs and an
// fp32 accumulator -- but in which order and with how many roundings?  Candidates evaluated on the host in exact
// arithmetic (products of two f16 are exact in fp32; sums are carried in long double / __float128-free integer-free
// form: double is exact for up to 16 products whose exponents lie within ~2^29 of each other, which the generated
// data respects in the "narrow" case and violates on purpose in the "wide" case):
//   seq      : fmaf chain over k = 0..15, one rounding per product
//   exact16  : round_f32(C + exact sum of the 16 products)                 (one rounding per instruction)
//   exact8x2 : round_f32(round_f32(C + exact sum of k = 0..7) + exact sum of k = 8..15)   (one per half: lanes 0-31 / 32-63)
//   exact4x4 : four groups of four
// A strict CPU twin of the f16x3 mode needs one of these to hold for EVERY output.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <random>
#include <vector>

using half8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;

__global__ void k(const _Float16* A, const _Float16* B, const float* C, float* D) {
  // A: [32 rows][16 k], B: [16 k][32 cols], C/D: [32][32]
  const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
  half8 a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = A[i * 16 + 8 * h + e];
    b[e] = B[(8 * h + e) * 32 + i];
  }
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = C[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i];
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i] = c[r];
}

int main() {
  std::mt19937 gen(7);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::uniform_int_distribution<int> ex(-12, 12);
  for (int wide = 0; wide < 3; ++wide) {
    long n = 0, m_seq = 0, m_e16 = 0, m_e8 = 0, m_e4 = 0, m_e16_then_c = 0, m_e2 = 0;
    for (int rep = 0; rep < 200; ++rep) {
      std::vector<_Float16> A(32 * 16), B(16 * 32);
      std::vector<float> C(32 * 32), D(32 * 32);
      for (auto& v : A) v = static_cast<_Float16>(nd(gen) * (wide ? std::ldexp(1.f, ex(gen)) : 1.f));
      for (auto& v : B) v = static_cast<_Float16>(nd(gen) * (wide == 2 ? std::ldexp(1.f, ex(gen) / 2) : 1.f));
      for (auto& v : C) v = nd(gen) * (wide ? std::ldexp(1.f, ex(gen)) : 1.f);
      _Float16 *dA, *dB; float *dC, *dD;
      (void)hipMalloc(&dA, A.size() * 2); (void)hipMalloc(&dB, B.size() * 2); (void)hipMalloc(&dC, C.size() * 4); (void)hipMalloc(&dD, D.size() * 4);
      (void)hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice);
      (void)hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
      (void)hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
      k<<<1, 64>>>(dA, dB, dC, dD);
      (void)hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
      (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dC); (void)hipFree(dD);
      for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
          double p[16];
          for (int kk = 0; kk < 16; ++kk) p[kk] = static_cast<double>(static_cast<float>(A[i * 16 + kk])) * static_cast<double>(static_cast<float>(B[kk * 32 + j]));
          const float c = C[i * 32 + j], got = D[i * 32 + j];
          float seq = c;
          for (int kk = 0; kk < 16; ++kk) seq = std::fmaf(static_cast<float>(A[i * 16 + kk]), static_cast<float>(B[kk * 32 + j]), seq);
          long double s16 = 0; for (int kk = 0; kk < 16; ++kk) s16 += p[kk];
          const float e16 = static_cast<float>(static_cast<long double>(c) + s16);
          const float e16c = static_cast<float>(static_cast<double>(static_cast<float>(s16)) + c);  // sum rounded first, then + C
          long double s0 = 0, s1 = 0; for (int kk = 0; kk < 8; ++kk) { s0 += p[kk]; s1 += p[8 + kk]; }
          const float e8 = static_cast<float>(static_cast<long double>(static_cast<float>(static_cast<long double>(c) + s0)) + s1);
          float e4 = c;
          for (int g4 = 0; g4 < 4; ++g4) { long double s = 0; for (int kk = 0; kk < 4; ++kk) s += p[4 * g4 + kk]; e4 = static_cast<float>(static_cast<long double>(e4) + s); }
          float e2 = c;
          for (int g2 = 0; g2 < 8; ++g2) e2 = static_cast<float>(static_cast<long double>(e2) + (static_cast<long double>(p[2 * g2]) + p[2 * g2 + 1]));
          m_e2 += std::memcmp(&got, &e2, 4) == 0;
          ++n;
          m_seq += std::memcmp(&got, &seq, 4) == 0;
          m_e16 += std::memcmp(&got, &e16, 4) == 0;
          m_e8 += std::memcmp(&got, &e8, 4) == 0;
          m_e4 += std::memcmp(&got, &e4, 4) == 0;
          m_e16_then_c += std::memcmp(&got, &e16c, 4) == 0;
        }
    }
    printf("%s: outputs %ld | bit-equal to: fmaf chain %.4f, exact16+C one rounding %.4f, exact8 x 2 %.4f, exact4 x 4 %.4f, round(sum16) + C %.4f\n",
           wide == 0 ? "narrow exponents" : (wide == 1 ? "wide exponents (A, C)" : "wide exponents (A, B, C)"), n, 1.0 * m_seq / n, 1.0 * m_e16 / n, 1.0 * m_e8 / n, 1.0 * m_e4 / n, 1.0 * m_e16_then_c / n);
    printf("   exact pairs x 8: %.4f\n", 1.0 * m_e2 / n);
  }
  return 0;
}
