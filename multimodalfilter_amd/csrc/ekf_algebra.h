// Every product-sum below is an EXPLICIT fused multiply-add and implicit contraction is off inside these functions: which
// a * b + c the compiler fuses otherwise depends on the kernel the function is inlined into, and the two callers
// (ekf_step_kernel, the persistent loop) must agree to the bit.
// The d x d algebra of K3 (one trajectory per lane, everything in registers) and the two halves of an EKF step -- predict +
// correct of ONE sub-filter, and the fusion of K sub-filters -- as device functions: ekf_step_kernel (ekf.hip) and the
// persistent EKF loop (ekf_persistent.inc) run the same statements in the same order, so their results are the same bits.
//   torchfilter ExtendedKalmanFilter._predict_step / _update_step with C = I (external dependency)
//   /root/reference/crossmodal/base_models/crossmodal_kf.py:153-167  (crossmodal fusion)
//   /root/reference/crossmodal/base_models/utility.py:4-11           (weighted_average)
//   /root/reference/crossmodal/base_models/unimodal_kf.py:204-242    (information-form fusion)
#pragma once
#include "mmf_common.h"

namespace mmf_ekf {

template <int D>
struct Mat {
  float a[D][D];
};

template <int D>
__device__ __forceinline__ Mat<D> load_mat(const float* p) {
  Mat<D> m;
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) m.a[i][j] = p[i * D + j];
  return m;
}

template <int D>
__device__ __forceinline__ void store_mat(float* p, const Mat<D>& m) {
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) p[i * D + j] = m.a[i][j];
}

template <int D>
__device__ __forceinline__ Mat<D> matmul(const Mat<D>& x, const Mat<D>& y) {
#pragma clang fp contract(off)
  Mat<D> r;
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < D; ++k) s = __builtin_fmaf(x.a[i][k], y.a[k][j], s);
      r.a[i][j] = s;
    }
  return r;
}

template <int D>
__device__ __forceinline__ Mat<D> matmul_nt(const Mat<D>& x, const Mat<D>& y) {  // x y^T
#pragma clang fp contract(off)
  Mat<D> r;
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < D; ++k) s = __builtin_fmaf(x.a[i][k], y.a[j][k], s);
      r.a[i][j] = s;
    }
  return r;
}

// Gauss-Jordan with partial pivoting (the pivot order LU-based torch.inverse uses); row
// swaps are branch-free selects so every index stays compile-time and the matrix stays in
// registers.
template <int D>
__device__ __forceinline__ Mat<D> inverse(Mat<D> m) {
#pragma clang fp contract(off)
  Mat<D> inv;
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) inv.a[i][j] = (i == j) ? 1.f : 0.f;
#pragma unroll
  for (int c = 0; c < D; ++c) {
#pragma unroll
    for (int r = c + 1; r < D; ++r) {
      const bool sw = fabsf(m.a[r][c]) > fabsf(m.a[c][c]);
#pragma unroll
      for (int j = 0; j < D; ++j) {
        const float t0 = m.a[c][j], t1 = m.a[r][j];
        m.a[c][j] = sw ? t1 : t0;
        m.a[r][j] = sw ? t0 : t1;
        const float u0 = inv.a[c][j], u1 = inv.a[r][j];
        inv.a[c][j] = sw ? u1 : u0;
        inv.a[r][j] = sw ? u0 : u1;
      }
    }
    const float piv = 1.0f / m.a[c][c];
#pragma unroll
    for (int j = 0; j < D; ++j) {
      m.a[c][j] *= piv;
      inv.a[c][j] *= piv;
    }
#pragma unroll
    for (int r = 0; r < D; ++r) {
      if (r == c) continue;
      const float f = m.a[r][c];
#pragma unroll
      for (int j = 0; j < D; ++j) {
        m.a[r][j] = __builtin_fmaf(-f, m.a[c][j], m.a[r][j]);
        inv.a[r][j] = __builtin_fmaf(-f, inv.a[c][j], inv.a[r][j]);
      }
    }
  }
  return inv;
}

constexpr int kMaxK = 4;

// predict + correct of one sub-filter (C = I):  S- = A S A^T + L L^T;  K = S- (S- + T T^T)^-1;
// mu = mu- + K (z - mu-);  S = (I - K) S-
template <int D>
__device__ __forceinline__ void predict_correct(const Mat<D>& Ak, const Mat<D>& S0, const Mat<D>& L, const Mat<D>& T,
                                                const float (&mp)[D], const float (&z)[D], float (&mu_out)[D], Mat<D>& S_out) {
#pragma clang fp contract(off)
  const Mat<D> AS = matmul<D>(Ak, S0);
  Mat<D> Sp = matmul_nt<D>(AS, Ak);
  const Mat<D> Q = matmul_nt<D>(L, L);
  const Mat<D> Rm = matmul_nt<D>(T, T);
  Mat<D> Sinn;
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) {
      Sp.a[i][j] += Q.a[i][j];
      Sinn.a[i][j] = Sp.a[i][j] + Rm.a[i][j];
    }
  const Mat<D> G = matmul<D>(Sp, inverse<D>(Sinn));
  float innov[D];
#pragma unroll
  for (int i = 0; i < D; ++i) innov[i] = z[i] - mp[i];
  Mat<D> ImG;
#pragma unroll
  for (int i = 0; i < D; ++i) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < D; ++j) {
      s = __builtin_fmaf(G.a[i][j], innov[j], s);
      ImG.a[i][j] = ((i == j) ? 1.f : 0.f) - G.a[i][j];
    }
    mu_out[i] = mp[i] + s;
  }
  S_out = matmul<D>(ImG, Sp);
}

// fusion of K corrected sub-filters: 1 = crossmodal weighted average (w: (K, D) weights of this trajectory),
// 2 = information form
template <int D>
__device__ __forceinline__ void fuse(int K, int fusion, const float (&w)[kMaxK][D], const float (&mus)[kMaxK][D],
                                     const Mat<D> (&Ss)[kMaxK], float (&mf)[D], Mat<D>& Sf) {
#pragma clang fp contract(off)
  if (fusion == 1) {
    // mu_f = sum_k (w_k / (sum_k w_k + 1e-9)) mu_k ; Sigma_f = sum_k (w_k w_k^T) (.) Sigma_k
    float wsum[D];
#pragma unroll
    for (int i = 0; i < D; ++i) wsum[i] = 0.f;
#pragma unroll
    for (int k = 0; k < kMaxK; ++k) {
      if (k >= K) break;
#pragma unroll
      for (int i = 0; i < D; ++i) wsum[i] += w[k][i];
    }
#pragma unroll
    for (int i = 0; i < D; ++i) {
      mf[i] = 0.f;
#pragma unroll
      for (int j = 0; j < D; ++j) Sf.a[i][j] = 0.f;
    }
#pragma unroll
    for (int k = 0; k < kMaxK; ++k) {
      if (k >= K) break;
#pragma unroll
      for (int i = 0; i < D; ++i) {
        mf[i] = __builtin_fmaf(w[k][i] / (wsum[i] + 1e-9f), mus[k][i], mf[i]);
#pragma unroll
        for (int j = 0; j < D; ++j) Sf.a[i][j] = __builtin_fmaf(w[k][i] * w[k][j], Ss[k].a[i][j], Sf.a[i][j]);
      }
    }
  } else if (fusion == 2) {
    // P_k = (Sigma_k + 1e-9)^-1 ; Sigma_f = (sum_k P_k + 1e-9)^-1 ; mu_f = Sigma_f sum_k P_k mu_k
    Mat<D> Psum;
    float info[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
      info[i] = 0.f;
#pragma unroll
      for (int j = 0; j < D; ++j) Psum.a[i][j] = 0.f;
    }
#pragma unroll
    for (int k = 0; k < kMaxK; ++k) {
      if (k >= K) break;
      Mat<D> t = Ss[k];
#pragma unroll
      for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) t.a[i][j] += 1e-9f;
      const Mat<D> P = inverse<D>(t);
#pragma unroll
      for (int i = 0; i < D; ++i) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < D; ++j) {
          s = __builtin_fmaf(P.a[i][j], mus[k][j], s);
          Psum.a[i][j] += P.a[i][j];
        }
        info[i] += s;
      }
    }
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j < D; ++j) Psum.a[i][j] += 1e-9f;
    Sf = inverse<D>(Psum);
#pragma unroll
    for (int i = 0; i < D; ++i) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < D; ++j) s = __builtin_fmaf(Sf.a[i][j], info[j], s);
      mf[i] = s;
    }
  }
}

}  // namespace mmf_ekf
