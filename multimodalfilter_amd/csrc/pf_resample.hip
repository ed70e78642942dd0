// K1: particle reweight + normalise + weighted-mean estimate + fixed-point CDF + systematic /
// multinomial resample + gather, one kernel, one workgroup per trajectory.
//
// Replaces the post-measurement half of torchfilter's ParticleFilter.forward (external
// dependency of the reference; call sites /root/reference/crossmodal/eval_helpers.py:139-142;
// SURVEY.md 3.2 / T1).  The resampling definition is the integer one of
// oracle/resample.py (weights quantised to 2^-24 of the row max, u64 inclusive CDF), so the
// ancestor indices do not depend on the scan tree and match the oracle bit for bit.
//
// HBM traffic per particle (algorithmic): loglik 4 + logw 4 + states 4d read, states 4d written
// (+4 logw written in mode 0) = 32 B (d=3) / 24 B (d=2); everything else stays in LDS:
//   slot[i] (8 B/particle): first the fp32 unnormalised log-weight x_i, then (modes 1/2)
//   overwritten in place by the u64 inclusive CDF.
#include <cmath>

#include "mmf_common.h"
#include "../../include/mmf_detmath.h"

// Phase stamps for scripts/ubench/k1_phases.hip (compiled out of the library): thread 0 of every
// workgroup stores s_memtime at phase boundaries into a buffer of its own.
#ifdef MMF_K1_PHASE_CLOCKS
__device__ unsigned long long g_k1_stamps[1024][8];
#define K1_STAMP(i)                                                                          \
  do {                                                                                       \
    if (threadIdx.x == 0 && blockIdx.x < 1024) {                                             \
      unsigned long long t_;                                                                 \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");             \
      g_k1_stamps[blockIdx.x][i] = t_;                                                       \
    }                                                                                        \
  } while (0)
#else
#define K1_STAMP(i)
#endif

#include "pf_resample_systematic.inc"

namespace {
#include "pf_resample_cluster.inc"
int g_cluster_enabled = getenv("MMF_K1_CLUSTER") && getenv("MMF_K1_CLUSTER")[0] == '1';  // off: see mmf_pf_set_resample_cluster

constexpr int kBlock = mmf::kK1Block;  // 16 waves; M = 4096 -> one float4 chunk per thread
constexpr int kMaxWaves = mmf::kK1MaxWaves;
constexpr int kFixBits = mmf::kK1FixBits;
using Scratch = mmf::K1Scratch;  // lives behind the slots in dynamic LDS

// STAGE: the particle states are kept in LDS next to the CDF (M * 4D more bytes), so the gather
// of the resampling step reads LDS instead of going back to L2 / HBM for rows this workgroup
// has just streamed through.
// SOFT (torchfilter's ``soft_resample_alpha`` < 1): ancestors are drawn from the mixture
// alpha * w_i + (1 - alpha) / M, in fixed point q'_i = ((A q_i << 8) + (2^24 - A) floor((Q << 8) / M)) >> 32
// with A = floor(alpha 2^24) (so q'_i <= 2^24 and every bound of the integer scheme still holds), and the
// survivors carry the importance weights w / mixture, normalised (oracle/resample.py).
template <int D, bool STAGE, bool SOFT = false>
__global__ __launch_bounds__(kBlock) void pf_reweight_resample_kernel(
    const float* __restrict__ loglik, const float* __restrict__ logw_in,
    const float* __restrict__ states_in, const float* __restrict__ u,
    float* __restrict__ estimate, float* states_out, float* logw_out,
    int32_t* __restrict__ indices_out, int M, int M_out, int mode, float alpha, float lw_uniform,
    float log_uniform) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const bool need_cdf = mode != 0;
  const int slot_bytes = need_cdf ? 8 : 4;
  const size_t slots_sz = (static_cast<size_t>(M) * slot_bytes + 15) & ~static_cast<size_t>(15);
  float* xf = reinterpret_cast<float*>(smem);                            // mode 0: x_i at [i]
  unsigned long long* cdf = reinterpret_cast<unsigned long long*>(smem);  // modes 1/2
  Scratch& sc = *reinterpret_cast<Scratch*>(smem + slots_sz);
  float* xs_lds = reinterpret_cast<float*>(smem + slots_sz + ((sizeof(Scratch) + 15) & ~static_cast<size_t>(15)));  // STAGE

  const int n = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & (MMF_WAVE - 1);
  const int wave = tid >> 6;
  const int nwaves = blockDim.x >> 6;
  const int chunk = blockDim.x * 4;

  const float* ll = loglik + static_cast<size_t>(n) * M;
  // logw_in == nullptr: uniform weights -log M (what every resampling step leaves behind): nothing to read
  const float* lw = logw_in ? logw_in + static_cast<size_t>(n) * M : nullptr;
  // lw_uniform = float(-log(double(M))), rounded once on the host: the value mmf_pf_init_particles and
  // every resampling step write, and the one the oracle holds (device logf may differ by an ulp)
  const float* xs = states_in + static_cast<size_t>(n) * M * D;
  const bool vec = (M & 3) == 0;  // rows 16-B aligned -> float4 path

  auto x_store = [&](int i, float v) {
    if (need_cdf) reinterpret_cast<float*>(cdf + i)[0] = v; else xf[i] = v;
  };
  auto x_load = [&](int i) -> float {
    return need_cdf ? reinterpret_cast<const float*>(cdf + i)[0] : xf[i];
  };

  K1_STAMP(0);
  // the first chunk's particle states are requested before anything waits on memory: their
  // latency overlaps pass 1 (log-weights, row maximum, two barriers)
  float st0[4 * D];
  {
    const int i0 = tid * 4;
    if (vec && i0 + 3 < M) {
      const float4* p = reinterpret_cast<const float4*>(xs + static_cast<size_t>(i0) * D);
#pragma unroll
      for (int k = 0; k < D; ++k) {
        const float4 t = p[k];
        st0[4 * k] = t.x; st0[4 * k + 1] = t.y; st0[4 * k + 2] = t.z; st0[4 * k + 3] = t.w;
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4 * D; ++k)
        st0[k] = (i0 * D + k < M * D) ? xs[static_cast<size_t>(i0) * D + k] : 0.f;
    }
  }

  // ---- pass 1: x_i = logw_i + loglik_i -> LDS, row max
  float mx = -INFINITY;
  for (int base = 0; base < M; base += chunk) {
    const int i0 = base + tid * 4;
    float v[4];
    if (vec && i0 + 3 < M) {
      const float4 a = *reinterpret_cast<const float4*>(ll + i0);
      const float4 b = lw ? *reinterpret_cast<const float4*>(lw + i0) : make_float4(lw_uniform, lw_uniform, lw_uniform, lw_uniform);
      v[0] = b.x + a.x; v[1] = b.y + a.y; v[2] = b.z + a.z; v[3] = b.w + a.w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = (i0 + j < M) ? (lw ? lw[i0 + j] : lw_uniform) + ll[i0 + j] : -INFINITY;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (i0 + j < M) { x_store(i0 + j, v[j]); mx = fmaxf(mx, v[j]); }
  }
  mx = mmf::wave_max(mx);
  if (lane == 0) sc.red[wave][0] = mx;
  __syncthreads();
  mx = sc.red[0][0];
  for (int w = 1; w < nwaves; ++w) mx = fmaxf(mx, sc.red[w][0]);
  __syncthreads();
  K1_STAMP(1);

  // ---- pass 2: e_i = detexp(x_i - max); float sums for the estimate; integer CDF
  float S = 0.f, acc[D];
#pragma unroll
  for (int c = 0; c < D; ++c) acc[c] = 0.f;
  unsigned long long carry = 0, qsum = 0;
  for (int base = 0; base < M; base += chunk) {
    const int i0 = base + tid * 4;
    float e[4];
    unsigned long long q[4], tsum = 0;
    float st[4 * D];
    if (base == 0) {
#pragma unroll
      for (int k = 0; k < 4 * D; ++k) st[k] = st0[k];
    } else if (vec && i0 + 3 < M) {
      const float4* p = reinterpret_cast<const float4*>(xs + static_cast<size_t>(i0) * D);
#pragma unroll
      for (int k = 0; k < D; ++k) {
        const float4 t = p[k];
        st[4 * k] = t.x; st[4 * k + 1] = t.y; st[4 * k + 2] = t.z; st[4 * k + 3] = t.w;
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4 * D; ++k)
        st[k] = (i0 * D + k < M * D) ? xs[static_cast<size_t>(i0) * D + k] : 0.f;
    }
    if (STAGE && need_cdf) {
#pragma unroll
      for (int k = 0; k < 4 * D; ++k)
        if (i0 * D + k < M * D) xs_lds[i0 * D + k] = st[k];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool ok = i0 + j < M;
      e[j] = ok ? mmf::detexp(x_load(i0 + j) - mx) : 0.f;
      q[j] = static_cast<unsigned long long>(floorf(e[j] * 16777216.0f));
      tsum += q[j];
      S += e[j];
#pragma unroll
      for (int c = 0; c < D; ++c) acc[c] = __builtin_fmaf(e[j], st[j * D + c], acc[c]);  // explicit: oracle/strict restates this chain
    }
    qsum += tsum;
    if (need_cdf && !SOFT) {
      const unsigned long long incl = mmf::wave_inclusive_scan(tsum, lane);
      if (lane == MMF_WAVE - 1) sc.wave_tot[wave] = incl;
      // the float sums of the estimate ride on the last chunk's two barriers instead of two of their own
      const bool last_chunk = base + chunk >= M;
      if (last_chunk) {
        const float Sw = mmf::wave_sum(S);
        float aw[D];
#pragma unroll
        for (int c = 0; c < D; ++c) aw[c] = mmf::wave_sum(acc[c]);
        if (lane == 0) {
          sc.red[wave][0] = Sw;
#pragma unroll
          for (int c = 0; c < D; ++c) sc.red[wave][1 + c] = aw[c];
        }
      }
      __syncthreads();
      if (last_chunk && tid <= D) {
        float t = 0.f;
        for (int w = 0; w < nwaves; ++w) t += sc.red[w][tid];
        sc.bcast[tid] = t;
      }
      unsigned long long before = carry, total = 0;
      for (int w = 0; w < nwaves; ++w) {
        const unsigned long long t = sc.wave_tot[w];
        if (w < wave) before += t;
        total += t;
      }
      unsigned long long run = before + incl - tsum;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        run += q[j];
        if (i0 + j < M) cdf[i0 + j] = run;
      }
      carry += total;
      __syncthreads();
    }
  }
  if (SOFT && need_cdf) {
    // Q first (the mixture needs it), then a second sweep: e and q again from the staged x_i (each
    // thread re-reads only the slots it overwrites), mixture weights, inclusive scan -> CDF
    const unsigned long long incl = mmf::wave_inclusive_scan(qsum, lane);
    if (lane == MMF_WAVE - 1) sc.wave_tot[wave] = incl;
    __syncthreads();
    unsigned long long Qall = 0;
    for (int w = 0; w < nwaves; ++w) Qall += sc.wave_tot[w];
    __syncthreads();
    const unsigned long long A = static_cast<unsigned long long>(floorf(alpha * 16777216.0f));
    const unsigned long long uni = (16777216ull - A) * ((Qall << 8) / static_cast<unsigned long long>(M));
    for (int base = 0; base < M; base += chunk) {
      const int i0 = base + tid * 4;
      unsigned long long q[4], tsum = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        q[j] = 0;
        if (i0 + j < M) {
          const float e = mmf::detexp(x_load(i0 + j) - mx);
          q[j] = (((A * static_cast<unsigned long long>(floorf(e * 16777216.0f))) << 8) + uni) >> (kFixBits + 8);
        }
        tsum += q[j];
      }
      const unsigned long long incl2 = mmf::wave_inclusive_scan(tsum, lane);
      if (lane == MMF_WAVE - 1) sc.wave_tot[wave] = incl2;
      __syncthreads();
      unsigned long long before = carry, total = 0;
      for (int w = 0; w < nwaves; ++w) {
        const unsigned long long t = sc.wave_tot[w];
        if (w < wave) before += t;
        total += t;
      }
      unsigned long long run = before + incl2 - tsum;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        run += q[j];
        if (i0 + j < M) cdf[i0 + j] = run;
      }
      carry += total;
      __syncthreads();
    }
  }
  if (!need_cdf || SOFT) {  // (plain resampling did this on its last chunk's barriers)
    S = mmf::wave_sum(S);
#pragma unroll
    for (int c = 0; c < D; ++c) acc[c] = mmf::wave_sum(acc[c]);
    if (lane == 0) {
      sc.red[wave][0] = S;
#pragma unroll
      for (int c = 0; c < D; ++c) sc.red[wave][1 + c] = acc[c];
    }
    __syncthreads();
    if (tid <= D) {
      float t = 0.f;
      for (int w = 0; w < nwaves; ++w) t += sc.red[w][tid];
      sc.bcast[tid] = t;
    }
    __syncthreads();
  }
  K1_STAMP(2);
  S = sc.bcast[0];
  if (tid < D) estimate[static_cast<size_t>(n) * D + tid] = sc.bcast[1 + tid] / S;

  // ---- mode 0: normalised log-weights out, particles stay
  if (!need_cdf) {
    const float logS = logf(S);
    float* lo = logw_out + static_cast<size_t>(n) * M;
    float* so = states_out ? states_out + static_cast<size_t>(n) * M * D : nullptr;
    const bool copy = so != nullptr && so != xs;
    for (int i = tid; i < M; i += blockDim.x) {
      lo[i] = (xf[i] - mx) - logS;
      if (copy) {
#pragma unroll
        for (int c = 0; c < D; ++c) so[static_cast<size_t>(i) * D + c] = xs[static_cast<size_t>(i) * D + c];
      }
    }
    return;
  }

  // ---- modes 1/2: positions -> ancestor index by upper_bound in the LDS CDF -> gather
  const unsigned long long Q = cdf[M - 1];
  unsigned long long R = 0;
  if (mode == 1) {
    const unsigned long long U = static_cast<unsigned long long>(floorf(u[n] * 16777216.0f));
    R = (U * Q) >> kFixBits;
  }
  const bool pow2 = (M_out & (M_out - 1)) == 0;
  const int shift = __ffs(M_out) - 1;
  float* so = states_out + static_cast<size_t>(n) * M_out * D;
  // logw_out == nullptr (plain resampling only): the survivors' weights are -log M_out by definition
  float* lo = logw_out ? logw_out + static_cast<size_t>(n) * M_out : nullptr;
  int32_t* io = indices_out ? indices_out + static_cast<size_t>(n) * M_out : nullptr;
  const float* un = (mode == 2) ? u + static_cast<size_t>(n) * M_out : nullptr;
  const bool vec_out = (M_out & 3) == 0;
  const float mix_uniform = (1.0f - alpha) * S / static_cast<float>(M);
  float rsum = 0.f;

  for (int base = 0; base < M_out; base += chunk) {
    const int k0 = base + tid * 4;
    int idx[4];
    float g[4 * D];
    float lr[4] = {log_uniform, log_uniform, log_uniform, log_uniform};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k0 + j;
      idx[j] = 0;
      if (k < M_out) {
        unsigned long long p;
        if (mode == 1) {
          const unsigned long long num = static_cast<unsigned long long>(k) * Q + R;
          p = pow2 ? (num >> shift) : (num / static_cast<unsigned long long>(M_out));
        } else {
          const unsigned long long U = static_cast<unsigned long long>(floorf(un[k] * 16777216.0f));
          p = (U * Q) >> kFixBits;
        }
        int lo_i = 0, hi_i = M;
        if (mode == 1 && j > 0) {
          // systematic positions are sorted and evenly spaced: the ancestor of output k sits at or
          // just above that of output k - 1 -- gallop from there (2-3 LDS reads on average
          // instead of log2 M) before the bisection
          lo_i = idx[j - 1];
          int step = 1;
          while (lo_i + step <= M && cdf[lo_i + step - 1] <= p) { lo_i += step; step <<= 1; }
          hi_i = min(M, lo_i + step - 1);
        }
        while (lo_i < hi_i) {
          const int mid = (lo_i + hi_i) >> 1;
          if (cdf[mid] <= p) lo_i = mid + 1; else hi_i = mid;
        }
        idx[j] = lo_i;
#pragma unroll
        for (int c = 0; c < D; ++c) g[j * D + c] = STAGE ? xs_lds[lo_i * D + c] : xs[static_cast<size_t>(lo_i) * D + c];
        if (SOFT) {  // importance weight of the survivor: w / (alpha w + (1 - alpha) / M), up to the common 1 / S
          const float e = mmf::detexp(((lw ? lw[lo_i] : lw_uniform) + ll[lo_i]) - mx);
          const float r = e / (alpha * e + mix_uniform);
          rsum += r;
          lr[j] = logf(r);
        }
      }
    }
    if (vec_out && k0 + 3 < M_out) {
      float4* p = reinterpret_cast<float4*>(so + static_cast<size_t>(k0) * D);
#pragma unroll
      for (int k = 0; k < D; ++k) p[k] = make_float4(g[4 * k], g[4 * k + 1], g[4 * k + 2], g[4 * k + 3]);
      if (lo) *reinterpret_cast<float4*>(lo + k0) = make_float4(lr[0], lr[1], lr[2], lr[3]);
      if (io) *reinterpret_cast<int4*>(io + k0) = make_int4(idx[0], idx[1], idx[2], idx[3]);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = k0 + j;
        if (k < M_out) {
#pragma unroll
          for (int c = 0; c < D; ++c) so[static_cast<size_t>(k) * D + c] = g[j * D + c];
          if (lo) lo[k] = lr[j];
          if (io) io[k] = idx[j];
        }
      }
    }
  }
  K1_STAMP(3);
  if (SOFT) {  // normalise the survivors' weights (every thread revisits the outputs it wrote)
    rsum = mmf::wave_sum(rsum);
    __syncthreads();
    if (lane == 0) sc.red[wave][0] = rsum;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < nwaves; ++w) t += sc.red[w][0];
    const float log_r = logf(t);
    for (int base = 0; base < M_out; base += chunk)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = base + tid * 4 + j;
        if (k < M_out) lo[k] -= log_r;
      }
  }
}


// ---- plain systematic resampling WITHOUT a search (the bench's and the reference's evaluation mode): the body is
// mmf::resample_systematic_trajectory (pf_resample_systematic.inc, shared with the persistent small-problem loop)
template <int D, bool STAGE>
__global__ __launch_bounds__(kBlock) void pf_resample_systematic_kernel(
    const float* __restrict__ loglik, const float* __restrict__ logw_in, const float* __restrict__ states_in,
    const float* __restrict__ u, float* __restrict__ estimate, float* states_out, float* logw_out,
    int32_t* __restrict__ indices_out, int M, int M_out, float lw_uniform, float log_uniform) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int n = blockIdx.x;
  mmf::K1Trajectory a{};
  a.ll[0] = loglik + static_cast<size_t>(n) * M;
  a.n_ll = 1;
  a.lw = logw_in ? logw_in + static_cast<size_t>(n) * M : nullptr;
  a.xs = states_in + static_cast<size_t>(n) * M * D;
  a.u = u[n];
  a.estimate = estimate + static_cast<size_t>(n) * D;
  a.so = states_out + static_cast<size_t>(n) * M_out * D;
  a.lo = logw_out ? logw_out + static_cast<size_t>(n) * M_out : nullptr;
  a.io = indices_out ? indices_out + static_cast<size_t>(n) * M_out : nullptr;
  mmf::resample_systematic_trajectory<D, STAGE, false, false>(smem, a, M, M_out, lw_uniform, log_uniform);
}

}  // namespace

extern "C" size_t mmf_pf_reweight_resample_lds_bytes(int M, int mode) {
  const size_t slots = (static_cast<size_t>(M) * (mode ? 8 : 4) + 15) & ~static_cast<size_t>(15);
  return slots + sizeof(Scratch);
}

namespace {
// LDS with the particle states staged behind the CDF; used when one workgroup per CU still fits
size_t staged_lds_bytes(int M, int d, int mode) {
  const size_t slots = (static_cast<size_t>(M) * (mode ? 8 : 4) + 15) & ~static_cast<size_t>(15);
  return slots + ((sizeof(Scratch) + 15) & ~static_cast<size_t>(15)) + static_cast<size_t>(M) * d * sizeof(float);
}
}  // namespace

namespace {
int launch_reweight_resample(const float* loglik, const float* logw_in, const float* states_in, const float* u,
                             float* estimate, float* states_out, float* logw_out, int32_t* indices_out, int N,
                             int M, int M_out, int d, int mode, float alpha, void* stream) {
  if (!loglik || !states_in || !estimate) return MMF_EINVAL;
  if (N < 0 || M < 1 || M_out < 1 || d < 1 || d > MMF_MAX_STATE_DIM || mode < 0 || mode > 2) return MMF_EINVAL;
  // the uniform-weight shortcuts (null logw_in / logw_out) belong to plain resampling
  if ((!logw_in || !logw_out) && (mode == 0 || alpha < 1.f)) return MMF_EINVAL;
  if (mode != 0 && (!u || !states_out || states_out == states_in)) return MMF_EINVAL;
  if (mode == 0 && M_out != M) return MMF_EINVAL;
  if (!(alpha > 0.f && alpha <= 1.f)) return MMF_EINVAL;
  if (M > 65536 || M_out > 65536) return MMF_ETOOLARGE;
  size_t lds = mmf_pf_reweight_resample_lds_bytes(M, mode);
  if (lds > 160 * 1024) return MMF_ETOOLARGE;
  if (N == 0) return 0;
  const bool soft = mode != 0 && alpha < 1.f;
  // stage the states in LDS when occupancy does not pay for it: always if every trajectory gets
  // a CU of its own (N <= 256), otherwise only while two workgroups still fit a CU (<= 80 KB each)
  const size_t staged = staged_lds_bytes(M, d, mode);
  const bool stage = mode != 0 && (N <= 256 ? staged <= 160 * 1024 : staged <= 80 * 1024);
  if (stage) lds = staged;
  // enough threads to give each one a float4 of work, at least one wave
  int block = ((M + 3) / 4 + MMF_WAVE - 1) / MMF_WAVE * MMF_WAVE;
  if (block > kBlock) block = kBlock;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const float lw_uniform = static_cast<float>(-std::log(static_cast<double>(M)));
  const float log_uniform = static_cast<float>(-std::log(static_cast<double>(M_out)));
  if (mode == 1 && !soft && (d == 2 || d == 3)) {
    // few trajectories (round 6): a cluster of workgroups per trajectory, two meetings through L2, the same bits
    int dev = 0, cus = 0;
    if (g_cluster_enabled && hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) {
      const int C = cluster::plan(N, M, M_out, cus);
      if (C > 0) {
        // exchange area: kLanes launches' worth (consecutive launches never share one: a late reader of launch k cannot
        // meet launch k + 1's granules), tags from a counter that never repeats, zeroed once; per device
        constexpr int kLanes = 8, kMaxDevices = 16;
        static mmf::Granule* area[kMaxDevices] = {};
        static unsigned counter = 0;
        if (dev < 0 || dev >= kMaxDevices) return MMF_EINVAL;
        const size_t lane_granules = static_cast<size_t>(cus / 2) * cluster::kGranulesPerTraj;  // plan(): N <= cus / 2
        if (!area[dev]) {
          const size_t bytes = (kLanes * lane_granules + 2) * sizeof(mmf::Granule);
          if (hipMalloc(reinterpret_cast<void**>(&area[dev]), bytes) != hipSuccess) return MMF_EINVAL;
          if (hipMemsetAsync(area[dev], 0, bytes, s) != hipSuccess) return MMF_EINVAL;
        }
        ++counter;
        if (counter == 0) counter = 1;  // tag 0 is "never written"
        cluster::Args ca{};
        ca.loglik = loglik; ca.logw_in = logw_in; ca.states_in = states_in; ca.u = u; ca.estimate = estimate;
        ca.states_out = states_out; ca.logw_out = logw_out; ca.indices_out = indices_out;
        ca.exchange = area[dev] + (counter % kLanes) * lane_granules;
        ca.error_word = reinterpret_cast<unsigned*>(area[dev] + kLanes * lane_granules);
        ca.M = M; ca.M_out = M_out; ca.C = C; ca.tag = counter;
        ca.lw_uniform = lw_uniform; ca.log_uniform = log_uniform;
        const int threads = M / (4 * C);
        const size_t bytes = cluster::lds_bytes(M_out, threads, d);
        if (d == 3) cluster::pf_resample_cluster_kernel<3><<<dim3(C, N), threads, bytes, s>>>(ca);
        else cluster::pf_resample_cluster_kernel<2><<<dim3(C, N), threads, bytes, s>>>(ca);
        MMF_CHECK_LAUNCH();
        return 0;
      }
    }
  }
  if (mode == 1 && !soft && M <= 20000 && M_out <= 20000) {
    // plain systematic resampling: the search-free kernel (offspring boundaries + prefix sum of marks)
    const size_t slots = (static_cast<size_t>(M) * 8 + 15) & ~static_cast<size_t>(15);
    const size_t sc_sz = (sizeof(Scratch) + 15) & ~static_cast<size_t>(15);
    const size_t marks_sz = ((static_cast<size_t>(M_out) + 4) * 4 + 15) & ~static_cast<size_t>(15);
    const size_t base_sz = slots + sc_sz + marks_sz;
    const size_t with_states = base_sz + static_cast<size_t>(M) * d * sizeof(float);
    if (base_sz <= 160 * 1024) {
      const bool st = N <= 256 ? with_states <= 160 * 1024 : with_states <= 80 * 1024;
      const size_t bytes = st ? with_states : base_sz;
#define MMF_K1S_LAUNCH(D, ST)                                                                        \
  {                                                                                                  \
    if (bytes > 64 * 1024) {                                                                         \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pf_resample_systematic_kernel<D, ST>), \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes)); \
      if (e != hipSuccess) return static_cast<int>(e);                                               \
    }                                                                                                \
    pf_resample_systematic_kernel<D, ST><<<N, block, bytes, s>>>(loglik, logw_in, states_in, u, estimate, states_out, \
                                                                 logw_out, indices_out, M, M_out, lw_uniform, log_uniform); \
  }
#define MMF_K1S(D) \
  case D: { if (st) MMF_K1S_LAUNCH(D, true) else MMF_K1S_LAUNCH(D, false) } break;
      switch (d) { MMF_K1S(1) MMF_K1S(2) MMF_K1S(3) MMF_K1S(4) }
#undef MMF_K1S
#undef MMF_K1S_LAUNCH
      MMF_CHECK_LAUNCH();
      return 0;
    }
  }
#define MMF_K1_LAUNCH(D, ST, SO)                                                               \
  {                                                                                            \
    if (lds > 64 * 1024) {                                                                     \
      hipError_t e = hipFuncSetAttribute(                                                      \
          reinterpret_cast<const void*>(&pf_reweight_resample_kernel<D, ST, SO>),              \
          hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));                  \
      if (e != hipSuccess) return static_cast<int>(e);                                         \
    }                                                                                          \
    pf_reweight_resample_kernel<D, ST, SO><<<N, block, lds, s>>>(loglik, logw_in, states_in, u, \
        estimate, states_out, logw_out, indices_out, M, M_out, mode, alpha, lw_uniform, log_uniform); \
  }
#define MMF_K1(D)                                                                              \
  case D: {                                                                                    \
    if (soft) { if (stage) MMF_K1_LAUNCH(D, true, true) else MMF_K1_LAUNCH(D, false, true) }   \
    else if (stage) MMF_K1_LAUNCH(D, true, false) else MMF_K1_LAUNCH(D, false, false)          \
  } break;
  switch (d) {
    MMF_K1(1) MMF_K1(2) MMF_K1(3) MMF_K1(4)
  }
#undef MMF_K1
#undef MMF_K1_LAUNCH
  MMF_CHECK_LAUNCH();
  return 0;
}
}  // namespace

namespace {
// estimation_method = "argmax": one workgroup per trajectory; value = logw_in + loglik (ONE fp32 add, as the
// step-by-step path's torch expression), larger value wins, smaller index on ties (torch.argmax)
template <int D>
__global__ __launch_bounds__(256) void pf_argmax_estimate_kernel(const float* __restrict__ loglik,
                                                                 const float* __restrict__ logw_in,
                                                                 const float* __restrict__ states,
                                                                 float* __restrict__ estimate, int M, float lw_uniform) {
  __shared__ float best_v[256 / MMF_WAVE];
  __shared__ int best_i[256 / MMF_WAVE];
  const int n = blockIdx.x, tid = threadIdx.x;
  const float* ll = loglik + static_cast<size_t>(n) * M;
  const float* lw = logw_in ? logw_in + static_cast<size_t>(n) * M : nullptr;
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  for (int i = tid; i < M; i += blockDim.x) {
    const float v = (lw ? lw[i] : lw_uniform) + ll[i];
    if (v > bv || (v == bv && i < bi) || bi == 0x7fffffff) { bv = v; bi = i; }
  }
  auto better = [](float v, int i, float w, int j) { return v > w || (v == w && i < j); };
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const float ov = __shfl_xor(bv, off);
    const int oi = __shfl_xor(bi, off);
    if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
  }
  if ((tid & (MMF_WAVE - 1)) == 0) { best_v[tid >> 6] = bv; best_i[tid >> 6] = bi; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < static_cast<int>(blockDim.x >> 6); ++w)
      if (better(best_v[w], best_i[w], bv, bi)) { bv = best_v[w]; bi = best_i[w]; }
    if (bi >= M) bi = 0;  // every value NaN: torch.argmax would pick a NaN's index; the filter is lost either way
#pragma unroll
    for (int c = 0; c < D; ++c) estimate[static_cast<size_t>(n) * D + c] = states[(static_cast<size_t>(n) * M + bi) * D + c];
  }
}
}  // namespace

extern "C" int mmf_pf_argmax_estimate(const float* loglik, const float* logw_in, const float* states, float* estimate,
                                      int N, int M, int d, void* stream) {
  if (!loglik || !states || !estimate) return MMF_EINVAL;
  if (N < 0 || M < 1 || d < 1 || d > MMF_MAX_STATE_DIM) return MMF_EINVAL;
  if (N == 0) return 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const float lw_uniform = static_cast<float>(-std::log(static_cast<double>(M)));
  switch (d) {
    case 1: pf_argmax_estimate_kernel<1><<<N, 256, 0, s>>>(loglik, logw_in, states, estimate, M, lw_uniform); break;
    case 2: pf_argmax_estimate_kernel<2><<<N, 256, 0, s>>>(loglik, logw_in, states, estimate, M, lw_uniform); break;
    case 3: pf_argmax_estimate_kernel<3><<<N, 256, 0, s>>>(loglik, logw_in, states, estimate, M, lw_uniform); break;
    default: pf_argmax_estimate_kernel<4><<<N, 256, 0, s>>>(loglik, logw_in, states, estimate, M, lw_uniform); break;
  }
  MMF_CHECK_LAUNCH();
  return 0;
}

extern "C" void mmf_pf_set_resample_cluster(int enabled) { g_cluster_enabled = enabled != 0; }
extern "C" int mmf_pf_get_resample_cluster(void) { return g_cluster_enabled; }

extern "C" int mmf_pf_reweight_resample(const float* loglik, const float* logw_in,
                                        const float* states_in, const float* u, float* estimate,
                                        float* states_out, float* logw_out, int32_t* indices_out,
                                        int N, int M, int M_out, int d, int mode, void* stream) {
  return launch_reweight_resample(loglik, logw_in, states_in, u, estimate, states_out, logw_out, indices_out, N, M,
                                  M_out, d, mode, 1.0f, stream);
}

extern "C" int mmf_pf_reweight_resample_soft(const float* loglik, const float* logw_in,
                                             const float* states_in, const float* u, float* estimate,
                                             float* states_out, float* logw_out, int32_t* indices_out,
                                             int N, int M, int M_out, int d, int mode, float alpha,
                                             void* stream) {
  if (mode == 0) return MMF_EINVAL;
  return launch_reweight_resample(loglik, logw_in, states_in, u, estimate, states_out, logw_out, indices_out, N, M,
                                  M_out, d, mode, alpha, stream);
}
