// Per-set point sub-sampling on the device.
//
// Replaces the per-item numpy work of the reference's sub-sampling experiments:
//   (-pc[:, -1]).argsort()[:K]              Code/dataset.py:196   ESC_pc_temp_maxKSS
//   np.random.permutation(N)[:K]            Code/dataset.py:236   ESC_pc_temp_randKSS
//   the same two selections per frame       Code/utils.py:42,70   pc_maxK / pc_randK
// (449 us per item on the host) by ONE launch per batch: a workgroup sorts the N <= 16384
// (key, point index) pairs of its set in LDS with a bitonic network and writes the first K
// points, already packed as (f, [t,] value) rows.
//   max-K   key = value, descending; equal values keep ascending point order (what a stable
//           argsort of the negated values gives; NaNs last as numpy sorts them)
//   rand-K  key = hash(seed, draw counter, set, point): the first K of a uniformly random
//           permutation.  The reference draws from the global numpy RNG, so only the
//           distribution can be reproduced, not the stream.
// HBM traffic: N values read + K rows written per set; the sort itself never leaves LDS.
#include "pca_common.h"

#include <stdint.h>

#include <mutex>

namespace pca {
namespace {

// ascending order of the returned key = descending order of v; -0 == +0; NaN last
__device__ __forceinline__ uint32_t desc_key(float v) {
  if (v != v) return 0xffffffffu;
  v += 0.0f;                                        // -0 -> +0
  const uint32_t u = __float_as_uint(v);
  const uint32_t asc = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ~asc;
}
// splitmix64 finaliser
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

__global__ __launch_bounds__(1024) void k_subsample(
    const float* __restrict__ spec, int64_t stride_f, int64_t stride_t, int64_t stride_s,
    const float* __restrict__ farr, const float* __restrict__ tarr,     // tarr null: 2-D rows
    const int64_t* __restrict__ idx, int F, int Nt, int K, int mode, uint64_t seed,
    uint64_t draw, const int32_t* __restrict__ draw_dev, int Np, float* __restrict__ out,
    int32_t* __restrict__ sel, const int64_t* __restrict__ labels,
    int64_t* __restrict__ labels_out) {
  extern __shared__ uint64_t keys[];                 // Np = power of two >= N
  const int b = blockIdx.x, tid = threadIdx.x;
  if (draw_dev != nullptr) draw += (uint64_t)(uint32_t)draw_dev[0];   // device-side counter
  const int64_t set = idx[b];
  const int N = F * Nt;
  if (labels != nullptr && labels_out != nullptr && tid == 0) labels_out[b] = labels[set];
  const float* __restrict__ base = spec + set * stride_s;
  // one stream per (seed, draw, batch slot, set): a set that appears twice in a batch gets two
  // independent selections, as two __getitem__ calls of the reference would
  const uint64_t stream = mix64(seed ^ mix64(draw * 0x9e3779b97f4a7c15ull + (uint64_t)set) ^
                                mix64(0x632be59bd9b4e019ull * (uint64_t)(b + 1)));
  for (int p = tid; p < Np; p += 1024) {
    uint64_t k = ~0ull;
    if (p < N) {
      uint32_t hi;
      if (mode == 0) {
        const int t = p / F, f = p - t * F;
        hi = desc_key(base[f * stride_f + t * stride_t]);
      } else {
        hi = (uint32_t)(mix64(stream + (uint64_t)p * 0xd1342543de82ef95ull) >> 32);
      }
      k = ((uint64_t)hi << 32) | (uint32_t)p;
    }
    keys[p] = k;
  }
  __syncthreads();
  // bitonic network, ascending; every pair (i, i | j) is touched by exactly one thread
  for (int k = 2; k <= Np; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < (Np >> 1); t += 1024) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const int l = i | j;
        const uint64_t a = keys[i], c = keys[l];
        const bool up = (i & k) == 0;
        if ((a > c) == up) {
          keys[i] = c;
          keys[l] = a;
        }
      }
      __syncthreads();
    }
  }
  const int din = tarr != nullptr ? 3 : 2;
  for (int q = tid; q < K; q += 1024) {
    const int p = (int)(uint32_t)keys[q];
    const int t = p / F, f = p - t * F;
    float* o = out + ((int64_t)b * K + q) * din;
    o[0] = farr[f];
    if (tarr != nullptr) o[1] = tarr[t];
    o[din - 1] = base[f * stride_f + t * stride_t];
    if (sel != nullptr) sel[(int64_t)b * K + q] = p;
  }
}

// ---------------------------------------------------------------------------------
// Importance sampling (Code/dataset.py:243-289, ESC_pc_temp_importancerandKSS): per set
//   G0 = |d/df xt| + |d/dt xt|            torch.gradient: central differences, one-sided edges
//   heat = corr2d(G0, kern[2][winF], zero 'same' padding) + 1e-6     (pad 0/1 over f,
//          (winF-1)/2 / rest over t, as F.conv2d(padding='same') pads)
//   choice 1: the K largest heat values, descending;  choice 0: K draws with replacement from
//   the distribution heat / sum(heat).
// The selected FLAT heat-map index i = f*Nt + t is then used by the reference as the row of
// the time-major point table (row p <-> f' = p % F, t' = p / F).  That row mismatch is part of
// the reference's behaviour and is reproduced: output row = (farr[i % F], tarr[i / F],
// xt[i % F][i / F]).
// LDS: xt and G0 (4N bytes each), later aliased by the sort keys / the CDF.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_importance(
    const float* __restrict__ spec, int64_t stride_f, int64_t stride_t, int64_t stride_s,
    const float* __restrict__ farr, const float* __restrict__ tarr,
    const int64_t* __restrict__ idx, int F, int Nt, int K, int choice,
    const float* __restrict__ kern, int winF, uint64_t seed, uint64_t draw,
    const int32_t* __restrict__ draw_dev, int Np, float* __restrict__ out,
    int32_t* __restrict__ sel, float* __restrict__ heat_out,
    const int64_t* __restrict__ labels, int64_t* __restrict__ labels_out) {
  extern __shared__ uint64_t lds64[];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (draw_dev != nullptr) draw += (uint64_t)(uint32_t)draw_dev[0];   // device-side counter
  const int64_t set = idx[b];
  const int N = F * Nt;
  float* sx = reinterpret_cast<float*>(lds64);        // xt [F][Nt], later the heat map
  float* sg = sx + N;                                 // G0 [F][Nt]
  if (labels != nullptr && labels_out != nullptr && tid == 0) labels_out[b] = labels[set];
  const float* __restrict__ base = spec + set * stride_s;
  for (int i = tid; i < N; i += 1024) {
    const int f = i / Nt, t = i - f * Nt;
    sx[i] = base[f * stride_f + t * stride_t];
  }
  __syncthreads();
  for (int i = tid; i < N; i += 1024) {
    const int f = i / Nt, t = i - f * Nt;
    const float gf = f == 0       ? sx[Nt + t] - sx[t]
                     : f == F - 1 ? sx[i] - sx[i - Nt]
                                  : (sx[i + Nt] - sx[i - Nt]) * 0.5f;
    const float gt = t == 0        ? sx[i + 1] - sx[i]
                     : t == Nt - 1 ? sx[i] - sx[i - 1]
                                   : (sx[i + 1] - sx[i - 1]) * 0.5f;
    sg[i] = fabsf(gf) + fabsf(gt);
  }
  __syncthreads();
  const int padl = (winF - 1) / 2;
  for (int i = tid; i < N; i += 1024) {
    const int f = i / Nt, t = i - f * Nt;
    float acc = 0.f;
    for (int a = 0; a < 2; ++a) {
      if (f + a >= F) break;
      for (int c = 0; c < winF; ++c) {
        const int tt = t + c - padl;
        if (tt >= 0 && tt < Nt) acc = fmaf(sg[(f + a) * Nt + tt], kern[a * winF + c], acc);
      }
    }
    sx[i] = acc + 1.0e-6f;
  }
  __syncthreads();
  if (heat_out != nullptr)
    for (int i = tid; i < N; i += 1024) heat_out[(int64_t)b * N + i] = sx[i];
  const uint64_t stream = mix64(seed ^ mix64(draw * 0x9e3779b97f4a7c15ull + (uint64_t)set) ^
                                mix64(0x632be59bd9b4e019ull * (uint64_t)(b + 1)));
  if (choice == 1) {
    // keys alias xt/G0: collect this thread's heat values first
    uint32_t hk[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int i = tid + 1024 * u;
      hk[u] = i < N ? desc_key(sx[i]) : 0xffffffffu;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int i = tid + 1024 * u;
      if (i < Np) lds64[i] = i < N ? (((uint64_t)hk[u] << 32) | (uint32_t)i) : ~0ull;
    }
    __syncthreads();
    for (int k = 2; k <= Np; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int t = tid; t < (Np >> 1); t += 1024) {
          const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
          const int l = i | j;
          const uint64_t a = lds64[i], c = lds64[l];
          const bool up = (i & k) == 0;
          if ((a > c) == up) {
            lds64[i] = c;
            lds64[l] = a;
          }
        }
        __syncthreads();
      }
    }
  } else {
    // inclusive CDF of the heat map in place: per-thread chunks, then a scan of the 1024 sums
    __shared__ double part[1024];
    const int per = (N + 1023) / 1024;
    const int lo = tid * per, hi = (lo + per < N) ? lo + per : N;
    double s = 0.0;
    for (int i = lo; i < hi; ++i) s += (double)sx[i];
    part[tid] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      const double v = tid >= o ? part[tid - o] : 0.0;
      __syncthreads();
      part[tid] += v;
      __syncthreads();
    }
    double run = tid > 0 ? part[tid - 1] : 0.0;
    for (int i = lo; i < hi; ++i) {
      run += (double)sx[i];
      sg[i] = (float)run;                   // CDF lives in the G0 slot
    }
    __syncthreads();
  }
  const double total = choice == 1 ? 0.0 : (double)sg[N - 1];
  for (int q = tid; q < K; q += 1024) {
    int i;
    if (choice == 1) {
      i = (int)(uint32_t)lds64[q];
    } else {
      const uint64_t r = mix64(stream + (uint64_t)q * 0xd1342543de82ef95ull);
      const float target = (float)(((double)(r >> 11) * (1.0 / 9007199254740992.0)) * total);
      int a = 0, c = N - 1;                 // first index whose CDF exceeds the target
      while (a < c) {
        const int m = (a + c) >> 1;
        if (sg[m] > target) c = m; else a = m + 1;
      }
      i = a;
    }
    const int fq = i % F, tq = i / F;       // row i of the time-major point table
    float* o = out + ((int64_t)b * K + q) * 3;
    o[0] = farr[fq];
    o[1] = tarr[tq];
    o[2] = base[fq * stride_f + tq * stride_t];
    if (sel != nullptr) sel[(int64_t)b * K + q] = i;
  }
}

// ESC_pc_ss batches: rows of two frame-major tables (per-frame coordinates and values)
__global__ __launch_bounds__(256) void k_pack_2d_ss(const float* __restrict__ x_tk,
                                                     const float* __restrict__ f_tk,
                                                     const int64_t* __restrict__ idx, int K,
                                                     float* __restrict__ out,
                                                     const int64_t* __restrict__ labels,
                                                     int64_t* __restrict__ labels_out) {
  const int b = blockIdx.y;
  const int64_t frame = idx[b];
  if (labels != nullptr && labels_out != nullptr && blockIdx.x == 0 && threadIdx.x == 0)
    labels_out[b] = labels[frame];
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= K) return;
  float2 p;
  p.x = f_tk[frame * K + q];
  p.y = x_tk[frame * K + q];
  reinterpret_cast<float2*>(out)[(int64_t)b * K + q] = p;
}

}  // namespace
}  // namespace pca

extern "C" {

int pca_subsample_points(const float* spec, int64_t stride_f, int64_t stride_t,
                         int64_t stride_s, const float* farr, const float* tarr,
                         const int64_t* idx, int B, int F, int Nt, int K, int mode,
                         uint64_t seed, uint64_t draw, const int32_t* draw_dev, float* out,
                         int32_t* sel, const int64_t* labels, int64_t* labels_out,
                         void* stream) {
  PCA_REQUIRE(spec && farr && idx && out, "subsample_points: null pointer");
  PCA_REQUIRE(B > 0 && F > 0 && Nt > 0, "subsample_points: B=%d F=%d Nt=%d", B, F, Nt);
  PCA_REQUIRE(tarr != nullptr || Nt == 1, "subsample_points: Nt=%d needs tarr", Nt);
  PCA_REQUIRE(mode == 0 || mode == 1, "subsample_points: mode=%d", mode);
  const int64_t N = (int64_t)F * Nt;
  PCA_REQUIRE(N <= 16384, "subsample_points: %lld points per set (max 16384)", (long long)N);
  PCA_REQUIRE(K > 0 && K <= N, "subsample_points: K=%d outside [1, %lld]", K, (long long)N);
  int Np = 2;
  while (Np < N) Np <<= 1;
  static std::once_flag once;
  std::call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pca::k_subsample),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  });
  hipLaunchKernelGGL(pca::k_subsample, dim3((unsigned)B), dim3(1024),
                     (size_t)Np * sizeof(uint64_t), pca::as_stream(stream), spec, stride_f,
                     stride_t, stride_s, farr, tarr, idx, F, Nt, K, mode, seed, draw, draw_dev,
                     Np, out, sel, labels, labels_out);
  return pca::check_launch("k_subsample");
}

int pca_importance_points(const float* spec, int64_t stride_f, int64_t stride_t,
                          int64_t stride_s, const float* farr, const float* tarr,
                          const int64_t* idx, int B, int F, int Nt, int K, int choice,
                          const float* kern, int winF, uint64_t seed, uint64_t draw,
                          const int32_t* draw_dev, float* out, int32_t* sel, float* heat,
                          const int64_t* labels, int64_t* labels_out, void* stream) {
  PCA_REQUIRE(spec && farr && tarr && idx && kern && out, "importance_points: null pointer");
  PCA_REQUIRE(B > 0 && F >= 2 && Nt >= 2, "importance_points: B=%d F=%d Nt=%d (torch.gradient "
              "needs two samples per axis)", B, F, Nt);
  PCA_REQUIRE(choice == 0 || choice == 1, "importance_points: choice=%d", choice);
  PCA_REQUIRE(winF >= 1 && winF <= 4096, "importance_points: winF=%d", winF);
  const int64_t N = (int64_t)F * Nt;
  PCA_REQUIRE(N <= 16384, "importance_points: %lld points per set (max 16384)", (long long)N);
  PCA_REQUIRE(K > 0 && (choice == 0 || K <= N), "importance_points: K=%d with %lld points", K,
              (long long)N);
  int Np = 2;
  while (Np < N) Np <<= 1;
  static std::once_flag once;
  std::call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pca::k_importance),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  });
  size_t lds = (size_t)Np * sizeof(uint64_t);
  if (lds < (size_t)N * 8) lds = (size_t)N * 8;
  hipLaunchKernelGGL(pca::k_importance, dim3((unsigned)B), dim3(1024), lds,
                     pca::as_stream(stream), spec, stride_f, stride_t, stride_s, farr, tarr, idx,
                     F, Nt, K, choice, kern, winF, seed, draw, draw_dev, Np, out, sel, heat,
                     labels, labels_out);
  return pca::check_launch("k_importance");
}

int pca_pack_points_2d_ss(const float* x_tk, const float* f_tk, const int64_t* idx, int B,
                          int K, float* out, const int64_t* labels, int64_t* labels_out,
                          void* stream) {
  PCA_REQUIRE(x_tk && f_tk && idx && out, "pack_points_2d_ss: null pointer");
  PCA_REQUIRE(B > 0 && K > 0 && B <= 65535, "pack_points_2d_ss: B=%d K=%d", B, K);
  hipLaunchKernelGGL(pca::k_pack_2d_ss, dim3((unsigned)pca::cdiv(K, 256), (unsigned)B),
                     dim3(256), 0, pca::as_stream(stream), x_tk, f_tk, idx, K, out, labels,
                     labels_out);
  return pca::check_launch("k_pack_2d_ss");
}
}
