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
    uint64_t draw, int Np, float* __restrict__ out, int32_t* __restrict__ sel,
    const int64_t* __restrict__ labels, int64_t* __restrict__ labels_out) {
  extern __shared__ uint64_t keys[];                 // Np = power of two >= N
  const int b = blockIdx.x, tid = threadIdx.x;
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
                         uint64_t seed, uint64_t draw, float* out, int32_t* sel,
                         const int64_t* labels, int64_t* labels_out, void* stream) {
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
                     stride_t, stride_s, farr, tarr, idx, F, Nt, K, mode, seed, draw, Np, out,
                     sel, labels, labels_out);
  return pca::check_launch("k_subsample");
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
