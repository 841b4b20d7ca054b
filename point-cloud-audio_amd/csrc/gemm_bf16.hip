// Generic batched / strided GEMM with bf16 MFMA operands and fp32 accumulation: what
// PCA_MODE_BF16 uses for the blocks that have no fused kernel (d = 256 / m = 32 of BASELINE
// configs[3], the shipped d = 64 / 8-head models).  Same descriptor, strides, batching, split-K
// and epilogue as k_gemm_f32 (gemm_f32.hip); the inner product is v_mfma_f32_16x16x32_bf16.
//
// A 256-thread workgroup computes a 128x128 tile of C; wave w owns the 64x64 quadrant
// (w >> 1, w & 1) as 4x4 MFMA tiles (16 MFMAs per 8 fragment reads).  fp32 operands are rounded
// to bf16 while they are staged: As[m][k] and Bs[n][k] (k contiguous, 80-byte rows: 16-byte
// fragment reads, rows spread over the LDS banks).  Staging is vectorised along whichever index
// is contiguous in memory and the next K tile is prefetched into registers under the MFMAs.
#include "pca_common.h"

#include <stdint.h>
#include "mfma_common.hpp"

namespace pca {

namespace {
constexpr int BM = 128, BN = 128, BK = 32, PITCH = 80;   // bytes per LDS row (32 bf16 + pad)

// Stage a [128 rows][32 k] operand tile into LDS as bf16, rows = the non-contracted index.
//   elem(i, k) = P[i * s_row + k * s_k]; rows >= n_rows and k >= k_left read as zero.
// KVEC: s_k == 1   -> float4 along k, one 8-byte LDS store
// RVEC: s_row == 1 -> float4 along the rows, 4x4 register transpose, 8-byte LDS stores
// else scalar.  VEC needs 16-byte aligned addresses (checked by the host).
template <int MODE>
__device__ __forceinline__ void stage_load(const float* __restrict__ P, int64_t s_row, int64_t s_k,
                                           int64_t n_rows, int64_t k_left, int tid,
                                           float (&v)[16]) {
  if (MODE == 0) {              // KVEC: 128 x 8 float4; thread -> 4 of them
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int idx = tid + e * 256;
      const int i = idx >> 3, k4 = (idx & 7) * 4;
      const float* src = P + i * s_row + k4;
      if (i < n_rows && k4 + 3 < k_left) {
        const float4 t = *reinterpret_cast<const float4*>(src);
        v[4 * e] = t.x; v[4 * e + 1] = t.y; v[4 * e + 2] = t.z; v[4 * e + 3] = t.w;
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) v[4 * e + u] = (i < n_rows && k4 + u < k_left) ? src[u] : 0.f;
      }
    }
  } else if (MODE == 1) {       // RVEC: one 4 rows x 4 k block per thread, float4 along the rows
    const int i4 = (tid & 31) * 4, k4 = (tid >> 5) * 4;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float* src = P + i4 + (k4 + u) * s_k;
      if (k4 + u < k_left && i4 + 3 < n_rows) {
        const float4 t = *reinterpret_cast<const float4*>(src);
        v[4 * u] = t.x; v[4 * u + 1] = t.y; v[4 * u + 2] = t.z; v[4 * u + 3] = t.w;
      } else {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
          v[4 * u + rr] = (k4 + u < k_left && i4 + rr < n_rows) ? src[rr] : 0.f;
      }
    }
  } else {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int idx = tid + e * 256;
      const int i = idx >> 5, k = idx & 31;
      v[e] = (i < n_rows && k < k_left) ? P[i * s_row + k * s_k] : 0.f;
    }
  }
}
template <int MODE>
__device__ __forceinline__ void stage_store(char* S, int tid, const float (&v)[16]) {
  if (MODE == 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int idx = tid + e * 256;
      const int i = idx >> 3, k4 = (idx & 7) * 4;
      bf16x4 h;
      h[0] = (__bf16)v[4 * e]; h[1] = (__bf16)v[4 * e + 1];
      h[2] = (__bf16)v[4 * e + 2]; h[3] = (__bf16)v[4 * e + 3];
      *reinterpret_cast<bf16x4*>(S + i * PITCH + k4 * 2) = h;
    }
  } else if (MODE == 1) {       // transposed in registers: 8-byte stores, k contiguous
    const int i4 = (tid & 31) * 4, k4 = (tid >> 5) * 4;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      bf16x4 h;
      h[0] = (__bf16)v[rr]; h[1] = (__bf16)v[4 + rr];
      h[2] = (__bf16)v[8 + rr]; h[3] = (__bf16)v[12 + rr];
      *reinterpret_cast<bf16x4*>(S + (i4 + rr) * PITCH + k4 * 2) = h;
    }
  } else {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int idx = tid + e * 256;
      const int i = idx >> 5, k = idx & 31;
      *reinterpret_cast<__bf16*>(S + i * PITCH + k * 2) = (__bf16)v[e];
    }
  }
}

template <int MA, int MB>
__global__ __launch_bounds__(256) void k_gemm_bf16(pca_gemm_desc g, const float* __restrict__ A,
                                                    const float* __restrict__ B,
                                                    const float* __restrict__ bias,
                                                    float* __restrict__ C, int split_k,
                                                    int64_t kchunk) {
  __shared__ __attribute__((aligned(16))) char As[BM * PITCH];
  __shared__ __attribute__((aligned(16))) char Bs[BN * PITCH];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, gq = lane >> 4;
  const int zb = blockIdx.z / split_k;
  const int ks = blockIdx.z % split_k;
  const int z1 = zb / g.nb2, z2 = zb % g.nb2;
  A += z1 * g.sa_b1 + z2 * g.sa_b2;
  B += z1 * g.sb_b1 + z2 * g.sb_b2;
  C += z1 * g.sc_b1 + z2 * g.sc_b2;

  const int64_t m0 = (int64_t)blockIdx.x * BM;
  const int64_t n0 = (int64_t)blockIdx.y * BN;
  const int64_t k_begin = (int64_t)ks * kchunk;
  const int64_t k_end = (k_begin + kchunk < g.K) ? (k_begin + kchunk) : g.K;
  const int wm = 64 * (wave >> 1), wn = 64 * (wave & 1);
  const float* Ab = A + m0 * g.sa_m;
  const float* Bb = B + n0 * g.sb_n;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  float av[16], bv[16];
  stage_load<MA>(Ab + k_begin * g.sa_k, g.sa_m, g.sa_k, g.M - m0, k_end - k_begin, tid, av);
  stage_load<MB>(Bb + k_begin * g.sb_k, g.sb_n, g.sb_k, g.N - n0, k_end - k_begin, tid, bv);
  for (int64_t kt = k_begin; kt < k_end; kt += BK) {
    __syncthreads();                     // previous tile consumed
    stage_store<MA>(As, tid, av);
    stage_store<MB>(Bs, tid, bv);
    __syncthreads();
    if (kt + BK < k_end) {               // next tile's loads fly under the MFMAs
      stage_load<MA>(Ab + (kt + BK) * g.sa_k, g.sa_m, g.sa_k, g.M - m0, k_end - kt - BK, tid, av);
      stage_load<MB>(Bb + (kt + BK) * g.sb_k, g.sb_n, g.sb_k, g.N - n0, k_end - kt - BK, tid, bv);
    }
    bf16x8 af[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      af[i] = *reinterpret_cast<const bf16x8*>(As + (wm + 16 * i + r) * PITCH + 16 * gq);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bf16x8 bf = *reinterpret_cast<const bf16x8*>(Bs + (wn + 16 * j + r) * PITCH + 16 * gq);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][j] = mfma32(af[i], bf, acc[i][j]);
    }
  }

  // D[row = 4 gq + e][col = r] of tile (i, j)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t row = m0 + wm + 16 * i + 4 * gq + e;
      if (row >= g.M) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t col = n0 + wn + 16 * j + r;
        if (col >= g.N) continue;
        float v = g.alpha * acc[i][j][e];
        if (bias != nullptr && ks == 0) v += bias[col];
        float* dst = C + row * g.sc_m + col;
        if (split_k > 1) atomicAdd(dst, v);
        else if (g.accumulate) *dst += v;
        else *dst = v;
      }
    }
}

// staging mode of an operand: 0 k-vectors, 1 row-vectors, 2 scalar
inline int stage_mode(const float* p, int64_t s_row, int64_t s_k, int64_t b1, int64_t b2) {
  const bool al = (reinterpret_cast<uintptr_t>(p) & 15) == 0 && b1 % 4 == 0 && b2 % 4 == 0;
  if (s_k == 1 && s_row % 4 == 0 && al) return 0;
  if (s_row == 1 && s_k % 4 == 0 && al) return 1;
  return 2;
}
}  // namespace

int gemm_bf16(const pca_gemm_desc& gin, const float* A, const float* B, const float* bias,
              float* C, hipStream_t st) {
  pca_gemm_desc g = gin;
  PCA_REQUIRE(A && B && C, "gemm_bf16: null operand");
  PCA_REQUIRE(g.M >= 0 && g.N >= 0 && g.K >= 0, "gemm_bf16: negative extent");
  if (g.nb1 <= 0) g.nb1 = 1;
  if (g.nb2 <= 0) g.nb2 = 1;
  if (g.M == 0 || g.N == 0) return PCA_OK;
  const int64_t tiles_m = cdiv(g.M, BM), tiles_n = cdiv(g.N, BN);
  const int64_t nbatch = (int64_t)g.nb1 * g.nb2;
  int split = g.split_k;
  if (split <= 0) {          // same policy as gemm_f32: only an initialised C can take atomics
    split = 1;
    const int64_t wgs = tiles_m * tiles_n * nbatch;
    if (g.accumulate && wgs < 256 && g.K >= 256) {
      int64_t want = 512 / wgs;
      int64_t maxs = g.K / 64;
      split = (int)(want < maxs ? want : maxs);
      if (split < 1) split = 1;
    }
  }
  int64_t ksteps = cdiv(g.K > 0 ? g.K : 1, BK);
  if (split > ksteps) split = (int)ksteps;
  const int64_t kchunk = cdiv(ksteps, split) * BK;
  split = (int)cdiv(g.K > 0 ? g.K : 1, kchunk);
  PCA_REQUIRE(tiles_n <= 65535 && nbatch * split <= 65535,
              "gemm_bf16: grid too large (N tiles %lld, batch*split %lld)",
              (long long)tiles_n, (long long)(nbatch * split));
  dim3 grid((unsigned)tiles_m, (unsigned)tiles_n, (unsigned)(nbatch * split));
  // K chunks start at multiples of 32 floats, so the vector paths stay aligned
  const int ma = stage_mode(A, g.sa_m, g.sa_k, g.sa_b1, g.sa_b2);
  const int mb = stage_mode(B, g.sb_n, g.sb_k, g.sb_b1, g.sb_b2);
#define PCA_GEMM_CASE(X, Y)                                                                   \
  if (ma == X && mb == Y)                                                                     \
    hipLaunchKernelGGL((k_gemm_bf16<X, Y>), grid, dim3(256), 0, st, g, A, B, bias, C, split, kchunk)
  PCA_GEMM_CASE(0, 0); PCA_GEMM_CASE(0, 1); PCA_GEMM_CASE(0, 2);
  PCA_GEMM_CASE(1, 0); PCA_GEMM_CASE(1, 1); PCA_GEMM_CASE(1, 2);
  PCA_GEMM_CASE(2, 0); PCA_GEMM_CASE(2, 1); PCA_GEMM_CASE(2, 2);
#undef PCA_GEMM_CASE
  return check_launch("k_gemm_bf16");
}

}  // namespace pca
