// Generic batched / strided fp32 GEMM for the exact (PCA_MODE_F32) path.
//
// One 256-thread workgroup computes a 64x64 tile of C with a 4x4 register block per
// thread; A and B tiles go through LDS in K-steps of 16.  Arbitrary strides cover the
// transposed uses (dX = dY.W, dW = dY^T.X) and the head-strided attention GEMMs of
// set_transformer-master/modules.py:24-29 without materialising split/cat copies.
// K can be split over workgroups (dW reductions over B*N rows) with fp32 atomics.
#include "pca_common.h"

namespace pca {

namespace {
constexpr int BM = 64, BN = 64, BK = 16, PAD = 4;

__global__ __launch_bounds__(256) void k_gemm_f32(pca_gemm_desc g, const float* __restrict__ A,
                                                   const float* __restrict__ B,
                                                   const float* __restrict__ bias,
                                                   float* __restrict__ C, int split_k,
                                                   int64_t kchunk) {
  __shared__ __attribute__((aligned(16))) float As[BK][BM + PAD];
  __shared__ __attribute__((aligned(16))) float Bs[BK][BN + PAD];

  const int tid = threadIdx.x;
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

  const bool a_kcontig = (g.sa_k == 1);
  const bool b_ncontig = (g.sb_n == 1);

  const int tx = tid & 15, ty = tid >> 4;
  float acc[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[r][c] = 0.f;

  // The next K-step's operands are fetched into registers while the current one is multiplied:
  // with the loads issued only after the barrier, a [4096 x 256 x 256] product (16 K-steps, one
  // workgroup per CU) spent 40 us waiting on 16 dependent memory round trips.
  float ra[4], rb[4];
  auto fetch = [&](int64_t kt) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int idx = tid + e * 256;
      int i, k;
      if (a_kcontig) { k = idx & (BK - 1); i = idx >> 4; }
      else           { i = idx & (BM - 1); k = idx >> 6; }
      const int64_t gi = m0 + i, gk = kt + k;
      ra[e] = (gi < g.M && gk < k_end) ? A[gi * g.sa_m + gk * g.sa_k] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int idx = tid + e * 256;
      int j, k;
      if (b_ncontig) { j = idx & (BN - 1); k = idx >> 6; }
      else           { k = idx & (BK - 1); j = idx >> 4; }
      const int64_t gj = n0 + j, gk = kt + k;
      rb[e] = (gj < g.N && gk < k_end) ? B[gk * g.sb_k + gj * g.sb_n] : 0.f;
    }
  };
  if (k_begin < k_end) fetch(k_begin);
  for (int64_t kt = k_begin; kt < k_end; kt += BK) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int idx = tid + e * 256;
      int i, k;
      if (a_kcontig) { k = idx & (BK - 1); i = idx >> 4; }
      else           { i = idx & (BM - 1); k = idx >> 6; }
      As[k][i] = ra[e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int idx = tid + e * 256;
      int j, k;
      if (b_ncontig) { j = idx & (BN - 1); k = idx >> 6; }
      else           { k = idx & (BK - 1); j = idx >> 4; }
      Bs[k][j] = rb[e];
    }
    __syncthreads();
    if (kt + BK < k_end) fetch(kt + BK);
#pragma unroll
    for (int k = 0; k < BK; ++k) {
      const float4 a = *reinterpret_cast<const float4*>(&As[k][ty * 4]);
      const float4 b = *reinterpret_cast<const float4*>(&Bs[k][tx * 4]);
      const float av[4] = {a.x, a.y, a.z, a.w};
      const float bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r][c] = fmaf(av[r], bv[c], acc[r][c]);
    }
    __syncthreads();
  }

#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t i = m0 + ty * 4 + r;
    if (i >= g.M) continue;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int64_t j = n0 + tx * 4 + c;
      if (j >= g.N) continue;
      float v = g.alpha * acc[r][c];
      if (bias != nullptr && ks == 0) v += bias[j];
      float* dst = C + i * g.sc_m + j;
      if (split_k > 1) atomicAdd(dst, v);
      else if (g.accumulate) *dst += v;
      else *dst = v;
    }
  }
}
}  // namespace

int gemm_f32(const pca_gemm_desc& gin, const float* A, const float* B, const float* bias,
             float* C, hipStream_t st) {
  pca_gemm_desc g = gin;
  PCA_REQUIRE(A && B && C, "gemm_f32: null operand");
  PCA_REQUIRE(g.M >= 0 && g.N >= 0 && g.K >= 0, "gemm_f32: negative extent");
  if (g.nb1 <= 0) g.nb1 = 1;
  if (g.nb2 <= 0) g.nb2 = 1;
  if (g.M == 0 || g.N == 0) return PCA_OK;
  const int64_t tiles_m = cdiv(g.M, BM), tiles_n = cdiv(g.N, BN);
  const int64_t nbatch = (int64_t)g.nb1 * g.nb2;
  int split = g.split_k;
  if (split <= 0) {
    // small outputs with a long K (weight gradients over B*N or B*m rows): split K until the
    // launch has ~512 workgroups, but keep at least 64 of K per workgroup
    split = 1;
    const int64_t wgs = tiles_m * tiles_n * nbatch;
    if (g.accumulate && wgs < 256 && g.K >= 256) {   // atomics need an initialised C
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
              "gemm_f32: grid too large (N tiles %lld, batch*split %lld)",
              (long long)tiles_n, (long long)(nbatch * split));
  dim3 grid((unsigned)tiles_m, (unsigned)tiles_n, (unsigned)(nbatch * split));
  const double work = (double)g.M * g.N * g.K * nbatch;
  ProfScope ps(PCA_K_GEMM_F32, st, 2.0 * work,
               4.0 * nbatch * ((double)g.M * g.K + (double)g.K * g.N + (double)g.M * g.N));
  hipLaunchKernelGGL(k_gemm_f32, grid, dim3(256), 0, st, g, A, B, bias, C, split, kchunk);
  ps.end();
  return check_launch("k_gemm_f32");
}

}  // namespace pca

extern "C" int pca_gemm_f32(const pca_gemm_desc* g, const float* A, const float* B,
                            const float* bias, float* C, void* stream) {
  PCA_REQUIRE(g != nullptr, "pca_gemm_f32: null descriptor");
  return pca::gemm_f32(*g, A, B, bias, C, pca::as_stream(stream));
}
