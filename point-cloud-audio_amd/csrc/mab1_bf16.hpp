// Declarations shared by the fused bf16 mab1 forward / backward translation units.
#pragma once
#include "pca_common.h"
#include "mfma_common.hpp"

namespace pca {

constexpr int M1_TP = 128;     // points per workgroup tile (4 waves x 32)
constexpr int M1_NB = 2;       // 16-point blocks per wave

struct Mab1Saved {
  __bf16 *KpP, *VpP, *Kt, *Vt, *QpS, *OS;
  uint32_t* mask;
};
size_t mab1_carve_saved(const pca_mab_shape& s, Mab1Saved* out, void* base);

// one 32-bit word per lane covers 8 feature tiles (4 bits each)
template <int D>
__host__ __device__ __forceinline__ int64_t mab1_mask_index(int b, int tiles_per_set, int tile,
                                                            int wave, int nb, int w, int lane) {
  const int64_t pb = ((int64_t)b * tiles_per_set + tile) * (M1_TP / 16) + wave * M1_NB + nb;
  return (pb * (D / 128) + w) * 64 + lane;
}

// saved-for-backward block of the fused mab0 (few queries, many keys)
struct Mab0Saved {
  float* Qp;      // [m][d]     fc_q(I)
  float* Gf;      // [Rpad][dk] scale*log2e * Qp_h Wk_h   (fp32)
  __bf16* Gb;     // same, bf16 (MFMA operand)
  float* T;       // [B][R][dk] A X
  float* LSE;     // [B][R]     log2-domain
  float *O, *Z;   // [B][m][d]
  float *WvT, *WoT;   // transposed fp32 weights ([in][out]) for the per-set epilogue
  float *Tp, *Mp, *Lp;   // per point-range partials of the attention (merged by the epilogue)
};
int mab0_splits(const pca_mab_shape& s);
size_t mab0_carve_saved(const pca_mab_shape& s, Mab0Saved* out, void* base);

// dst[c][r] = src[r][c]  (fp32): gives the per-set row-GEMM kernels coalesced weight reads
int transpose_f32(const float* src, float* dst, int rows, int cols, hipStream_t st);

// acc[q] += sum_c sX[q*ldx + c] * WT[c*ldw + f]   for q < MQ : thread-owned output column f,
// activations broadcast from LDS, weights read coalesced (consecutive threads = consecutive f)
template <int MQ>
__device__ __forceinline__ void col_gemm(const float* sX, int ldx, const float* __restrict__ WT,
                                         int ldw, int K, int f, float (&acc)[MQ]) {
  // 16 independent weight loads in flight per thread: these per-set kernels run at one
  // workgroup per set and are otherwise bound by L2 latency, not bandwidth
  int c = 0;
  for (; c + 16 <= K; c += 16) {
    float w[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) w[u] = WT[(int64_t)(c + u) * ldw + f];
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int q = 0; q < MQ; ++q) acc[q] = fmaf(sX[q * ldx + c + u], w[u], acc[q]);
  }
  for (; c < K; ++c) {
    const float w = WT[(int64_t)c * ldw + f];
#pragma unroll
    for (int q = 0; q < MQ; ++q) acc[q] = fmaf(sX[q * ldx + c], w, acc[q]);
  }
}

// ---- batched weight-gradient reduction on the MFMA (mab1_bwd_bf16.hip) ------------------
// job: dW[128 x 128] += G[M x 128]^T . A[M x 128]  (only output rows [g_lo, g_hi) are written:
// block-diagonal per-head products), db[128] += column sums of G (nullable).
struct WgradJob {
  const void* G;
  const void* A;
  float* dW;
  float* db;
  int64_t M;
  int g_lo, g_hi;
};
struct WgradJobs {
  WgradJob j[6];
  int n;
};
// g_bf16 / a_bf16: element type of every job's G / A (bf16 or fp32)
int wgrad128_launch(const WgradJobs& jobs, bool g_bf16, bool a_bf16, int rows_per_wg,
                    hipStream_t st);
// dW[128 x dq] += G[M x 128]^T . X_h[M x dq] (dq <= 4; X_h = X + head(f)*x_head_stride), db += colsum
int wgrad_small_f32_launch(const float* G, const float* X, int64_t M, int dq,
                           int64_t x_head_stride, float* dW, float* db, hipStream_t st);
// dH[q][c] (+)= dKp[q][:] . Wk[:][c] + dVp[q][:] . Wv[:][c]   per set (m = 16 rows, d = 128)
int kv_dh_launch(const float* dKp, const float* dVp, const float* Wk, const float* Wv, float* dH,
                 int B, int m, int d, int accumulate, hipStream_t st);

// fp32 weight -> bf16 image; mode 0 natural, 1 K-permuted, 2 transposed + K-permuted
int prep_weight(const float* src, __bf16* dst, int rows, int cols, int mode, hipStream_t st);

}  // namespace pca
