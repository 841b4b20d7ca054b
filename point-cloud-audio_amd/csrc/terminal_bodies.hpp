// Terminal reductions that are small enough to share ONE launch at the end of a backward phase
// (k_terminal1 in mab0_bwd_bf16.hip): the classifier's weight gradient + loss counters and the
// layer-1 fc_v gradient, next to the shared-query gradients.  The bodies are device functions
// so that the stand-alone kernels (k_cls_wgrad, k_wgrad_small) stay thin wrappers.
#pragma once
#include "mab1_bf16.hpp"

namespace pca {

// dWc += dlogits^T P, dbc += colsum(dlogits); class c = one workgroup (any blockDim <= 256);
// the workgroup of class 0 also reduces the per-set loss / accuracy counters
__device__ __forceinline__ void cls_wgrad_body(
    const float* __restrict__ dlogits, const float* __restrict__ P,
    const float* __restrict__ lossv, const float* __restrict__ corrv, int B, int d, int C,
    float* __restrict__ dWc, float* __restrict__ dbc, float* __restrict__ loss_out,
    float* __restrict__ stats, int c) {
  const int tid = threadIdx.x, NT = blockDim.x;
  // this class's column of dlogits goes to LDS in one round trip (chunks of 1024 sets); the
  // P loads are then the only global traffic of the reduction, 16 in flight
  __shared__ float sg[1024];
  for (int f0 = 0; f0 < d; f0 += NT) {
    const int f = f0 + tid;
    float acc = 0.f;
    for (int b0 = 0; b0 < B; b0 += 1024) {
      const int nb = (B - b0 < 1024) ? B - b0 : 1024;
      __syncthreads();
      for (int i = tid; i < nb; i += NT) sg[i] = dlogits[(int64_t)(b0 + i) * C + c];
      __syncthreads();
      if (f < d) {
        int bb = 0;
        for (; bb + 64 <= nb; bb += 64) {        // (a few dependent L2 round trips and nothing else)
          float pv[64];
#pragma unroll
          for (int u = 0; u < 64; ++u) pv[u] = P[(int64_t)(b0 + bb + u) * d + f];
#pragma unroll
          for (int u = 0; u < 64; ++u) acc = fmaf(sg[bb + u], pv[u], acc);
        }
        for (; bb + 16 <= nb; bb += 16) {
          float pv[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) pv[u] = P[(int64_t)(b0 + bb + u) * d + f];
#pragma unroll
          for (int u = 0; u < 16; ++u) acc = fmaf(sg[bb + u], pv[u], acc);
        }
        for (; bb < nb; ++bb) acc = fmaf(sg[bb], P[(int64_t)(b0 + bb) * d + f], acc);
      }
    }
    if (f < d) dWc[(int64_t)c * d + f] += acc;
  }
  __shared__ float red[256];
  float part = 0.f;
  for (int bb = tid; bb < B; bb += NT) part += dlogits[(int64_t)bb * C + c];
  red[tid] = part;
  __syncthreads();
  if (tid == 0) {
    float t = 0.f;
    for (int i = 0; i < NT; ++i) t += red[i];
    dbc[c] += t;
  }
  if (c == 0) {              // loss / accuracy counters, summed in a fixed order
    __syncthreads();
    float l = 0.f, k = 0.f;
    for (int bb = tid; bb < B; bb += NT) { l += lossv[bb]; k += corrv[bb]; }
    red[tid] = l;
    __syncthreads();
    float lt = 0.f;
    if (tid == 0) for (int i = 0; i < NT; ++i) lt += red[i];
    __syncthreads();
    red[tid] = k;
    __syncthreads();
    if (tid == 0) {
      float kt = 0.f;
      for (int i = 0; i < NT; ++i) kt += red[i];
      loss_out[0] = lt / (float)B;
      if (stats != nullptr) { stats[0] += lt; stats[1] += kt; }
    }
  }
}

// dW[128 x dq] += G[M x 128]^T . X_h[M x dq] (dq <= 4), db += colsum(G); 256 threads
template <typename GT>
__device__ __forceinline__ void wgrad_small_body(const GT* __restrict__ G,
                                                 const float* __restrict__ X, int64_t M, int dq,
                                                 int rows_per_wg, int64_t x_head_stride,
                                                 float* __restrict__ dW, float* __restrict__ db,
                                                 int blk, float* __restrict__ slab = nullptr) {
  constexpr int D = 128;
  __shared__ float red[128][5];
  const int f = threadIdx.x & 127, ph = threadIdx.x >> 7;
  const float* __restrict__ Xh = X + (int64_t)(f >> 5) * x_head_stride;
  const int64_t r0 = (int64_t)blk * rows_per_wg;
  const int64_t r1 = (r0 + rows_per_wg < M) ? r0 + rows_per_wg : M;
  float acc[4] = {0.f, 0.f, 0.f, 0.f}, bs = 0.f;
  // batches of 16 rows: 16 independent G loads (+ the broadcast X rows) in flight; rows beyond the
  // range are read from the last valid row (unconditional loads) and zeroed
  for (int64_t row = r0 + ph; row < r1; row += 32) {
    float gv[16], xv[16][4];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int64_t rr = row + 2 * u;
      const bool ok = rr < r1;
      const int64_t rc = ok ? rr : r1 - 1;
      const float gq = (float)G[rc * D + f];
      gv[u] = ok ? gq : 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) xv[u][c] = c < dq ? Xh[rc * dq + c] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      bs += gv[u];
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] = fmaf(gv[u], xv[u][c], acc[c]);
    }
  }
  if (ph == 1) {
#pragma unroll
    for (int c = 0; c < 4; ++c) red[f][c] = acc[c];
    red[f][4] = bs;
  }
  __syncthreads();
  if (ph == 0 && slab != nullptr) {       // this workgroup's partial (summed by rider rows later)
    float* out = slab + (int64_t)blk * (D * dq + D);
    for (int c = 0; c < dq; ++c) out[f * dq + c] = acc[c] + red[f][c];
    out[D * dq + f] = bs + red[f][4];
  } else if (ph == 0) {
    for (int c = 0; c < dq; ++c) atomicAdd(&dW[f * dq + c], acc[c] + red[f][c]);
    if (db != nullptr) atomicAdd(&db[f], bs + red[f][4]);
  }
}

}  // namespace pca
