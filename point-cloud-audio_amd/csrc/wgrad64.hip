// Weight and bias gradient of a 64 -> 64 Linear over a tall activation, one launch:
//   dW[64][64] += dY[M][64]^T X[M][64],  db[64] += colsum(dY)
// (the adjoint of fc_q / fc_k / fc_v / fc_o, set_transformer-master/modules.py:13-16, for the shipped
// d = 64 models, Code/models.py:34-44 with dim_hidden = 64).  The generic split-K GEMM reads the
// two operands along their strided index (62 us at M = 131 200, 1.1 TB/s) and the column sum reads dY
// again; here both are read ONCE, along their rows.
//
// A wave takes 16 rows at a time: lane (og = lane & 15, kg = lane >> 4) loads columns 4 og .. 4 og + 3
// of rows 4 kg .. 4 kg + 3 of both operands (float4: every 256-byte row is one coalesced segment), and
// the 4 x 4 block it then holds IS the operand of v_mfma_f32_16x16x16_bf16 for four row-permuted
// products: A_j = dY[4 kg + e][4 og + j] (e = the k slot), B_j' likewise from X, so that
// acc[j][j'][e] = dW[16 kg + 4 e + j][4 og + j'] -- no transpose through LDS.  fp32 operands are rounded
// to bf16 (RNE) as the generic bf16-operand GEMM does, accumulation is fp32, the bias sum is fp32 of the
// unrounded dY.  The next group's rows are loaded under the current group's products.  Each wave
// leaves its block in an LDS slab of its own (float atomics in LDS measured 4 x slower), the workgroup
// sums the slabs and adds the result with coalesced atomics (as the split-K GEMM it replaces).
#include "pca_common.h"

#include <stdint.h>
#include "mab1_bf16.hpp"
#include "mfma_common.hpp"

namespace pca {

namespace {

// clamped, unconditional, and NOT zeroed here: a select behind the load would wait for it where it is
// issued (the rows of the next group are requested a trip ahead); rows past the end are zeroed by the one
// ragged trip that has them
__device__ __forceinline__ float4 ld_row(const float* __restrict__ P, int64_t r, int64_t M, int c) {
  const int64_t rc = r < M ? r : M - 1;
  return *reinterpret_cast<const float4*>(P + rc * 64 + c);
}
__device__ __forceinline__ bf16x4 col_of(const float4 (&t)[4], int j) {
  bf16x4 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float v = j == 0 ? t[e].x : j == 1 ? t[e].y : j == 2 ? t[e].z : t[e].w;
    r[e] = (__bf16)v;
  }
  return r;
}

constexpr int WG_WAVES = 16;       // waves per workgroup
constexpr int WG_SLABS = 8;        // LDS slabs: waves w and w + 8 share one (two rounds)

__global__ __launch_bounds__(64 * WG_WAVES) void k_wgrad64(const float* __restrict__ dY,
                                                           const float* __restrict__ X,
                                                           float* __restrict__ dW,
                                                           float* __restrict__ db, int64_t M) {
  extern __shared__ float sW[];                 // [WG_SLABS][64 * 64 + 64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int og = lane & 15, kg = lane >> 4;

  f32x4 acc[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[j][k] = f32x4{0.f, 0.f, 0.f, 0.f};
  float4 bs = make_float4(0.f, 0.f, 0.f, 0.f);

  const int64_t groups = (M + 15) / 16;
  const int64_t nw = (int64_t)gridDim.x * WG_WAVES;
  int64_t g = (int64_t)blockIdx.x * WG_WAVES + wave;
  float4 y[4], x[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    y[e] = ld_row(dY, g * 16 + 4 * kg + e, M, 4 * og);
    x[e] = ld_row(X, g * 16 + 4 * kg + e, M, 4 * og);
  }
  for (; g < groups; g += nw) {
    if (g * 16 + 16 > M) {                        // (uniform) the ragged last group
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (g * 16 + 4 * kg + e >= M) {
          y[e] = make_float4(0.f, 0.f, 0.f, 0.f);
          x[e] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      bs.x += y[e].x; bs.y += y[e].y; bs.z += y[e].z; bs.w += y[e].w;
    }
    bf16x4 a[4], b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { a[j] = col_of(y, j); b[j] = col_of(x, j); }
    // the next group's rows into the registers the conversions just freed, in flight under the products
    // (a second register set does not fit beside the 64 accumulators at 16 waves per workgroup)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      y[e] = ld_row(dY, (g + nw) * 16 + 4 * kg + e, M, 4 * og);
      x[e] = ld_row(X, (g + nw) * 16 + 4 * kg + e, M, 4 * og);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[j][k] = mfma16(a[j], b[k], acc[j][k]);
  }

  bs.x += __shfl_xor(bs.x, 16); bs.y += __shfl_xor(bs.y, 16);
  bs.z += __shfl_xor(bs.z, 16); bs.w += __shfl_xor(bs.w, 16);
  bs.x += __shfl_xor(bs.x, 32); bs.y += __shfl_xor(bs.y, 32);
  bs.z += __shfl_xor(bs.z, 32); bs.w += __shfl_xor(bs.w, 32);
  // waves 0..7 store their blocks, waves 8..15 add theirs on top (each lane its own elements)
  float* slab = sW + (wave & (WG_SLABS - 1)) * (64 * 64 + 64);
#pragma unroll
  for (int round = 0; round < WG_WAVES / WG_SLABS; ++round) {
    if (wave / WG_SLABS == round) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int o = 16 * kg + 4 * e + j;
          float4* dst = reinterpret_cast<float4*>(slab + o * 64 + 4 * og);
          float4 v = make_float4(acc[j][0][e], acc[j][1][e], acc[j][2][e], acc[j][3][e]);
          if (round > 0) { const float4 p = *dst; v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w; }
          *dst = v;
        }
      if (kg == 0) {
        float4* dst = reinterpret_cast<float4*>(slab + 64 * 64 + 4 * og);
        float4 v = bs;
        if (round > 0) { const float4 p = *dst; v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w; }
        *dst = v;
      }
    }
    __syncthreads();
  }
  // (few, wide workgroups: atomics of different workgroups onto one cache line serialise, ~25 ns each -
  // 256 workgroups x 130 lines cost ~6 us per call)
  for (int i = tid; i < 64 * 64 + (db != nullptr ? 64 : 0); i += 64 * WG_WAVES) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < WG_SLABS; ++w) v += sW[w * (64 * 64 + 64) + i];
    atomicAdd(i < 64 * 64 ? dW + i : db + (i - 64 * 64), v);
  }
}

// Narrow input (layer 1: the points have DQ <= 4 coordinates): dW[64][DQ] += dY[M][64]^T X[M][DQ],
// db += colsum(dY), all fp32 on the vector ALU -- dY is the only stream (256 B per row), X rides in L1.
// Lane (og, kg) of a wave takes columns 4 og .. 4 og + 3 of four rows of the wave's 16-row group.
constexpr int NWN = 16;            // waves per workgroup of the narrow kernel

template <int DQ>
__global__ __launch_bounds__(64 * NWN) void k_wgrad64_narrow(const float* __restrict__ dY,
                                                        const float* __restrict__ X,
                                                        float* __restrict__ dW,
                                                        float* __restrict__ db, int64_t M) {
  __shared__ float sP[NWN][64 * (DQ + 1)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int og = lane & 15, kg = lane >> 4;
  float acc[4][DQ];
  float bs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int q = 0; q < DQ; ++q) acc[c][q] = 0.f;
  // a wave takes 16 consecutive rows per trip (lane (og, kg): rows 4 kg .. 4 kg + 3, columns 4 og ..), the
  // next group's rows requested under the current one's arithmetic - k_wgrad64's stream (two groups per
  // trip do not fit the 128 registers of a 16-wave workgroup)
  const int64_t groups = (M + 15) / 16;
  const int64_t nw = (int64_t)gridDim.x * NWN;
  int64_t g = (int64_t)blockIdx.x * NWN + wave;
  float4 y[4];
  float x[4][DQ];
  auto rows_of = [&](int64_t gg, float4 (&yy)[4], float (&xx)[4][DQ]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t r = gg * 16 + 4 * kg + e;
      const int64_t rc = r < M ? r : M - 1;                     // clamped; zeroed in the ragged trip
      yy[e] = *reinterpret_cast<const float4*>(dY + rc * 64 + 4 * og);
#pragma unroll
      for (int q = 0; q < DQ; ++q) xx[e][q] = X[rc * DQ + q];
    }
  };
  rows_of(g, y, x);
  for (; g < groups; g += nw) {
    float4 yn[4];
    float xn[4][DQ];
    rows_of(g + nw, yn, xn);
    if (g * 16 + 16 > M) {                                      // (uniform) the ragged last group
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (g * 16 + 4 * kg + e >= M) y[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float yv[4] = {y[e].x, y[e].y, y[e].z, y[e].w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        bs[c] += yv[c];
#pragma unroll
        for (int q = 0; q < DQ; ++q) acc[c][q] = fmaf(yv[c], x[e][q], acc[c][q]);
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      y[e] = yn[e];
#pragma unroll
      for (int q = 0; q < DQ; ++q) x[e][q] = xn[e][q];
    }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
#pragma unroll
    for (int q = 0; q < DQ; ++q) {
      acc[c][q] += __shfl_xor(acc[c][q], 16);
      acc[c][q] += __shfl_xor(acc[c][q], 32);
    }
    bs[c] += __shfl_xor(bs[c], 16);
    bs[c] += __shfl_xor(bs[c], 32);
  }
  if (kg == 0) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
      for (int q = 0; q < DQ; ++q) sP[wave][(4 * og + c) * DQ + q] = acc[c][q];
      sP[wave][64 * DQ + 4 * og + c] = bs[c];
    }
  }
  __syncthreads();
  // few, wide workgroups on purpose: the 64 DQ + 64 results are a handful of cache lines, and atomics of
  // different workgroups onto ONE line serialise at ~25 ns each - 512 workgroups cost 13 us here
  for (int i = tid; i < 64 * DQ + (db != nullptr ? 64 : 0); i += 64 * NWN) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < NWN; ++w) v += sP[w][i];
    atomicAdd(i < 64 * DQ ? dW + i : db + (i - 64 * DQ), v);
  }
}

}  // namespace

bool wgrad64_ok(const float* dY, const float* X, int64_t M, int din, int dout) {
  if (dout == 64 && din >= 1 && din <= 4 && M >= 1)
    return (reinterpret_cast<uintptr_t>(dY) & 15) == 0;
  return din == 64 && dout == 64 && M >= 1 && ((reinterpret_cast<uintptr_t>(dY) | reinterpret_cast<uintptr_t>(X)) & 15) == 0;
}

// dW += dY^T X and (db != nullptr) db += colsum(dY); both outputs are accumulated into
int wgrad64(const float* dY, const float* X, float* dW, float* db, int64_t M, int din,
            hipStream_t st) {
  PCA_REQUIRE(dY && X && dW && M > 0 && (din == 64 || (din >= 1 && din <= 4)),
              "wgrad64: bad arguments");
  if (din <= 4) {
    int64_t wgs = cdiv(cdiv(M, 16), NWN);       // a 16-row group per wave before anyone gets a second
    if (wgs > 128) wgs = 128;                   // (see the kernel's last loop)
#define PCA_NARROW(Q)                                                                          \
  case Q:                                                                                      \
    hipLaunchKernelGGL((k_wgrad64_narrow<Q>), dim3((unsigned)wgs), dim3(64 * NWN), 0, st, dY, X, dW, \
                       db, M);                                                                 \
    break;
    switch (din) {
      PCA_NARROW(1) PCA_NARROW(2) PCA_NARROW(3) PCA_NARROW(4)
    }
#undef PCA_NARROW
    return check_launch("k_wgrad64_narrow");
  }
  const int64_t groups = (M + 15) / 16;
  // one workgroup per CU at most (its 8 slabs take 130 KB of LDS); 128 x 4160 atomics per call
  int64_t wgs = cdiv(groups, WG_WAVES);
  if (wgs > 128) wgs = 128;
  if (wgs < 1) wgs = 1;
  static const int once = [] {
    return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(k_wgrad64),
                                    hipFuncAttributeMaxDynamicSharedMemorySize,
                                    WG_SLABS * (64 * 64 + 64) * (int)sizeof(float));
  }();
  PCA_REQUIRE(once == 0, "wgrad64: cannot reserve %d bytes of LDS",
              WG_SLABS * (64 * 64 + 64) * (int)sizeof(float));
  hipLaunchKernelGGL(k_wgrad64, dim3((unsigned)wgs), dim3(64 * WG_WAVES),
                     WG_SLABS * (64 * 64 + 64) * sizeof(float), st, dY, X, dW, db, M);
  return check_launch("k_wgrad64");
}

}  // namespace pca
