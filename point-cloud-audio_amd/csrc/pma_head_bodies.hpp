// The three per-set stages around the classifier of the train step, as device functions: the
// PMA forward epilogue, the classifier + cross-entropy (forward and backward) and the PMA backward
// epilogue.  Stand-alone they are the kernels k_mab0_epi / k_cls_fwd_bwd / k_mab0_epi_bwd; the
// engine runs them back to back inside ONE launch (k_pma_head, mab0_bwd_bf16.hip): each is a
// chain of dependent L2 round trips for one set, so two launches less is ~10 us of a 0.4 ms step.
#pragma once
#include "mab1_bf16.hpp"

#include <math.h>

namespace pca {

template <int MQ>
__device__ __forceinline__ void mab0_epi_body(const float* __restrict__ Tp,   // [B][S][R][dk]
                                                  const float* __restrict__ Mp,
                                                  const float* __restrict__ Lp, int S,
                                                  float* __restrict__ T,      // merged, saved
                                                  float* __restrict__ LSE,
                                                  const float* __restrict__ Qp,
                                                  const float* __restrict__ WvT,   // [dk][d]
                                                  const float* __restrict__ bv,
                                                  const float* __restrict__ WoT,   // [d][d]
                                                  const float* __restrict__ bo, int m, int d,
                                                  int dk, int h, float* __restrict__ H,
                                                  float* __restrict__ Osave,
                                                  float* __restrict__ Zsave, int bset) {
  extern __shared__ float sm[];
  float* sT = sm;                 // [h*m][dk]
  float* sO = sT + h * m * dk;    // [m][d]
  const int b = bset, tid = threadIdx.x;
  const int R = h * m, dh = d / h;
  if (S == 0) {                 // T already merged (layer-1 path)
    for (int i = tid; i < R * dk; i += 256) sT[i] = T[(int64_t)b * R * dk + i];
  } else {
    // merge the S point-range partials: T = sum_s f_s Tp_s / sum_s f_s Lp_s, f_s = 2^(M_s - M)
    for (int i = tid; i < R * dk; i += 256) {
      const int r = i / dk, c = i - r * dk;
      float M = -INFINITY;
      for (int s = 0; s < S; ++s) M = fmaxf(M, Mp[((int64_t)b * S + s) * R + r]);
      float L = 0.f, t = 0.f;
      for (int s = 0; s < S; ++s) {
        const float ms = Mp[((int64_t)b * S + s) * R + r];
        if (ms == -INFINITY) continue;
        const float fs = exp2f(ms - M);
        L += fs * Lp[((int64_t)b * S + s) * R + r];
        t += fs * Tp[(((int64_t)b * S + s) * R + r) * dk + c];
      }
      const float v = t / L;
      sT[i] = v;
      T[(int64_t)b * R * dk + i] = v;
      if (c == 0) LSE[(int64_t)b * R + r] = M + log2f(L);
    }
  }
  __syncthreads();
  const int f = tid % d, qh = tid / d;          // d == 128: two query halves
  if (MQ == 1 && m == 1) {
    // PMA (one seed): the second half of the workgroup would idle - it takes the second half
    // of every contraction instead (these GEMVs are chains of dependent L2 round trips)
    __shared__ float part[128];
    const int j = f / dh;
    float a1[1] = {qh == 0 ? Qp[f] + bv[f] : 0.f};
    col_gemm<1>(sT + j * dk + qh * (dk / 2), dk, WvT + (int64_t)qh * (dk / 2) * d, d, dk / 2, f, a1);
    if (qh == 1) part[f] = a1[0];
    __syncthreads();
    if (qh == 0) { a1[0] += part[f]; sO[f] = a1[0]; }
    __syncthreads();
    float z1[1] = {qh == 0 ? bo[f] : 0.f};
    col_gemm<1>(sO + qh * (d / 2), d, WoT + (int64_t)qh * (d / 2) * d, d, d / 2, f, z1);
    if (qh == 1) part[f] = z1[0];
    __syncthreads();
    if (qh == 0) {
      z1[0] += part[f];
      const float o1 = sO[f];
      const int64_t o = (int64_t)b * d + f;
      H[o] = o1 + fmaxf(z1[0], 0.f);
      if (Osave != nullptr) {
        Osave[o] = o1;
        Zsave[o] = z1[0];
      }
    }
    return;
  }
  const int q0 = qh * MQ;
  const bool act = q0 < m;
  float acc[MQ];
  if (act) {
    const int j = f / dh;
#pragma unroll
    for (int q = 0; q < MQ; ++q) acc[q] = (q0 + q < m) ? Qp[(q0 + q) * d + f] + bv[f] : 0.f;
    col_gemm<MQ>(sT + (j * m + q0) * dk, dk, WvT, d, dk, f, acc);
#pragma unroll
    for (int q = 0; q < MQ; ++q)
      if (q0 + q < m) sO[(q0 + q) * d + f] = acc[q];
  }
  __syncthreads();
  if (act) {
    float z[MQ];
#pragma unroll
    for (int q = 0; q < MQ; ++q) z[q] = bo[f];
    col_gemm<MQ>(sO + q0 * d, d, WoT, d, d, f, z);
#pragma unroll
    for (int q = 0; q < MQ; ++q) {
      if (q0 + q >= m) continue;
      const int64_t o = (int64_t)b * m * d + (q0 + q) * d + f;
      H[o] = acc[q] + fmaxf(z[q], 0.f);
      if (Osave != nullptr) {
        Osave[o] = acc[q];
        Zsave[o] = z[q];
      }
    }
  }
}

__device__ __forceinline__ void cls_fwd_bwd_body(
    const float* __restrict__ P, const float* __restrict__ Wc, const float* __restrict__ bc,
    const int64_t* __restrict__ labels, int B, int d, int C, float grad_scale,
    float* __restrict__ logits, float* __restrict__ dlogits, float* __restrict__ dP,
    float* __restrict__ lossv, float* __restrict__ corrv, int bset) {
  extern __shared__ float sm[];
  float* sP = sm;            // [d]
  float* sL = sP + d;        // [C] logits, then dlogits
  __shared__ float red[2];
  __shared__ int ramax;
  const int b = bset, tid = threadIdx.x, NT = blockDim.x;
  for (int f = tid; f < d; f += NT) sP[f] = P[(int64_t)b * d + f];
  __syncthreads();
  if (d % 128 == 0 && 4 * C <= NT) {
    // four lanes per class, each a quarter of the contraction: one round of 8 loads instead of
    // four (this stage sits in the middle of a per-set chain of dependent L2 round trips)
    const int c = tid >> 2, part = tid & 3, seg = d / 4;
    float acc = 0.f;
    if (c < C) {
      const float* w = Wc + (int64_t)c * d + part * seg;
      const float* x = sP + part * seg;
      for (int f = 0; f < seg; f += 32) {
        float4 w4[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) w4[u] = *reinterpret_cast<const float4*>(w + f + 4 * u);
#pragma unroll
        for (int u = 0; u < 8; ++u)
          acc += x[f + 4 * u] * w4[u].x + x[f + 4 * u + 1] * w4[u].y + x[f + 4 * u + 2] * w4[u].z +
                 x[f + 4 * u + 3] * w4[u].w;
      }
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    if (c < C && part == 0) {
      acc += bc[c];
      sL[c] = acc;
      logits[(int64_t)b * C + c] = acc;
    }
  } else {
  for (int c = tid; c < C; c += NT) {
    const float* w = Wc + (int64_t)c * d;
    float acc = bc[c];
    int f = 0;
    for (; f + 32 <= d; f += 32) {          // 8 independent 16-byte loads in flight
      float4 w4[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) w4[u] = *reinterpret_cast<const float4*>(w + f + 4 * u);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        acc += sP[f + 4 * u] * w4[u].x + sP[f + 4 * u + 1] * w4[u].y + sP[f + 4 * u + 2] * w4[u].z +
               sP[f + 4 * u + 3] * w4[u].w;
    }
    for (; f < d; f += 4) {
      const float4 w4 = *reinterpret_cast<const float4*>(w + f);
      acc += sP[f] * w4.x + sP[f + 1] * w4.y + sP[f + 2] * w4.z + sP[f + 3] * w4.w;
    }
    sL[c] = acc;
    logits[(int64_t)b * C + c] = acc;
  }
  }
  __syncthreads();
  if (tid < 64) {            // one wave: max / argmax / sum over the C logits
    float m = -INFINITY;
    int am = 0x7fffffff;
    for (int j = tid; j < C; j += 64)
      if (sL[j] > m) { m = sL[j]; am = j; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float om = __shfl_xor(m, o, 64);
      const int oa = __shfl_xor(am, o, 64);
      if (om > m || (om == m && oa < am)) { m = om; am = oa; }
    }
    float s = 0.f;
    for (int j = tid; j < C; j += 64) s += expf(sL[j] - m);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (tid == 0) { red[0] = m; red[1] = s; ramax = am; }
  }
  __syncthreads();
  const float m = red[0], s = red[1];
  const int64_t y = labels[b];
  const float gs = grad_scale / (float)B;
  if (tid == 0) {
    lossv[b] = m + logf(s) - sL[y];
    corrv[b] = ramax == (int)y ? 1.f : 0.f;
  }
  __syncthreads();
  for (int c = tid; c < C; c += NT) {
    const float g = (expf(sL[c] - m) / s - (c == y ? 1.f : 0.f)) * gs;
    sL[c] = g;
    dlogits[(int64_t)b * C + c] = g;
  }
  __syncthreads();
  if (NT == 2 * d) {
    // both halves of the workgroup: half the classes each, partial sums through LDS (sP is free)
    const int f = tid % d, half = tid / d;
    const int c0 = half * (C / 2), c1 = half ? C : C / 2;
    float acc = 0.f;
    int c = c0;
    for (; c + 13 <= c1; c += 13) {
      float wv[13];
#pragma unroll
      for (int u = 0; u < 13; ++u) wv[u] = Wc[(int64_t)(c + u) * d + f];
#pragma unroll
      for (int u = 0; u < 13; ++u) acc = fmaf(sL[c + u], wv[u], acc);
    }
    for (; c < c1; ++c) acc += sL[c] * Wc[(int64_t)c * d + f];
    if (half == 1) sP[f] = acc;
    __syncthreads();
    if (half == 0) dP[(int64_t)b * d + f] = acc + sP[f];
  } else {
  for (int f = tid; f < d; f += NT) {
    float acc = 0.f;
    int c = 0;
    for (; c + 10 <= C; c += 10) {          // 10 independent loads in flight
      float wv[10];
#pragma unroll
      for (int u = 0; u < 10; ++u) wv[u] = Wc[(int64_t)(c + u) * d + f];
#pragma unroll
      for (int u = 0; u < 10; ++u) acc = fmaf(sL[c + u], wv[u], acc);
    }
    for (; c < C; ++c) acc += sL[c] * Wc[(int64_t)c * d + f];
    dP[(int64_t)b * d + f] = acc;
  }
  }
}

template <int MQ>
__device__ __forceinline__ void mab0_epi_bwd_body(
    const float* __restrict__ dH, const float* __restrict__ Z, const float* __restrict__ T,
    const float* __restrict__ LSE, const float* __restrict__ Wo, const float* __restrict__ Wv,
    int m, int d, int dk, int h, int Rp, float* __restrict__ dZ, float* __restrict__ dO,
    float* __restrict__ Th,      // [h][B*m][dk] head-major copy of T (for dWv)
    float* __restrict__ dTf,     // [B][R][dk] fp32 (small-dk path)
    __bf16* __restrict__ dTb,    // [B][Rp][dk] natural rows
    __bf16* __restrict__ dTt,    // [B][dk][Rp] r-permuted
    float* __restrict__ Delta,   // [B][Rp]
    float* __restrict__ LSEp,    // [B][Rp] padded with +1e30
    int B, float* __restrict__ zero_ptr, int zero_n, int bset) {   // optional: clears the DG accumulator
  extern __shared__ float sm[];
  if (zero_ptr != nullptr)
    for (int i = bset * 256 + threadIdx.x; i < zero_n; i += gridDim.x * 256)
      zero_ptr[i] = 0.f;
  float* sdZ = sm;               // [m][d]
  float* sdO = sdZ + m * d;      // [m][d]
  float* sDl = sdO + m * d;      // [Rp] partial Delta
  const int b = bset, tid = threadIdx.x;
  const int R = h * m, dh = d / h;
  for (int o = tid; o < m * d; o += 256) {
    const float g = dH[(int64_t)b * m * d + o];
    const float z = Z[(int64_t)b * m * d + o];
    const float v = z > 0.f ? g : 0.f;
    sdZ[o] = v;
    dZ[(int64_t)b * m * d + o] = v;
  }
  for (int i = tid; i < Rp; i += 256) sDl[i] = 0.f;
  __syncthreads();
  const int c = tid % d;
  // PMA (one seed): the second half of the workgroup takes the second half of the
  // contraction over fc_o and the heads 2, 3 of the dT products instead of idling
  const bool pma = MQ == 1 && m == 1 && (h % 2) == 0;
  const int half = tid / d;
  int q0 = half * MQ;
  bool act = q0 < m;
  if (pma) {
    float* part = sDl + Rp;                 // [d] scratch behind the Delta slots
    float a1[1] = {half == 0 ? dH[(int64_t)b * d + c] : 0.f};
    col_gemm<1>(sdZ + half * (d / 2), d, Wo + (int64_t)half * (d / 2) * d, d, d / 2, c, a1);
    if (half == 1) part[c] = a1[0];
    __syncthreads();
    if (half == 0) {
      a1[0] += part[c];
      sdO[c] = a1[0];
      dO[(int64_t)b * d + c] = a1[0];
    }
    q0 = 0;
    act = true;
  } else if (act) {
    float acc[MQ];
#pragma unroll
    for (int q = 0; q < MQ; ++q)
      acc[q] = (q0 + q < m) ? dH[(int64_t)b * m * d + (q0 + q) * d + c] : 0.f;
    col_gemm<MQ>(sdZ + q0 * d, d, Wo, d, d, c, acc);
#pragma unroll
    for (int q = 0; q < MQ; ++q)
      if (q0 + q < m) {
        sdO[(q0 + q) * d + c] = acc[q];
        dO[(int64_t)b * m * d + (q0 + q) * d + c] = acc[q];
      }
  }
  __syncthreads();
  // dT[j m + q][cc] = sum_f dO[q][j dh + f] Wv[j dh + f][cc] ; thread owns column cc of dk
  const int j_lo = pma ? half * (h / 2) : 0, j_hi = pma ? j_lo + h / 2 : h;
  for (int cc = tid % d; cc < dk && act; cc += d) {
    for (int j = j_lo; j < j_hi; ++j) {
      float acc[MQ];
#pragma unroll
      for (int q = 0; q < MQ; ++q) acc[q] = 0.f;
      col_gemm<MQ>(sdO + q0 * d + j * dh, d, Wv + (int64_t)j * dh * dk, dk, dh, cc, acc);
#pragma unroll
      for (int q = 0; q < MQ; ++q) {
        if (q0 + q >= m) continue;
        const int r = j * m + q0 + q;
        const float tv = T[((int64_t)b * R + r) * dk + cc];
        Th[((int64_t)j * B * m + (int64_t)b * m + q0 + q) * dk + cc] = tv;
        atomicAdd(&sDl[r], acc[q] * tv);
        if (dTf != nullptr) dTf[((int64_t)b * R + r) * dk + cc] = acc[q];
        if (dTb != nullptr) {
          dTb[((int64_t)b * Rp + r) * dk + cc] = (__bf16)acc[q];
          const int rb32 = r & ~31, ro = r & 31;
          int pos = 0;
#pragma unroll
          for (int p = 0; p < 32; ++p)
            if (perm32(p) == ro) pos = p;
          dTt[((int64_t)b * dk + cc) * Rp + rb32 + pos] = (__bf16)acc[q];
        }
      }
    }
  }
  __syncthreads();
  for (int r = tid; r < Rp; r += 256) {
    Delta[(int64_t)b * Rp + r] = r < R ? sDl[r] : 0.f;
    LSEp[(int64_t)b * Rp + r] = r < R ? LSE[(int64_t)b * R + r] : 1.0e30f;
  }
  if (dTb != nullptr) {          // zero the padding rows / columns of the bf16 images
    for (int o = tid; o < (Rp - R) * dk; o += 256) {
      const int r = R + o / dk, cc = o % dk;
      dTb[((int64_t)b * Rp + r) * dk + cc] = (__bf16)0.f;
      const int rb32 = r & ~31, ro = r & 31;
      int pos = 0;
#pragma unroll
      for (int p = 0; p < 32; ++p)
        if (perm32(p) == ro) pos = p;
      dTt[((int64_t)b * dk + cc) * Rp + rb32 + pos] = (__bf16)0.f;
    }
  }
}

}  // namespace pca
