// Fused attention core for SMALL head dimensions (d / h <= 16: the reference's own hyper-parameters,
// d = 64, 8 heads of dim 8, m = 64, N = 1025 / 5120 - Code/settransformer.py:81-83,
// Code/settransformertemp.py:95-97) of the bf16-operand GEMM chain (mab_f32.hip with PCA_MODE_BF16 on a
// shape that has no fully fused block kernels):
//
//     A = softmax(Q_ K_^T / sqrt(dim_V)) ; O = Q_ + A V_          set_transformer-master/modules.py:28-29
//
// and its adjoint, WITHOUT the [B h, nq, nk] matrix A.  The chain materialises A (269 MB fp32 per block
// at the shipped FST shape, B = 128) and passes it through memory eight times per block and step -
// QK^T out, softmax in / out, AV in; A^T dO, dA out, softmax adjoint in / in / out, dS K, dS^T Q -
// which is 5.4 of the 7.2 ms of an FST training step (profiles/r03_bf16_fst_bench.json).  Here:
//   k_attnc_fwd     S^T = K Q^T on the MFMA (16x16x16, the head dim sitting in the low k slots), online
//                   softmax over 64 keys at a time, O^T += V^T P^T; saves the log-sum-exp (log2 domain)
//                   instead of A
//   k_attnc_bwd_q   P^T recomputed from the LSE, dP^T = V dO^T, dS^T = P^T (dP^T - delta) / sqrt d,
//                   dQ^T += K^T dS^T ; delta = rowdot(dO, O - Q_)
//   k_attnc_bwd_kv  the same tiles in the other orientation (queries on the accumulator rows),
//                   dV^T += dO^T P, dK^T += Q^T dS - every element of dK / dV is written once, no
//                   atomics, no zero-fill
// Work split: a workgroup owns a set and a few 16-row tiles of the side that stays in registers (queries
// in the first two kernels, keys in the third); its WAVES ARE THE HEADS.  The other side streams through
// LDS in chunks, staged ONCE per workgroup for all heads: whole [rows][d] fp32 rows are read coalesced,
// rounded to bf16 (as the chain's k_gemm_bf16 rounds its operands) and written in the two forms the MFMA
// operands want - row form [row][d + 4] (lane (r, g) reads features j dh + 4 g .. + 3 of row r: 8 bytes)
// and transposed form [feature][rows + 4] (lane (r, g) reads rows 4 g .. 4 g + 3 of feature j dh + r).
// A first version with one wave per (set, head, tile) reading its operands from global memory spent
// 146 vector instructions per 16 x 16 tile, 584 issue cycles: 75 us per block, all of it address
// arithmetic and conversions repeated by every wave; this one spends about 25.
// The streamed operand needs NO zero padding beyond the head dim: where it is the MFMA's A operand of a
// product with the register side, the register side's k slots >= dh are zero (the neighbouring head's
// finite features are multiplied by 0); where it produces output rows >= dh, those rows are not stored.
// Softmax statistics, accumulation and all tensors in memory stay fp32.  Queries shared by all sets
// (ISAB's I, PMA's S: modules.py:52,63) are read through a zero batch stride.  Keys at or beyond
// k_lengths[b] are masked (P = 0), as softmax_rows does for the chain.
// Workgroup ids: sets b and b + 8 k sit on the same XCD (id mod 8), the workgroups of one set in
// consecutive slots of it - what they stage comes from that XCD's L2 after the first.
#include "mab1_bf16.hpp"

#include <math.h>

namespace pca {

namespace {

struct AttnCoreArgs {
  const float *Qp, *Kp, *Vp;   // [Bq][nq][d], [B][nk][d], [B][nk][d]
  const float* O;              // [B][nq][d]   (backward: the forward's output)
  const float* dO;             // [B][nq][d]
  float* Oout;                 // forward
  float* LSE;                  // [B][h][nq]  log2 domain: m + log2(l)
  float* Delta;                // [B][h][nq]
  float *dQp, *dKp, *dVp;
  const int32_t* lengths;
  int B, nq, nk, d, dh, h;
  int gx;                      // workgroups per set
  int64_t qb;                  // batch stride of Qp (0: shared)
  float scale, c;              // 1 / sqrt(d), scale * log2(e)
};

constexpr int CH = 128;        // streamed rows per chunk: keys of k_attnc_fwd / k_attnc_bwd_q
constexpr int CHQ = 64;        //                          queries of k_attnc_bwd_kv
// bf16 elements of one staged chunk: row form has one spare row (the last head's upper lanes read on
// into the next row), transposed form 16 spare feature rows
__host__ __device__ constexpr int row_elems(int ch, int d) { return (ch + 1) * (d + 4); }
__host__ __device__ constexpr int tr_elems(int ch, int d) { return (d + 16) * (ch + 4); }

// workgroup id -> (set, index within the set): see the header
__device__ __forceinline__ void wg_coords(int gx, int& b, int& x) {
  const int id = blockIdx.x, slot = id >> 3;
  x = slot % gx;
  b = (slot / gx) * 8 + (id & 7);
}

// 4 consecutive fp32 features (f0 .. f0 + 3) of one row as a bf16 MFMA operand; zeros beyond the head dim
// (unconditional clamped load, value selected afterwards: DESIGN.md 4.5)
__device__ __forceinline__ bf16x4 row4(const float* row, int f0, int dh) {
  const bool on = f0 < dh;
  const float4 x = *reinterpret_cast<const float4*>(row + (on ? f0 : 0));
  bf16x4 v;
  v[0] = (__bf16)(on ? x.x : 0.f); v[1] = (__bf16)(on ? x.y : 0.f);
  v[2] = (__bf16)(on ? x.z : 0.f); v[3] = (__bf16)(on ? x.w : 0.f);
  return v;
}

// rows r0 .. r0 + CHN - 1 of X[nrows][d] (fp32) -> bf16 in LDS, all heads at once; rows >= nrows are zeros
template <int CHN, bool ROW, bool TR>
__device__ __forceinline__ void stage(const float* __restrict__ X, int nrows, int r0, int d,
                                      __bf16* __restrict__ Rw, __bf16* __restrict__ Tr, int tid,
                                      int nthr) {
  const int cg = d >> 2;
  for (int i = tid; i < (CHN / 4) * cg; i += nthr) {
    const int q = i / cg, c4 = i - q * cg;
    float4 v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = r0 + 4 * q + e;
      const int rc = row < nrows ? row : nrows - 1;
      v[e] = *reinterpret_cast<const float4*>(X + (int64_t)rc * d + 4 * c4);
      if (row >= nrows) v[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if constexpr (ROW) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        bf16x4 t;
        t[0] = (__bf16)v[e].x; t[1] = (__bf16)v[e].y; t[2] = (__bf16)v[e].z; t[3] = (__bf16)v[e].w;
        *reinterpret_cast<bf16x4*>(Rw + (4 * q + e) * (d + 4) + 4 * c4) = t;
      }
    }
    if constexpr (TR) {
      bf16x4 t0, t1, t2, t3;
      t0[0] = (__bf16)v[0].x; t0[1] = (__bf16)v[1].x; t0[2] = (__bf16)v[2].x; t0[3] = (__bf16)v[3].x;
      t1[0] = (__bf16)v[0].y; t1[1] = (__bf16)v[1].y; t1[2] = (__bf16)v[2].y; t1[3] = (__bf16)v[3].y;
      t2[0] = (__bf16)v[0].z; t2[1] = (__bf16)v[1].z; t2[2] = (__bf16)v[2].z; t2[3] = (__bf16)v[3].z;
      t3[0] = (__bf16)v[0].w; t3[1] = (__bf16)v[1].w; t3[2] = (__bf16)v[2].w; t3[3] = (__bf16)v[3].w;
      __bf16* dst = Tr + (4 * c4) * (CHN + 4) + 4 * q;
      *reinterpret_cast<bf16x4*>(dst) = t0;
      *reinterpret_cast<bf16x4*>(dst + (CHN + 4)) = t1;
      *reinterpret_cast<bf16x4*>(dst + 2 * (CHN + 4)) = t2;
      *reinterpret_cast<bf16x4*>(dst + 3 * (CHN + 4)) = t3;
    }
  }
}
__device__ __forceinline__ void lds_clear(__bf16* p, int elems, int tid, int nthr) {
  uint32_t* w = reinterpret_cast<uint32_t*>(p);
  for (int i = tid; i < (elems + 1) / 2; i += nthr) w[i] = 0u;
}
__device__ __forceinline__ float max2(float a, float b) {     // finite or -inf operands, no NaN
  return __builtin_amdgcn_fmed3f(a, b, INFINITY);
}

// 64 keys of one 16-query tile: scores, running max / sum, O^T += V^T P^T
template <bool MASK>
__device__ __forceinline__ void fwd_keys64(const bf16x4 (&ka)[4], const bf16x4 (&va)[4], bf16x4 qb4,
                                           float c, int klim, int g, float& m, float& l, f32x4& acc) {
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 s[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) s[u] = mfma16(ka[u], qb4, z4);        // [key 16 u + 4 g + e][query r]
  if constexpr (MASK) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (16 * u + 4 * g + e >= klim) s[u][e] = -INFINITY;
  }
  float mt = max2(max2(s[0][0], s[0][1]), max2(s[0][2], s[0][3]));
#pragma unroll
  for (int u = 1; u < 4; ++u) mt = max2(mt, max2(max2(s[u][0], s[u][1]), max2(s[u][2], s[u][3])));
  mt = wave16_max(mt);
  const float mn = max2(m, mt * c);                    // finite: the 64 keys hold at least one live key
  const float alpha = __builtin_amdgcn_exp2f(m - mn);
  float ls = 0.f;
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      s[u][e] = __builtin_amdgcn_exp2f(fmaf(s[u][e], c, -mn));
      ls += s[u][e];
    }
  l = fmaf(l, alpha, ls);                              // this lane's share; summed over the lanes at the end
  m = mn;
#pragma unroll
  for (int e = 0; e < 4; ++e) acc[e] *= alpha;        // the accumulator's column is this lane's query
#pragma unroll
  for (int u = 0; u < 4; ++u) acc = mfma16(va[u], pack4(s[u]), acc);   // [feature 4 g + e][query r]
}

template <int QT>
__global__ __launch_bounds__(1024) void k_attnc_fwd(const AttnCoreArgs a) {
  extern __shared__ __attribute__((aligned(16))) __bf16 lds[];
  const int tid = threadIdx.x, lane = tid & 63, j = tid >> 6, r = lane & 15, g = lane >> 4;
  const int nthr = blockDim.x, d = a.d, dh = a.dh, nq = a.nq, nk = a.nk;
  int b, x;
  wg_coords(a.gx, b, x);
  if (b >= a.B) return;
  __bf16* Krow = lds;
  __bf16* Vtr = lds + row_elems(CH, d);
  lds_clear(lds, row_elems(CH, d) + tr_elems(CH, d), tid, nthr);
  int len = nk;
  if (a.lengths != nullptr) len = a.lengths[b] < nk ? a.lengths[b] : nk;
  const int t0 = x * QT;                               // first query tile
  bf16x4 qb4[QT];
  float m[QT], l[QT];
  f32x4 acc[QT];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    const int q0 = (t0 + t) * 16;
    const int qi = q0 + r < nq ? q0 + r : nq - 1;
    qb4[t] = row4(a.Qp + (int64_t)b * a.qb + (int64_t)qi * d + j * dh, 4 * g, dh);
    m[t] = -INFINITY; l[t] = 0.f;
    acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const float* Kb = a.Kp + (int64_t)b * nk * d;
  const float* Vb = a.Vp + (int64_t)b * nk * d;
  for (int c0 = 0; c0 < len; c0 += CH) {
    __syncthreads();                                   // the previous chunk is consumed (first: cleared)
    stage<CH, true, false>(Kb, nk, c0, d, Krow, nullptr, tid, nthr);
    stage<CH, false, true>(Vb, nk, c0, d, nullptr, Vtr, tid, nthr);
    __syncthreads();
    const int cn = len - c0 < CH ? len - c0 : CH;
    for (int k0 = 0; k0 < cn; k0 += 64) {
      bf16x4 ka[4], va[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        ka[u] = *reinterpret_cast<const bf16x4*>(Krow + (k0 + 16 * u + r) * (d + 4) + j * dh + 4 * g);
        va[u] = *reinterpret_cast<const bf16x4*>(Vtr + (j * dh + r) * (CH + 4) + k0 + 16 * u + 4 * g);
      }
      const bool full = k0 + 64 <= cn;
#pragma unroll
      for (int t = 0; t < QT; ++t) {
        if ((t0 + t) * 16 >= nq) break;
        if (full) fwd_keys64<false>(ka, va, qb4[t], a.c, 64, g, m[t], l[t], acc[t]);
        else      fwd_keys64<true>(ka, va, qb4[t], a.c, cn - k0, g, m[t], l[t], acc[t]);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    const int q0 = (t0 + t) * 16;
    if (q0 >= nq) break;
    const float lt = wave16_sum(l[t]);
    if (q0 + r < nq) {
      if (4 * g < dh) {
        const float4 q4 = *reinterpret_cast<const float4*>(a.Qp + (int64_t)b * a.qb +
                                                           (int64_t)(q0 + r) * d + j * dh + 4 * g);
        const float inv = 1.f / lt;
        *reinterpret_cast<float4*>(a.Oout + ((int64_t)b * nq + q0 + r) * d + j * dh + 4 * g) =
            float4{q4.x + acc[t][0] * inv, q4.y + acc[t][1] * inv, q4.z + acc[t][2] * inv,
                   q4.w + acc[t][3] * inv};
      }
      if (g == 0) a.LSE[((int64_t)b * a.h + j) * nq + q0 + r] = m[t] + log2f(lt);
    }
  }
}

template <int QT>
__global__ __launch_bounds__(1024) void k_attnc_bwd_q(const AttnCoreArgs a) {
  extern __shared__ __attribute__((aligned(16))) __bf16 lds[];
  const int tid = threadIdx.x, lane = tid & 63, j = tid >> 6, r = lane & 15, g = lane >> 4;
  const int nthr = blockDim.x, d = a.d, dh = a.dh, nq = a.nq, nk = a.nk;
  int b, x;
  wg_coords(a.gx, b, x);
  if (b >= a.B) return;
  __bf16* Krow = lds;
  __bf16* Vrow = Krow + row_elems(CH, d);
  __bf16* Ktr = Vrow + row_elems(CH, d);
  lds_clear(lds, 2 * row_elems(CH, d) + tr_elems(CH, d), tid, nthr);
  int len = nk;
  if (a.lengths != nullptr) len = a.lengths[b] < nk ? a.lengths[b] : nk;
  const int t0 = x * QT;
  bf16x4 qb4[QT], dob[QT];
  float nlse[QT], nds[QT], delta[QT];
  float4 do4[QT];
  f32x4 acc[QT];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    const int q0 = (t0 + t) * 16;
    const int qi = q0 + r < nq ? q0 + r : nq - 1;
    const float* qrow = a.Qp + (int64_t)b * a.qb + (int64_t)qi * d + j * dh;
    const int64_t orow = ((int64_t)b * nq + qi) * d + j * dh;
    qb4[t] = row4(qrow, 4 * g, dh);
    dob[t] = row4(a.dO + orow, 4 * g, dh);
    nlse[t] = -a.LSE[((int64_t)b * a.h + j) * nq + qi];
    const int fo = 4 * g < dh ? 4 * g : 0;
    do4[t] = *reinterpret_cast<const float4*>(a.dO + orow + fo);
    const float4 o4 = *reinterpret_cast<const float4*>(a.O + orow + fo);
    const float4 q4 = *reinterpret_cast<const float4*>(qrow + fo);
    float dl = do4[t].x * (o4.x - q4.x) + do4[t].y * (o4.y - q4.y) + do4[t].z * (o4.z - q4.z) +
               do4[t].w * (o4.w - q4.w);
    if (4 * g >= dh) dl = 0.f;
    delta[t] = wave16_sum(dl);                         // rowdot(dO_j, A V_j) of query r
    nds[t] = -delta[t] * a.scale;
    acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const float* Kb = a.Kp + (int64_t)b * nk * d;
  const float* Vb = a.Vp + (int64_t)b * nk * d;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  for (int c0 = 0; c0 < len; c0 += CH) {
    __syncthreads();
    stage<CH, true, true>(Kb, nk, c0, d, Krow, Ktr, tid, nthr);
    stage<CH, true, false>(Vb, nk, c0, d, Vrow, nullptr, tid, nthr);
    __syncthreads();
    const int cn = len - c0 < CH ? len - c0 : CH;
    for (int k0 = 0; k0 < cn; k0 += 32) {
      bf16x4 ka[2], vr[2], kt[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int ro = (k0 + 16 * u + r) * (d + 4) + j * dh + 4 * g;
        ka[u] = *reinterpret_cast<const bf16x4*>(Krow + ro);
        vr[u] = *reinterpret_cast<const bf16x4*>(Vrow + ro);
        kt[u] = *reinterpret_cast<const bf16x4*>(Ktr + (j * dh + r) * (CH + 4) + k0 + 16 * u + 4 * g);
      }
      const bool full = k0 + 32 <= cn;
#pragma unroll
      for (int t = 0; t < QT; ++t) {
        if ((t0 + t) * 16 >= nq) break;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const f32x4 s = mfma16(ka[u], qb4[t], z4);    // [key 4 g + e][query r]
          const f32x4 dp = mfma16(vr[u], dob[t], z4);   // dP^T = V dO^T
          f32x4 ds;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float pe = __builtin_amdgcn_exp2f(fmaf(s[e], a.c, nlse[t]));
            if (!full && k0 + 16 * u + 4 * g + e >= cn) pe = 0.f;
            ds[e] = pe * fmaf(dp[e], a.scale, nds[t]);
          }
          acc[t] = mfma16(kt[u], pack4(ds), acc[t]);    // dQ^T += K^T dS^T
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    const int q0 = (t0 + t) * 16;
    if (q0 >= nq) break;
    if (q0 + r < nq) {
      if (4 * g < dh)
        *reinterpret_cast<float4*>(a.dQp + ((int64_t)b * nq + q0 + r) * d + j * dh + 4 * g) =
            float4{do4[t].x + acc[t][0], do4[t].y + acc[t][1], do4[t].z + acc[t][2],
                   do4[t].w + acc[t][3]};                                   // + the residual Q_
      if (g == 0) a.Delta[((int64_t)b * a.h + j) * nq + q0 + r] = delta[t];
    }
  }
}

template <int KT>
__global__ __launch_bounds__(1024) void k_attnc_bwd_kv(const AttnCoreArgs a) {
  extern __shared__ __attribute__((aligned(16))) __bf16 lds[];
  const int tid = threadIdx.x, lane = tid & 63, j = tid >> 6, r = lane & 15, g = lane >> 4;
  const int nthr = blockDim.x, d = a.d, dh = a.dh, nq = a.nq, nk = a.nk, h = a.h;
  int b, x;
  wg_coords(a.gx, b, x);
  if (b >= a.B) return;
  __bf16* Qrow = lds;
  __bf16* Drow = Qrow + row_elems(CHQ, d);
  __bf16* Qtr = Drow + row_elems(CHQ, d);
  __bf16* Dtr = Qtr + tr_elems(CHQ, d);
  const int nb16 = 2 * row_elems(CHQ, d) + 2 * tr_elems(CHQ, d);
  float* lseS = reinterpret_cast<float*>(lds + ((nb16 + 7) & ~7));   // [h][CHQ] each
  float* delS = lseS + h * CHQ;
  lds_clear(lds, nb16, tid, nthr);
  int len = nk;
  if (a.lengths != nullptr) len = a.lengths[b] < nk ? a.lengths[b] : nk;
  const int t0 = x * KT;                               // first key tile
  bf16x4 kb4[KT], vb4[KT];
  f32x4 dk[KT], dv[KT];
  const float* Kb = a.Kp + (int64_t)b * nk * d + j * dh;
  const float* Vb = a.Vp + (int64_t)b * nk * d + j * dh;
#pragma unroll
  for (int t = 0; t < KT; ++t) {
    const int k0 = (t0 + t) * 16;
    const int ki = k0 + r < nk ? k0 + r : nk - 1;
    kb4[t] = row4(Kb + (int64_t)ki * d, 4 * g, dh);    // B operand [k = feature][col = key r]
    vb4[t] = row4(Vb + (int64_t)ki * d, 4 * g, dh);
    dk[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    dv[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const float* Qb = a.Qp + (int64_t)b * a.qb;
  const float* dOb = a.dO + (int64_t)b * nq * d;
  const float* lseb = a.LSE + (int64_t)b * h * nq;
  const float* delb = a.Delta + (int64_t)b * h * nq;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  for (int c0 = 0; c0 < nq; c0 += CHQ) {
    __syncthreads();
    stage<CHQ, true, true>(Qb, nq, c0, d, Qrow, Qtr, tid, nthr);
    stage<CHQ, true, true>(dOb, nq, c0, d, Drow, Dtr, tid, nthr);
    for (int i = tid; i < h * CHQ; i += nthr) {        // queries past the end: P = exp2(-inf) = 0
      const int hh = i / CHQ, qq = i - hh * CHQ;
      const bool on = c0 + qq < nq;
      const int qc = on ? c0 + qq : nq - 1;
      const float lv = lseb[(int64_t)hh * nq + qc], dlv = delb[(int64_t)hh * nq + qc];
      lseS[i] = on ? -lv : -INFINITY;
      delS[i] = on ? -dlv * a.scale : 0.f;
    }
    __syncthreads();
#pragma unroll 2
    for (int u = 0; u < CHQ / 16; ++u) {
      if (c0 + 16 * u >= nq) break;
      const int ro = (16 * u + r) * (d + 4) + j * dh + 4 * g;
      const int to = (j * dh + r) * (CHQ + 4) + 16 * u + 4 * g;
      const bf16x4 qa = *reinterpret_cast<const bf16x4*>(Qrow + ro);    // A operand [query r][k = feature]
      const bf16x4 doa = *reinterpret_cast<const bf16x4*>(Drow + ro);
      const bf16x4 qt = *reinterpret_cast<const bf16x4*>(Qtr + to);     // Q^T, dO^T: [feature][k = query]
      const bf16x4 dot = *reinterpret_cast<const bf16x4*>(Dtr + to);
      const float4 nl = *reinterpret_cast<const float4*>(lseS + j * CHQ + 16 * u + 4 * g);
      const float4 nd = *reinterpret_cast<const float4*>(delS + j * CHQ + 16 * u + 4 * g);
      const float nl4[4] = {nl.x, nl.y, nl.z, nl.w}, nd4[4] = {nd.x, nd.y, nd.z, nd.w};
#pragma unroll
      for (int t = 0; t < KT; ++t) {
        const int k0 = (t0 + t) * 16;
        if (k0 >= nk) break;
        const f32x4 s = mfma16(qa, kb4[t], z4);        // [query 4 g + e][key r]
        const f32x4 dp = mfma16(doa, vb4[t], z4);      // dP = dO V^T
        const bool dead = k0 + r >= len;
        f32x4 p, ds;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          p[e] = __builtin_amdgcn_exp2f(fmaf(s[e], a.c, nl4[e]));
          if (k0 + 16 > len) p[e] = dead ? 0.f : p[e];
          ds[e] = p[e] * fmaf(dp[e], a.scale, nd4[e]);
        }
        dv[t] = mfma16(dot, pack4(p), dv[t]);          // dV^T += dO^T P     [feature 4 g + e][key r]
        dk[t] = mfma16(qt, pack4(ds), dk[t]);          // dK^T += Q^T dS
      }
    }
  }
#pragma unroll
  for (int t = 0; t < KT; ++t) {
    const int k0 = (t0 + t) * 16;
    if (k0 + r < nk && 4 * g < dh) {
      const int64_t o = ((int64_t)b * nk + k0 + r) * d + j * dh + 4 * g;
      *reinterpret_cast<float4*>(a.dKp + o) = float4{dk[t][0], dk[t][1], dk[t][2], dk[t][3]};
      *reinterpret_cast<float4*>(a.dVp + o) = float4{dv[t][0], dv[t][1], dv[t][2], dv[t][3]};
    }
  }
}

inline AttnCoreArgs core_args(const pca_mab_shape& s) {
  AttnCoreArgs a{};
  a.B = s.B; a.nq = s.nq; a.nk = s.nk; a.d = s.d; a.dh = s.d / s.h; a.h = s.h;
  a.qb = s.q_shared ? 0 : (int64_t)s.nq * s.d;
  a.scale = 1.0f / sqrtf((float)s.d);                 // modules.py:28: sqrt(dim_V)
  a.c = a.scale * 1.4426950408889634f;
  a.lengths = s.k_lengths;
  return a;
}

// tiles of the register side per wave: as many (4, 2, 1) as still leave two workgroups per CU
inline int tiles_per_wave(int B, int rows) {
  const int nt = (int)cdiv(rows, 16);
  for (int t = 4; t > 1; t >>= 1)
    if ((int64_t)B * cdiv(nt, t) >= 512) return t;
  return 1;
}
inline unsigned grid_of(AttnCoreArgs& a, int rows, int t) {
  a.gx = (int)cdiv(cdiv(rows, 16), t);
  return (unsigned)(8 * a.gx * cdiv(a.B, 8));
}

#define PCA_ATTNC_LAUNCH(KERN, T, GRID, LDSB)                                                     \
  switch (T) {                                                                                    \
    case 4: hipLaunchKernelGGL((KERN<4>), dim3(GRID), dim3(64 * s.h), LDSB, st, a); break;        \
    case 2: hipLaunchKernelGGL((KERN<2>), dim3(GRID), dim3(64 * s.h), LDSB, st, a); break;        \
    default: hipLaunchKernelGGL((KERN<1>), dim3(GRID), dim3(64 * s.h), LDSB, st, a); break;       \
  }

}  // namespace

// head dims whose 16-byte row pieces the K = 16 MFMA serves, d small enough for the staged chunks
// (55 KB of LDS at d = 64); the exact fp32 mode keeps its chain (the parity path materialises A like
// the reference)
bool attn_core_ok(const pca_mab_shape& s) {
  const int dh = s.d / s.h;
  return s.mode != PCA_MODE_F32 && !s.ln && dh <= 16 && dh % 4 == 0 && s.d % 4 == 0 && s.d <= 64 &&
         s.h <= 16;
}

int attn_core_fwd(const pca_mab_shape& s, const float* Qp, const float* Kp, const float* Vp, float* O,
                  float* LSE, hipStream_t st) {
  AttnCoreArgs a = core_args(s);
  a.Qp = Qp; a.Kp = Kp; a.Vp = Vp; a.Oout = O; a.LSE = LSE;
  const int t = tiles_per_wave(s.B, s.nq);
  const unsigned grid = grid_of(a, s.nq, t);
  const size_t ldsb = (size_t)(row_elems(CH, s.d) + tr_elems(CH, s.d)) * sizeof(__bf16);
  PCA_ATTNC_LAUNCH(k_attnc_fwd, t, grid, ldsb)
  return check_launch("k_attnc_fwd");
}

// Delta: [B][h][nq] floats of scratch
int attn_core_bwd(const pca_mab_shape& s, const float* Qp, const float* Kp, const float* Vp,
                  const float* O, const float* LSE, const float* dO, float* dQp, float* dKp, float* dVp,
                  float* Delta, hipStream_t st) {
  AttnCoreArgs a = core_args(s);
  a.Qp = Qp; a.Kp = Kp; a.Vp = Vp; a.O = O; a.LSE = const_cast<float*>(LSE); a.dO = dO;
  a.dQp = dQp; a.dKp = dKp; a.dVp = dVp; a.Delta = Delta;
  {
    const int t = tiles_per_wave(s.B, s.nq);
    const unsigned grid = grid_of(a, s.nq, t);
    const size_t ldsb = (size_t)(2 * row_elems(CH, s.d) + tr_elems(CH, s.d)) * sizeof(__bf16);
    PCA_ATTNC_LAUNCH(k_attnc_bwd_q, t, grid, ldsb)
    PCA_TRY(check_launch("k_attnc_bwd_q"));
  }
  const int t = tiles_per_wave(s.B, s.nk);
  const unsigned grid = grid_of(a, s.nk, t);
  const size_t nb16 = 2 * (size_t)row_elems(CHQ, s.d) + 2 * (size_t)tr_elems(CHQ, s.d);
  const size_t ldsb = ((nb16 + 7) & ~(size_t)7) * sizeof(__bf16) + 2 * (size_t)s.h * CHQ * sizeof(float);
  PCA_ATTNC_LAUNCH(k_attnc_bwd_kv, t, grid, ldsb)
  return check_launch("k_attnc_bwd_kv");
}

}  // namespace pca
