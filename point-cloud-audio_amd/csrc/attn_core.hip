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
//   k_attnc_fwd     per (set, head, 16 queries): S^T = K Q^T on the MFMA (16x16x16, the head dim zero-
//                   padded to 16), online softmax over the key tiles, O^T += V^T P^T; saves the
//                   log-sum-exp (log2 domain) instead of A
//   k_attnc_bwd_q   per (set, head, 16 queries): P^T recomputed from the LSE, dP^T = V dO^T,
//                   dS^T = P^T (dP^T - delta) / sqrt d, dQ^T += K^T dS^T ; delta = rowdot(dO, O - Q_)
//   k_attnc_bwd_kv  per (set, head, 16 keys): the same tiles in the other orientation (queries on the
//                   accumulator rows), dV^T += dO^T P, dK^T += Q^T dS - every element of dK / dV is
//                   written once, no atomics, no zero-fill
// Operands are rounded to bf16 on the way into the MFMA exactly as the chain's k_gemm_bf16 does; softmax
// statistics, accumulation and all tensors in memory stay fp32.  Queries shared by all sets (ISAB's I,
// PMA's S: modules.py:52,63) are read through a zero batch stride.  Keys at or beyond k_lengths[b] are
// masked (P = 0), as softmax_rows does for the chain.
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
  int B, nq, nk, d, dh;
  int64_t qb;                  // batch stride of Qp (0: shared)
  float scale, c;              // 1 / sqrt(d), scale * log2(e)
};

// 4 consecutive fp32 features (f0 .. f0 + 3) of one row as a bf16 MFMA operand; zeros beyond the head dim.
// The loads are UNCONDITIONAL (clamped address, value selected afterwards): under a divergent `if` every
// load gets a basic block and an s_waitcnt vmcnt(0) of its own, and these helpers are
// the kernels' whole memory traffic.
__device__ __forceinline__ bf16x4 row4(const float* row, int f0, int dh) {
  const bool on = f0 < dh;
  const float4 x = *reinterpret_cast<const float4*>(row + (on ? f0 : 0));
  bf16x4 v;
  v[0] = (__bf16)(on ? x.x : 0.f); v[1] = (__bf16)(on ? x.y : 0.f);
  v[2] = (__bf16)(on ? x.z : 0.f); v[3] = (__bf16)(on ? x.w : 0.f);
  return v;
}
// the transposed operand X^T[row = feature f][k = rows i0 .. i0 + 3 of X]: four strided scalars
__device__ __forceinline__ bf16x4 col4(const float* base, int64_t i0, int64_t imax, int64_t stride, int f,
                                       int dh) {
  const bool on = f < dh;
  const int fc = on ? f : 0;
  bf16x4 v;
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    const int64_t i = i0 + jj < imax ? i0 + jj : imax - 1;
    const float x = base[i * stride + fc];
    v[jj] = (__bf16)(on ? x : 0.f);
  }
  return v;
}

__global__ __launch_bounds__(256) void k_attnc_fwd(const AttnCoreArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
  const int b = blockIdx.z, j = blockIdx.y, q0 = (blockIdx.x * 4 + wave) * 16;
  if (q0 >= a.nq) return;
  const int d = a.d, dh = a.dh, nq = a.nq, nk = a.nk;
  int len = nk;
  if (a.lengths != nullptr) len = a.lengths[b] < nk ? a.lengths[b] : nk;
  const int qi = q0 + r < nq ? q0 + r : nq - 1;
  const float* qrow = a.Qp + (int64_t)b * a.qb + (int64_t)qi * d + j * dh;
  const bf16x4 qb4 = row4(qrow, 4 * g, dh);
  const float* Kb = a.Kp + (int64_t)b * nk * d + j * dh;
  const float* Vb = a.Vp + (int64_t)b * nk * d + j * dh;
  float m = -INFINITY, l = 0.f;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  // four key tiles per trip, the operands of all four requested before the first is used: one tile per
  // trip left every trip's load latency exposed (a wave of the few-queries blocks walks 65 tiles)
  for (int kb0 = 0; kb0 < len; kb0 += 64) {
    bf16x4 ka4[4], va4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k0 = kb0 + 16 * u;
      const int ki = k0 + r < nk ? k0 + r : nk - 1;
      ka4[u] = row4(Kb + (int64_t)ki * d, 4 * g, dh);
      va4[u] = col4(Vb, k0 + 4 * g, nk, d, r, dh);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k0 = kb0 + 16 * u;
      if (k0 >= len) break;
      const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
      f32x4 s = mfma16(ka4[u], qb4, z4);              // [key 4 g + e][query r]
      float mt = -INFINITY;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s[e] = k0 + 4 * g + e < len ? s[e] * a.c : -INFINITY;
        mt = __builtin_amdgcn_fmed3f(mt, s[e], INFINITY);
      }
      mt = wave16_max(mt);
      const float mn = __builtin_amdgcn_fmed3f(m, mt, INFINITY);                 // finite: the tile has at least one live key
      const float alpha = __builtin_amdgcn_exp2f(m - mn);
      float ls = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s[e] = __builtin_amdgcn_exp2f(s[e] - mn);
        ls += s[e];
      }
      l = l * alpha + wave16_sum(ls);
      m = mn;
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] *= alpha;    // the accumulator's column is this lane's query
      acc = mfma16(va4[u], pack4(s), acc);            // [feature 4 g + e][query r]
    }
  }
  if (q0 + r < nq) {
    if (4 * g < dh) {
      const float4 q4 = *reinterpret_cast<const float4*>(qrow + 4 * g);
      const float inv = 1.f / l;
      *reinterpret_cast<float4*>(a.Oout + ((int64_t)b * nq + q0 + r) * d + j * dh + 4 * g) =
          float4{q4.x + acc[0] * inv, q4.y + acc[1] * inv, q4.z + acc[2] * inv, q4.w + acc[3] * inv};
    }
    if (g == 0) a.LSE[((int64_t)b * gridDim.y + j) * nq + q0 + r] = m + log2f(l);
  }
}

__global__ __launch_bounds__(256) void k_attnc_bwd_q(const AttnCoreArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
  const int b = blockIdx.z, j = blockIdx.y, q0 = (blockIdx.x * 4 + wave) * 16;
  if (q0 >= a.nq) return;
  const int d = a.d, dh = a.dh, nq = a.nq, nk = a.nk;
  int len = nk;
  if (a.lengths != nullptr) len = a.lengths[b] < nk ? a.lengths[b] : nk;
  const int qi = q0 + r < nq ? q0 + r : nq - 1;
  const float* qrow = a.Qp + (int64_t)b * a.qb + (int64_t)qi * d + j * dh;
  const int64_t orow = ((int64_t)b * nq + qi) * d + j * dh;
  const bf16x4 qb4 = row4(qrow, 4 * g, dh);
  const bf16x4 dob = row4(a.dO + orow, 4 * g, dh);
  const float lse = a.LSE[((int64_t)b * gridDim.y + j) * nq + qi];
  float4 do4 = {0.f, 0.f, 0.f, 0.f};
  float delta = 0.f;
  if (4 * g < dh) {
    do4 = *reinterpret_cast<const float4*>(a.dO + orow + 4 * g);
    const float4 o4 = *reinterpret_cast<const float4*>(a.O + orow + 4 * g);
    const float4 q4 = *reinterpret_cast<const float4*>(qrow + 4 * g);
    delta = do4.x * (o4.x - q4.x) + do4.y * (o4.y - q4.y) + do4.z * (o4.z - q4.z) + do4.w * (o4.w - q4.w);
  }
  delta = wave16_sum(delta);                          // rowdot(dO_j, A V_j) of query r
  const float* Kb = a.Kp + (int64_t)b * nk * d + j * dh;
  const float* Vb = a.Vp + (int64_t)b * nk * d + j * dh;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int kb0 = 0; kb0 < len; kb0 += 64) {          // (four key tiles per trip: see the forward)
    bf16x4 ka4[4], vr4[4], kt4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k0 = kb0 + 16 * u;
      const int ki = k0 + r < nk ? k0 + r : nk - 1;
      ka4[u] = row4(Kb + (int64_t)ki * d, 4 * g, dh);
      vr4[u] = row4(Vb + (int64_t)ki * d, 4 * g, dh);
      kt4[u] = col4(Kb, k0 + 4 * g, nk, d, r, dh);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k0 = kb0 + 16 * u;
      if (k0 >= len) break;
      const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
      const f32x4 s = mfma16(ka4[u], qb4, z4);        // [key 4 g + e][query r]
      const f32x4 dp = mfma16(vr4[u], dob, z4);       // dP^T = V dO^T
      f32x4 ds;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float p = k0 + 4 * g + e < len ? __builtin_amdgcn_exp2f(s[e] * a.c - lse) : 0.f;
        ds[e] = p * (dp[e] - delta) * a.scale;
      }
      acc = mfma16(kt4[u], pack4(ds), acc);           // dQ^T += K^T dS^T
    }
  }
  if (q0 + r < nq) {
    if (4 * g < dh)
      *reinterpret_cast<float4*>(a.dQp + ((int64_t)b * nq + q0 + r) * d + j * dh + 4 * g) =
          float4{do4.x + acc[0], do4.y + acc[1], do4.z + acc[2], do4.w + acc[3]};   // + the residual Q_
    if (g == 0) a.Delta[((int64_t)b * gridDim.y + j) * nq + q0 + r] = delta;
  }
}

__global__ __launch_bounds__(256) void k_attnc_bwd_kv(const AttnCoreArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
  const int b = blockIdx.z, j = blockIdx.y, k0 = (blockIdx.x * 4 + wave) * 16;
  if (k0 >= a.nk) return;
  const int d = a.d, dh = a.dh, nq = a.nq, nk = a.nk;
  int len = nk;
  if (a.lengths != nullptr) len = a.lengths[b] < nk ? a.lengths[b] : nk;
  const int ki = k0 + r < nk ? k0 + r : nk - 1;
  const bool klive = k0 + r < len;
  const float* Kb = a.Kp + (int64_t)b * nk * d + j * dh;
  const float* Vb = a.Vp + (int64_t)b * nk * d + j * dh;
  const bf16x4 kb4 = row4(Kb + (int64_t)ki * d, 4 * g, dh);     // B operand [k = feature][col = key r]
  const bf16x4 vb4 = row4(Vb + (int64_t)ki * d, 4 * g, dh);
  const float* Qb = a.Qp + (int64_t)b * a.qb + j * dh;
  const float* dOb = a.dO + (int64_t)b * nq * d + j * dh;
  const float* lseb = a.LSE + ((int64_t)b * gridDim.y + j) * nq;
  const float* delb = a.Delta + ((int64_t)b * gridDim.y + j) * nq;
  f32x4 dk = {0.f, 0.f, 0.f, 0.f}, dv = {0.f, 0.f, 0.f, 0.f};
  for (int qb0 = 0; qb0 < nq; qb0 += 32) {           // (two query tiles per trip, operands first)
    bf16x4 qa2[2], doa2[2], qt2[2], dot2[2];
    float lse8[2][4], del8[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int q0 = qb0 + 16 * u;
      const int qi = q0 + r < nq ? q0 + r : nq - 1;
      qa2[u] = row4(Qb + (int64_t)qi * d, 4 * g, dh);     // A operand [row = query r][k = feature]
      doa2[u] = row4(dOb + (int64_t)qi * d, 4 * g, dh);
      qt2[u] = col4(Qb, q0 + 4 * g, nq, d, r, dh);        // Q^T, dO^T: [row = feature][k = query]
      dot2[u] = col4(dOb, q0 + 4 * g, nq, d, r, dh);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int qq = q0 + 4 * g + e < nq ? q0 + 4 * g + e : nq - 1;
        lse8[u][e] = lseb[qq];
        del8[u][e] = delb[qq];
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int q0 = qb0 + 16 * u;
      if (q0 >= nq) break;
      const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
      const f32x4 s = mfma16(qa2[u], kb4, z4);        // [query 4 g + e][key r]
      const f32x4 dp = mfma16(doa2[u], vb4, z4);      // dP = dO V^T
      f32x4 p, ds;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool on = klive && q0 + 4 * g + e < nq;
        p[e] = on ? __builtin_amdgcn_exp2f(s[e] * a.c - lse8[u][e]) : 0.f;
        ds[e] = p[e] * (dp[e] - del8[u][e]) * a.scale;
      }
      dv = mfma16(dot2[u], pack4(p), dv);             // dV^T += dO^T P     [feature 4 g + e][key r]
      dk = mfma16(qt2[u], pack4(ds), dk);             // dK^T += Q^T dS
    }
  }
  if (k0 + r < nk && 4 * g < dh) {
    const int64_t o = ((int64_t)b * nk + k0 + r) * d + j * dh + 4 * g;
    *reinterpret_cast<float4*>(a.dKp + o) = float4{dk[0], dk[1], dk[2], dk[3]};
    *reinterpret_cast<float4*>(a.dVp + o) = float4{dv[0], dv[1], dv[2], dv[3]};
  }
}

inline AttnCoreArgs core_args(const pca_mab_shape& s) {
  AttnCoreArgs a{};
  a.B = s.B; a.nq = s.nq; a.nk = s.nk; a.d = s.d; a.dh = s.d / s.h;
  a.qb = s.q_shared ? 0 : (int64_t)s.nq * s.d;
  a.scale = 1.0f / sqrtf((float)s.d);                 // modules.py:28: sqrt(dim_V)
  a.c = a.scale * 1.4426950408889634f;
  a.lengths = s.k_lengths;
  return a;
}

}  // namespace

// head dims the zero-padded K = 16 MFMA serves with 16-byte row pieces; the exact fp32 mode keeps its
// chain (the parity path materialises A like the reference)
bool attn_core_ok(const pca_mab_shape& s) {
  const int dh = s.d / s.h;
  return s.mode != PCA_MODE_F32 && !s.ln && dh <= 16 && dh % 4 == 0 && s.d % 4 == 0;
}

int attn_core_fwd(const pca_mab_shape& s, const float* Qp, const float* Kp, const float* Vp, float* O,
                  float* LSE, hipStream_t st) {
  AttnCoreArgs a = core_args(s);
  a.Qp = Qp; a.Kp = Kp; a.Vp = Vp; a.Oout = O; a.LSE = LSE;
  hipLaunchKernelGGL(k_attnc_fwd, dim3((unsigned)cdiv(s.nq, 64), s.h, s.B), dim3(256), 0, st, a);
  return check_launch("k_attnc_fwd");
}

// Delta: [B][h][nq] floats of scratch
int attn_core_bwd(const pca_mab_shape& s, const float* Qp, const float* Kp, const float* Vp,
                  const float* O, const float* LSE, const float* dO, float* dQp, float* dKp, float* dVp,
                  float* Delta, hipStream_t st) {
  AttnCoreArgs a = core_args(s);
  a.Qp = Qp; a.Kp = Kp; a.Vp = Vp; a.O = O; a.LSE = const_cast<float*>(LSE); a.dO = dO;
  a.dQp = dQp; a.dKp = dKp; a.dVp = dVp; a.Delta = Delta;
  hipLaunchKernelGGL(k_attnc_bwd_q, dim3((unsigned)cdiv(s.nq, 64), s.h, s.B), dim3(256), 0, st, a);
  PCA_TRY(check_launch("k_attnc_bwd_q"));
  hipLaunchKernelGGL(k_attnc_bwd_kv, dim3((unsigned)cdiv(s.nk, 64), s.h, s.B), dim3(256), 0, st, a);
  return check_launch("k_attnc_bwd_kv");
}

}  // namespace pca
