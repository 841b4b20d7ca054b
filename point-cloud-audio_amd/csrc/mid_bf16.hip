// Per-set "mid" kernels of an ISAB (set_transformer-master/modules.py:51-53) in the fused
// bf16 mode: everything that happens on the [m = 16, d = 128] inducing-point tensors between
// the two point-sized kernels, on the MFMA in the transposed-chain layout (mfma_common.hpp).
//
//   k_mid_fwd   merge the attention partials of mab0 -> T ; O = Qp + T_h Wv_h^T + bv ;
//               Z = O Wo^T + bo ; H = O + relu(Z)          (mab0 epilogue, modules.py:29-31)
//               Kp = H Wk^T + bk ; Vp = H Wv^T + bv        (mab1 projections, modules.py:21)
//               written in the four bf16 images k_mab1_fwd / k_mab1_bwd read.
//   k_mid_bwd   dH = dKp Wk + dVp Wv ; dZ = dH.[Z>0] ; dO = dH + dZ Wo ; dT_h = dO_h Wv_h ;
//               Delta = rowdot(dT, T)  -> the images k_mab0_bwd reads, plus dZ / dO / T_h
//               (fp32) for the weight-gradient reductions and sum_b dO for the shared query.
// One workgroup per set, wave w = head w.  Each wave needs every weight fragment exactly
// once, so fragments are read straight from L2 (no LDS staging); activations cross waves
// through small bf16 LDS images.
#include "mab1_bf16.hpp"

#include <math.h>

namespace pca {

namespace {

constexpr int D = 128, MQ = 16, ROWB = 256;

__device__ __forceinline__ bf16x8 gload8(const __bf16* p) {
  return *reinterpret_cast<const bf16x8*>(p);
}
// write one accumulator tile (rows = features 16t+4g+e, col = query r) into a [query][feature]
// bf16 image
__device__ __forceinline__ void put_tile(char* img, int t, int r, int g, f32x4 v) {
  *reinterpret_cast<bf16x4*>(img + swz(r, 2 * t + (g >> 1), ROWB) + 8 * (g & 1)) = pack4(v);
}

struct MidFwdArgs {
  // attention partials of mab0 (S > 0) or merged T (S == 0, layer 1)
  const float *Tp, *Mp, *Lp;
  int S;
  float *T, *LSE;             // [B][64][dk], [B][64]  (saved)
  const float* Qp;            // [16][128]
  const __bf16* Wv0;          // [128][dk] natural (dk == 128) ...
  const float* Wv0f;          // ... or fp32 for dk <= 4
  const float *bv0, *bo0;
  const __bf16* Wo0;          // [128][128] natural
  const __bf16 *Wk1, *Wv1;    // [128][128] natural
  const float *bk1, *bv1;
  float *O, *Z, *H;           // [B][16][128] fp32
  __bf16 *KpP, *VpP, *Kt, *Vt;
  int dk;
};

template <bool SMALL>
__global__ __launch_bounds__(256) void k_mid_fwd(const MidFwdArgs a) {
  __shared__ __attribute__((aligned(16))) char sT[64 * ROWB];   // T  bf16 [r][c]
  __shared__ __attribute__((aligned(16))) char sA[16 * ROWB];   // O, then H  bf16 [q][f]
  __shared__ float sTf[SMALL ? 64 * 4 : 1];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int dk = a.dk;

  // every weight fragment / bias of phases A-C is fetched now (latency-bound kernel: one
  // workgroup per set; the round trips overlap the partial merge instead of following it)
  bf16x8 wv0_pre[2][SMALL ? 1 : 4], wo0_pre[2][4], wk1_pre[2][4], wv1_pre[2][4];
  float4 q_pre[2], bv0_pre[2], bo0_pre[2], bk1_pre[2], bv1_pre[2];
  float bk1c_pre[2], bv1c_pre[2];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    const int t = 2 * w + tt;
    q_pre[tt] = *reinterpret_cast<const float4*>(a.Qp + r * D + 16 * t + 4 * g);
    bv0_pre[tt] = *reinterpret_cast<const float4*>(a.bv0 + 16 * t + 4 * g);
    bo0_pre[tt] = *reinterpret_cast<const float4*>(a.bo0 + 16 * t + 4 * g);
    bk1_pre[tt] = *reinterpret_cast<const float4*>(a.bk1 + 16 * t + 4 * g);
    bv1_pre[tt] = *reinterpret_cast<const float4*>(a.bv1 + 16 * t + 4 * g);
    bk1c_pre[tt] = a.bk1[16 * t + r];
    bv1c_pre[tt] = a.bv1[16 * t + r];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (!SMALL) wv0_pre[tt][ks] = gload8(a.Wv0 + (int64_t)(16 * t + r) * D + 32 * ks + 8 * g);
      wo0_pre[tt][ks] = gload8(a.Wo0 + (int64_t)(16 * t + r) * D + 32 * ks + 8 * g);
      wk1_pre[tt][ks] = gload8(a.Wk1 + (int64_t)(16 * t + r) * D + 32 * ks + 8 * g);
      wv1_pre[tt][ks] = gload8(a.Wv1 + (int64_t)(16 * t + r) * D + 32 * ks + 8 * g);
    }
  }
  // ---- merge partials -> T (global fp32, saved) and its bf16 image ----
  if (SMALL) {
    for (int i = tid; i < 64 * dk; i += 256) sTf[(i / dk) * 4 + (i % dk)] = a.T[(int64_t)b * 64 * dk + i];
  } else {
#pragma unroll
    for (int it = 0; it < 4; ++it) {                   // (row, 8-element chunk)
      const int i = tid + 256 * it;
      const int row = i >> 4, ch = i & 15;
      // the partials of up to four point ranges at a time, all loads issued together (clamped
      // slot, zero weight): one round trip per group instead of three dependent ones per range
      float M = -INFINITY;
      float L = 0.f, t[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      float msv[8];
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        msv[s] = a.Mp[((int64_t)b * a.S + (s < a.S ? s : 0)) * 64 + row];
        if (s < a.S) M = fmaxf(M, msv[s]);
      }
#pragma unroll
      for (int s0 = 0; s0 < 8; s0 += 4) {
        if (s0 >= a.S) break;
        float lpv[4];
        float4 lo[4], hi[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int64_t o = ((int64_t)b * a.S + (s0 + u < a.S ? s0 + u : 0)) * 64 + row;
          lpv[u] = a.Lp[o];
          const float4* tp = reinterpret_cast<const float4*>(a.Tp + o * D + ch * 8);
          lo[u] = tp[0];
          hi[u] = tp[1];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float ms = msv[s0 + u];
          if (s0 + u >= a.S || ms == -INFINITY) continue;
          const float fs = exp2f(ms - M);
          L += fs * lpv[u];
          t[0] += fs * lo[u].x; t[1] += fs * lo[u].y; t[2] += fs * lo[u].z; t[3] += fs * lo[u].w;
          t[4] += fs * hi[u].x; t[5] += fs * hi[u].y; t[6] += fs * hi[u].z; t[7] += fs * hi[u].w;
        }
      }
      const float inv = 1.f / L;
      bf16x8 v;
      float4 o0, o1;
#pragma unroll
      for (int k = 0; k < 8; ++k) { t[k] *= inv; v[k] = (__bf16)t[k]; }
      o0 = float4{t[0], t[1], t[2], t[3]};
      o1 = float4{t[4], t[5], t[6], t[7]};
      float4* tg = reinterpret_cast<float4*>(a.T + ((int64_t)b * 64 + row) * D + ch * 8);
      tg[0] = o0; tg[1] = o1;
      *reinterpret_cast<bf16x8*>(sT + swz(row, ch, ROWB)) = v;
      if (ch == 0) a.LSE[(int64_t)b * 64 + row] = M + log2f(L);
    }
  }
  __syncthreads();

  // ---- phase A: O^T tiles of head w (features 32w .. 32w+31) ----
  f32x4 o[2];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    const int t = 2 * w + tt;
    const float4 q4 = q_pre[tt];
    const float4 b4 = bv0_pre[tt];
    o[tt] = f32x4{q4.x + b4.x, q4.y + b4.y, q4.z + b4.z, q4.w + b4.w};
    if (SMALL) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int f = 16 * t + 4 * g + e;
        for (int c = 0; c < dk; ++c) o[tt][e] += sTf[(16 * w + r) * 4 + c] * a.Wv0f[f * dk + c];
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        o[tt] = mfma32(wv0_pre[tt][ks],
                       *reinterpret_cast<const bf16x8*>(sT + swz(16 * w + r, 4 * ks + g, ROWB)),
                       o[tt]);
    }
    put_tile(sA, t, r, g, o[tt]);
    *reinterpret_cast<float4*>(a.O + ((int64_t)b * MQ + r) * D + 16 * t + 4 * g) =
        float4{o[tt][0], o[tt][1], o[tt][2], o[tt][3]};
  }
  __syncthreads();

  // ---- phase B: Z^T = Wo O^T + bo ; H = O + relu(Z) ----
  f32x4 hq[2];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    const int t = 2 * w + tt;
    const float4 b4 = bo0_pre[tt];
    f32x4 z = f32x4{b4.x, b4.y, b4.z, b4.w};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      z = mfma32(wo0_pre[tt][ks],
                 *reinterpret_cast<const bf16x8*>(sA + swz(r, 4 * ks + g, ROWB)), z);
#pragma unroll
    for (int e = 0; e < 4; ++e) hq[tt][e] = o[tt][e] + fmaxf(z[e], 0.f);
    const int64_t off = ((int64_t)b * MQ + r) * D + 16 * t + 4 * g;
    *reinterpret_cast<float4*>(a.Z + off) = float4{z[0], z[1], z[2], z[3]};
    *reinterpret_cast<float4*>(a.H + off) = float4{hq[tt][0], hq[tt][1], hq[tt][2], hq[tt][3]};
  }
  __syncthreads();                     // every wave has read the O image
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) put_tile(sA, 2 * w + tt, r, g, hq[tt]);
  __syncthreads();

  // ---- phase C: Kp, Vp of mab1 in both orientations ----
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    __bf16* PP = which ? a.VpP : a.KpP;
    __bf16* TT = which ? a.Vt : a.Kt;
    f32x4 fr[2], kr[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      const int t = 2 * w + tt;
      const float4 b4 = which ? bv1_pre[tt] : bk1_pre[tt];
      const float bc = which ? bv1c_pre[tt] : bk1c_pre[tt];
      fr[tt] = f32x4{b4.x, b4.y, b4.z, b4.w};      // rows = features, col = key
      kr[tt] = f32x4{bc, bc, bc, bc};              // rows = keys,     col = feature
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 wf = which ? wv1_pre[tt][ks] : wk1_pre[tt][ks];
        const bf16x8 hf = *reinterpret_cast<const bf16x8*>(sA + swz(r, 4 * ks + g, ROWB));
        fr[tt] = mfma32(wf, hf, fr[tt]);
        kr[tt] = mfma32(hf, wf, kr[tt]);
      }
      // keys 4g..4g+3 of feature 16t + r : 8 contiguous bytes of the [feature][key] image
      *reinterpret_cast<bf16x4*>(TT + ((int64_t)b * D + 16 * t + r) * MQ + 4 * g) = pack4(kr[tt]);
    }
    // features 32w + perm32(8g + .) of key r : 16 contiguous bytes of the permuted image
    *reinterpret_cast<bf16x8*>(PP + ((int64_t)b * MQ + r) * D + 32 * w + 8 * g) =
        pack8(fr[0], fr[1]);
  }
}

struct MidBwdArgs {
  const float *dKpPart, *dVpPart;   // [B][nparts][16][128] fp32
  int nparts;
  float *dKp, *dVp;             // [B][16][128] fp32 sums (out)
  float* zero_ptr;
  int zero_n;
  const float *Z, *T, *LSE;     // saved by the forward
  const __bf16 *Wk1T, *Wv1T;    // [128][128] transposed natural:  W^T[c][f]
  const __bf16* Wo0TP;          // [128][128] transposed, K-permuted
  const __bf16* Wv0TP;          // [dk][128]  transposed, K-permuted (dk == 128)
  const __bf16* Wv0T;           // [dk][128]  transposed natural
  const float* Wv0f;            // fp32 [128][dk] for dk <= 4
  float *dZ, *dO;               // [B][16][128] fp32
  float* Th;                    // [4][B*16][dk] head-major copy of T
  float* dQs;                   // [16][128], atomically accumulated sum over sets of dO
  float* dTf;                   // [B][64][dk] fp32 (dk <= 4)
  __bf16 *dTb, *dTt;            // [B][64][128], [B][128][64]
  float *Delta, *LSEp;          // [B][64]
  int dk, B;
};

template <bool SMALL>
__global__ __launch_bounds__(256) void k_mid_bwd(const MidBwdArgs a) {
  __shared__ __attribute__((aligned(16))) char sK[16 * ROWB];    // dKp bf16 [q][f]
  __shared__ __attribute__((aligned(16))) char sV[16 * ROWB];    // dVp
  __shared__ __attribute__((aligned(16))) char sO[4][16 * 64];   // per wave: dO_j bf16 [q][32]
  // every wave needs ALL of Wk1^T and Wv1^T: stage them once, cooperatively (coalesced,
  // 16 loads of 16 B in flight per thread) instead of 4 waves chasing fragments through L2
  __shared__ __attribute__((aligned(16))) char sWk[128 * ROWB];
  __shared__ __attribute__((aligned(16))) char sWv[128 * ROWB];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int dk = a.dk;

#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = tid + 256 * e;                 // 2048 chunks of 16 B per matrix
    const int row = c >> 4, ch = c & 15;
    const uint4 kv = *reinterpret_cast<const uint4*>(a.Wk1T + (int64_t)row * D + ch * 8);
    const uint4 vv = *reinterpret_cast<const uint4*>(a.Wv1T + (int64_t)row * D + ch * 8);
    *reinterpret_cast<uint4*>(sWk + swz(row, ch, ROWB)) = kv;
    *reinterpret_cast<uint4*>(sWv + swz(row, ch, ROWB)) = vv;
  }
  // Every global operand of the chain below is fetched NOW, before the first barrier: the
  // kernel is one workgroup per set on half the CUs and purely latency-bound, so the round
  // trips must overlap instead of queueing behind each other.
  const int64_t trow = (int64_t)b * 64 + 16 * w + r;           // this lane's query row of T
  float4 zpre[8];
#pragma unroll
  for (int t = 0; t < 8; ++t)
    zpre[t] = *reinterpret_cast<const float4*>(a.Z + ((int64_t)b * MQ + r) * D + 16 * t + 4 * g);
  bf16x8 wo_pre[2][4];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
      wo_pre[tt][s4] = gload8(a.Wo0TP + (int64_t)(16 * (2 * w + tt) + r) * D + 32 * s4 + 8 * g);
  bf16x8 wvp_pre[SMALL ? 1 : 8], wvt_pre[SMALL ? 1 : 8];
  float4 t_pre[SMALL ? 1 : 8];
  float wvf_pre[SMALL ? 4 : 1][8], ts_pre[SMALL ? 4 : 1];
  if (SMALL) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      ts_pre[c] = c < dk ? a.T[trow * dk + c] : 0.f;
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          wvf_pre[c][4 * tt + e] =
              c < dk ? a.Wv0f[(32 * w + 16 * tt + 4 * g + e) * dk + c] : 0.f;
    }
  } else {
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) {
      wvp_pre[ct] = gload8(a.Wv0TP + (int64_t)(16 * ct + r) * D + 32 * w + 8 * g);
      wvt_pre[ct] = gload8(a.Wv0T + (int64_t)(16 * ct + r) * D + 32 * w + 8 * g);
      t_pre[ct] = *reinterpret_cast<const float4*>(a.T + trow * D + 16 * ct + 4 * g);
    }
  }
  const float lse_pre = a.LSE[(int64_t)b * 64 + 16 * w + r];
  if (a.zero_ptr != nullptr)
    for (int i = blockIdx.x * 256 + tid; i < a.zero_n; i += gridDim.x * 256) a.zero_ptr[i] = 0.f;
  for (int i = tid; i < 16 * 16; i += 256) {
    const int row = i >> 4, ch = i & 15;
    float k[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // four partials at a time, their loads issued together (one round trip per group, not per
    // partial); same summation order
    for (int p0 = 0; p0 < a.nparts; p0 += 4) {
      float4 k0[4], k1[4], v0[4], v1[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int pp = p0 + u < a.nparts ? p0 + u : p0;
        const int64_t off = (((int64_t)b * a.nparts + pp) * MQ + row) * D + ch * 8;
        const float4* pk = reinterpret_cast<const float4*>(a.dKpPart + off);
        const float4* pv = reinterpret_cast<const float4*>(a.dVpPart + off);
        k0[u] = pk[0]; k1[u] = pk[1]; v0[u] = pv[0]; v1[u] = pv[1];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (p0 + u >= a.nparts) continue;
        k[0] += k0[u].x; k[1] += k0[u].y; k[2] += k0[u].z; k[3] += k0[u].w;
        k[4] += k1[u].x; k[5] += k1[u].y; k[6] += k1[u].z; k[7] += k1[u].w;
        v[0] += v0[u].x; v[1] += v0[u].y; v[2] += v0[u].z; v[3] += v0[u].w;
        v[4] += v1[u].x; v[5] += v1[u].y; v[6] += v1[u].z; v[7] += v1[u].w;
      }
    }
    const int64_t so = ((int64_t)b * MQ + row) * D + ch * 8;
    reinterpret_cast<float4*>(a.dKp + so)[0] = float4{k[0], k[1], k[2], k[3]};
    reinterpret_cast<float4*>(a.dKp + so)[1] = float4{k[4], k[5], k[6], k[7]};
    reinterpret_cast<float4*>(a.dVp + so)[0] = float4{v[0], v[1], v[2], v[3]};
    reinterpret_cast<float4*>(a.dVp + so)[1] = float4{v[4], v[5], v[6], v[7]};
    bf16x8 kb, vb;
#pragma unroll
    for (int e = 0; e < 8; ++e) { kb[e] = (__bf16)k[e]; vb[e] = (__bf16)v[e]; }
    *reinterpret_cast<bf16x8*>(sK + swz(row, ch, ROWB)) = kb;
    *reinterpret_cast<bf16x8*>(sV + swz(row, ch, ROWB)) = vb;
  }
  __syncthreads();

  // ---- dH^T (all 8 feature tiles, every wave: avoids a cross-wave exchange) ----
  bf16x8 kb[4], vb[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    kb[ks] = *reinterpret_cast<const bf16x8*>(sK + swz(r, 4 * ks + g, ROWB));
    vb[ks] = *reinterpret_cast<const bf16x8*>(sV + swz(r, 4 * ks + g, ROWB));
  }
  f32x4 dh[8], dz[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    dh[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      dh[t] = mfma32(*reinterpret_cast<const bf16x8*>(sWk + swz(16 * t + r, 4 * ks + g, ROWB)),
                     kb[ks], dh[t]);
      dh[t] = mfma32(*reinterpret_cast<const bf16x8*>(sWv + swz(16 * t + r, 4 * ks + g, ROWB)),
                     vb[ks], dh[t]);
    }
    const int64_t off = ((int64_t)b * MQ + r) * D + 16 * t + 4 * g;
    const float4 z4 = zpre[t];
    dz[t][0] = z4.x > 0.f ? dh[t][0] : 0.f;
    dz[t][1] = z4.y > 0.f ? dh[t][1] : 0.f;
    dz[t][2] = z4.z > 0.f ? dh[t][2] : 0.f;
    dz[t][3] = z4.w > 0.f ? dh[t][3] : 0.f;
    if ((t >> 1) == w)
      *reinterpret_cast<float4*>(a.dZ + off) = float4{dz[t][0], dz[t][1], dz[t][2], dz[t][3]};
  }

  // ---- dO^T tiles of head w ----
  f32x4 dO2[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  // select this wave's two tiles without dynamic register indexing
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    if (t == 2 * w) dO2[0] = dh[t];
    if (t == 2 * w + 1) dO2[1] = dh[t];
  }
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    const int t = 2 * w + tt;
#pragma unroll
    for (int s = 0; s < 4; ++s)
      dO2[tt] = mfma32(wo_pre[tt][s], pack8(dz[2 * s], dz[2 * s + 1]), dO2[tt]);
    const int64_t off = ((int64_t)b * MQ + r) * D + 16 * t + 4 * g;
    *reinterpret_cast<float4*>(a.dO + off) =
        float4{dO2[tt][0], dO2[tt][1], dO2[tt][2], dO2[tt][3]};
    // (the sum of dO over the sets is taken by k_mab0_post1: B workgroups adding atomically
    //  into the same 2048 addresses serialised for microseconds)
    // wave-private [q][32] image of dO_w (64-byte rows): feature 16tt+4g.. of query r
    *reinterpret_cast<bf16x4*>(sO[w] + r * 64 + (16 * tt + 4 * g) * 2) = pack4(dO2[tt]);
  }

  // ---- dT of head w (rows 16w .. 16w+15 of the [64][dk] tensor) and Delta ----
  float dl = 0.f;
  if (SMALL) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (c >= dk) break;
      float part = 0.f;
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int e = 0; e < 4; ++e) part += dO2[tt][e] * wvf_pre[c][4 * tt + e];
      part = wave16_sum(part);
      const float tv = ts_pre[c];
      if (g == 0) {
        a.dTf[trow * dk + c] = part;
        a.Th[((int64_t)w * a.B * MQ + (int64_t)b * MQ + r) * dk + c] = tv;
      }
      dl += part * tv;
    }
  } else {
    const bf16x8 dob = pack8(dO2[0], dO2[1]);
    // query-row operand for the second orientation: dO_w[q][f], f natural, from the image
    const bf16x8 doa = *reinterpret_cast<const bf16x8*>(sO[w] + r * 64 + 16 * g);
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) {
      // (1) rows = columns c of dT, col = query: natural-row image + Delta
      f32x4 t1 = {0.f, 0.f, 0.f, 0.f};
      t1 = mfma32(wvp_pre[ct], dob, t1);
      const int64_t toff = trow * D + 16 * ct + 4 * g;
      *reinterpret_cast<bf16x4*>(a.dTb + toff) = pack4(t1);
      const float4 tv = t_pre[ct];
      dl += t1[0] * tv.x + t1[1] * tv.y + t1[2] * tv.z + t1[3] * tv.w;
      *reinterpret_cast<float4*>(a.Th + ((int64_t)w * a.B * MQ + (int64_t)b * MQ + r) * D +
                                 16 * ct + 4 * g) = tv;
      // (2) rows = queries 4g+e, col = column c = 16ct + r: the r-permuted transposed image
      f32x4 t2 = {0.f, 0.f, 0.f, 0.f};
      t2 = mfma32(doa, wvt_pre[ct], t2);
      *reinterpret_cast<bf16x4*>(a.dTt + ((int64_t)b * D + 16 * ct + r) * 64 + 32 * (w >> 1) +
                                 8 * g + 4 * (w & 1)) = pack4(t2);
    }
    dl = wave16_sum(dl);
  }
  if (SMALL) {
    // every g-lane holds the full dT (after the reduction) times T: dl is already complete
  }
  if (g == 0) {
    a.Delta[(int64_t)b * 64 + 16 * w + r] = dl;
    a.LSEp[(int64_t)b * 64 + 16 * w + r] = lse_pre;
  }
}

}  // namespace

int mid_fwd_launch(const MidFwdLaunch& L, hipStream_t st) {
  MidFwdArgs a{};
  a.Tp = L.Tp; a.Mp = L.Mp; a.Lp = L.Lp; a.S = L.S; a.T = L.T; a.LSE = L.LSE; a.Qp = L.Qp;
  a.Wv0 = L.Wv0; a.Wv0f = L.Wv0f; a.bv0 = L.bv0; a.bo0 = L.bo0; a.Wo0 = L.Wo0; a.Wk1 = L.Wk1;
  a.Wv1 = L.Wv1; a.bk1 = L.bk1; a.bv1 = L.bv1; a.O = L.O; a.Z = L.Z; a.H = L.H; a.KpP = L.KpP;
  a.VpP = L.VpP; a.Kt = L.Kt; a.Vt = L.Vt; a.dk = L.dk;
  if (L.dk <= 4) hipLaunchKernelGGL((k_mid_fwd<true>), dim3(L.B), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((k_mid_fwd<false>), dim3(L.B), dim3(256), 0, st, a);
  return check_launch("k_mid_fwd");
}

int mid_bwd_launch(const MidBwdLaunch& L, hipStream_t st) {
  MidBwdArgs a{};
  a.dKpPart = L.dKpPart; a.dVpPart = L.dVpPart; a.nparts = L.nparts; a.dKp = L.dKp;
  a.dVp = L.dVp; a.zero_ptr = L.zero_ptr; a.zero_n = L.zero_n; a.Z = L.Z; a.T = L.T; a.LSE = L.LSE; a.Wk1T = L.Wk1T;
  a.Wv1T = L.Wv1T; a.Wo0TP = L.Wo0TP; a.Wv0TP = L.Wv0TP; a.Wv0T = L.Wv0T; a.Wv0f = L.Wv0f;
  a.dZ = L.dZ; a.dO = L.dO; a.Th = L.Th; a.dQs = L.dQs; a.dTf = L.dTf; a.dTb = L.dTb;
  a.dTt = L.dTt; a.Delta = L.Delta; a.LSEp = L.LSEp; a.dk = L.dk; a.B = L.B;
  if (L.dk <= 4) hipLaunchKernelGGL((k_mid_bwd<true>), dim3(L.B), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((k_mid_bwd<false>), dim3(L.B), dim3(256), 0, st, a);
  return check_launch("k_mid_bwd");
}

}  // namespace pca
