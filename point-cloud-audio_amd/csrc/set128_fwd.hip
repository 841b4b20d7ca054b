// Set-resident forward of the d = 128 / 4 heads / m = 16 inducing points / k = 1 Set Transformer
// (Code/models.py:34-44: ISAB, ISAB, PMA; set_transformer-master/modules.py:19-33, 51-53, 62-63),
// BASELINE configs[1] (N = 512, din = 2, B = 128).
//
// ONE 1024-thread workgroup carries ONE set through
//     mab0(I1, X) -> mid -> mab1(X, H1) -> mab0(I2, Y1) -> mid -> mab1(Y1, H2) -> PMA attention partials
// with the set's [N, 128] bf16 activation tensor resident in LDS (128 KiB at N = 512) from the moment
// layer 1 produces it until the PMA has read it.  The per-block kernels this replaces (k_mab0_attn_small,
// k_mid_fwd, k_mab1_fwd, k_mab0_attn_h4, k_mid_fwd, k_isab1_fwd_t, k_mab0_attn) each ran ONE 32-point
// unit per wave with all 2048 waves of the chip in the same phase at the same time: every launch paid
// its weight images, the input burst, the dependent MFMA chain and the store drain in lock-step, plus
// ~5 us of launch and drain - seven times per forward.  Here a phase boundary is a workgroup barrier,
// the activations between blocks never leave the CU (they go to HBM only as the tensors the backward
// reads), and the attention partials of a set are merged by the workgroup that produced them.
//
// Wave w = (quad q = w >> 2, head j = w & 3): quad q owns the points [q N/4, (q+1) N/4) of the set,
// wave j of a quad the head j (features 32 j .. 32 j + 31) - the WAVE = HEAD, WEIGHTS IN REGISTERS form of
// k_isab1_fwd_t (d128_fused.hip) and k_mab0_attn_h4 (mab0_bf16.hip), whose arithmetic and saved layouts
// this kernel keeps bit for bit in form (same MFMA operand orders, same K-permuted images), so that the
// existing backward kernels read what it saves.
//
// LDS: sY [N][256 B] (rows XOR-swizzled as tr_off, so that both the row-major B-operand reads and the
// transposing ds_read_tr16_b64 reads are conflict-free) + 32 KiB of phase scratch.  The cross-head
// exchange of the many-queries blocks (fc_o contracts over all heads' features) happens IN PLACE in sY:
// O slices overwrite the unit's input rows, Y slices overwrite the O rows, three barriers per block.
//
// Roofline unit: MACs_fwd / set = N (3 din d + 7 d^2 + 8 m d + 2 k d) + 6 m d^2 (the
// PMA epilogue and the classifier run in k_pma_head1); algorithmic bytes 4 N din + 4 (2 N d) per set.
#include "set128.hpp"

#include <math.h>

#include <mutex>

namespace pca {

namespace {

constexpr int D = 128, MQ = 16, ROWB = 256, NT = 1024;

// byte offset of 16-byte chunk ch of row `row` of a [rows][256 B] image (the layout of the X tiles of
// k_mab0_attn_h4: the XOR term depends on row & 15 only, so any 16-aligned row block is an image)
__device__ __forceinline__ int y_off(int row, int ch) {
  return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
// B operand [k = point (k-slot order of pack8 of two score tiles)][col = feature 16 t + (lane & 15)] of
// the 32-row image `img`, through the transposing LDS read
__device__ __forceinline__ bf16x8 y_tr_frag(const char* img, int t, int lane) {
  const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
  const int a0 = y_off(4 * g + q, 2 * t + (p >> 1)) + 8 * (p & 1);
  const int a1 = y_off(16 + 4 * g + q, 2 * t + (p >> 1)) + 8 * (p & 1);
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + a0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + a1));
  const bf16x4 l4 = __builtin_bit_cast(bf16x4, lo), h4 = __builtin_bit_cast(bf16x4, hi);
  bf16x8 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) { r[e] = l4[e]; r[4 + e] = h4[e]; }
  return r;
}
__device__ __forceinline__ bf16x8 gload8(const __bf16* p) {
  return *reinterpret_cast<const bf16x8*>(p);
}
// accumulator tile (rows = features 16 t + 4 g + e, col = query r) -> [query][feature] bf16 image
__device__ __forceinline__ void put_tile(char* img, int t, int r, int g, f32x4 v) {
  *reinterpret_cast<bf16x4*>(img + swz(r, 2 * t + (g >> 1), ROWB) + 8 * (g & 1)) = pack4(v);
}

struct Ctx {
  int b, N, tid, lane, wave, q, j, r, g;
  int UPQ;          // 32-point units per quad (N / 128)
  int qn0;          // first point of this wave's quad
};

// ---------------------------------------------------------------------------------------------
// per-set stage between the two blocks of an ISAB (k_mid_fwd, mid_bf16.hip, on 16 waves):
//   A  O = Qp + T_h Wv_h^T + bv        (modules.py:29, reassociated: SURVEY 8d)     waves 0-7, tile t = wave
//   B  H = O + relu(O Wo^T + bo)       (modules.py:31)                              waves 0-7
//   C  Kp = H Wk^T + bk, Vp = H Wv^T + bv of the many-queries block (modules.py:21) all waves:
//      wave = (tile tc = wave >> 1, which = wave & 1), in the four bf16 images mab1 forward / backward read
// Every weight fragment a wave needs is requested by mid_prefetch() a phase ahead.
// ---------------------------------------------------------------------------------------------
struct MidPre {
  bf16x8 wa[4], wb[4], wc[4];
  float wvf[4][4];                 // SMALL: Wv0f[feature 16 t + 4 g + e][c]
  float4 qp, bv0, bo0, bkv;
  float bkvc;
};

template <bool SMALL>
__device__ __forceinline__ void mid_prefetch(const Set128Layer& L, const Ctx& c, int dk, MidPre& P) {
  const int r = c.r, g = c.g;
  if (c.wave < 8) {
    const int t = c.wave;
    P.qp = *reinterpret_cast<const float4*>(L.Qp0 + r * D + 16 * t + 4 * g);
    P.bv0 = *reinterpret_cast<const float4*>(L.bv0 + 16 * t + 4 * g);
    P.bo0 = *reinterpret_cast<const float4*>(L.bo0 + 16 * t + 4 * g);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (!SMALL) P.wa[ks] = gload8(L.Wv0 + (int64_t)(16 * t + r) * D + 32 * ks + 8 * g);
      P.wb[ks] = gload8(L.Wo0 + (int64_t)(16 * t + r) * D + 32 * ks + 8 * g);
    }
    if (SMALL) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
          P.wvf[e][cc] = cc < dk ? L.Wv0f[(16 * t + 4 * g + e) * dk + cc] : 0.f;
    }
  }
  {
    const int tc = c.wave >> 1, which = c.wave & 1;
    const __bf16* W = which ? L.Wv1 : L.Wk1;
    const float* bias = which ? L.bv1 : L.bk1;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) P.wc[ks] = gload8(W + (int64_t)(16 * tc + r) * D + 32 * ks + 8 * g);
    P.bkv = *reinterpret_cast<const float4*>(bias + 16 * tc + 4 * g);
    P.bkvc = bias[16 * tc + r];
  }
}

// sT: bf16 image [64][256 B] of the merged T (swz rows; !SMALL) / sTf: fp32 [64][4] (SMALL);
// sA, sH: [16][256 B] bf16 images of O and H.  Ends with a barrier: the K / V images are visible
// to the whole workgroup (global memory, same CU) when it returns.
template <bool SMALL>
__device__ __forceinline__ void mid_stage(const Set128Layer& L, const Ctx& c, const MidPre& P,
                                          const char* sT, const float* sTf, char* sA, char* sH) {
  const int r = c.r, g = c.g, b = c.b;
  f32x4 o = {0.f, 0.f, 0.f, 0.f};
  if (c.wave < 8) {                                   // ---- A
    const int t = c.wave, hh = t >> 1;
    o = f32x4{P.qp.x + P.bv0.x, P.qp.y + P.bv0.y, P.qp.z + P.bv0.z, P.qp.w + P.bv0.w};
    if (SMALL) {
      const float4 t4 = *reinterpret_cast<const float4*>(sTf + (16 * hh + r) * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        o[e] += t4.x * P.wvf[e][0] + t4.y * P.wvf[e][1] + t4.z * P.wvf[e][2] + t4.w * P.wvf[e][3];
    } else {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        o = mfma32(P.wa[ks], *reinterpret_cast<const bf16x8*>(sT + swz(16 * hh + r, 4 * ks + g, ROWB)), o);
    }
    put_tile(sA, t, r, g, o);
    *reinterpret_cast<float4*>(L.O0 + ((int64_t)b * MQ + r) * D + 16 * t + 4 * g) =
        float4{o[0], o[1], o[2], o[3]};
  }
  __syncthreads();
  if (c.wave < 8) {                                   // ---- B
    const int t = c.wave;
    f32x4 z = f32x4{P.bo0.x, P.bo0.y, P.bo0.z, P.bo0.w};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      z = mfma32(P.wb[ks], *reinterpret_cast<const bf16x8*>(sA + swz(r, 4 * ks + g, ROWB)), z);
    f32x4 hq;
#pragma unroll
    for (int e = 0; e < 4; ++e) hq[e] = o[e] + fmaxf(z[e], 0.f);
    const int64_t off = ((int64_t)b * MQ + r) * D + 16 * t + 4 * g;
    *reinterpret_cast<float4*>(L.Z0 + off) = float4{z[0], z[1], z[2], z[3]};
    *reinterpret_cast<float4*>(L.H + off) = float4{hq[0], hq[1], hq[2], hq[3]};
    put_tile(sH, t, r, g, hq);
  }
  __syncthreads();
  {                                                   // ---- C
    const int tc = c.wave >> 1, which = c.wave & 1;
    __bf16* PP = which ? L.VpP : L.KpP;
    __bf16* TT = which ? L.Vt : L.Kt;
    f32x4 fr = f32x4{P.bkv.x, P.bkv.y, P.bkv.z, P.bkv.w};     // rows = features, col = key
    f32x4 kr = f32x4{P.bkvc, P.bkvc, P.bkvc, P.bkvc};         // rows = keys,     col = feature
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8 hf = *reinterpret_cast<const bf16x8*>(sH + swz(r, 4 * ks + g, ROWB));
      fr = mfma32(P.wc[ks], hf, fr);
      kr = mfma32(hf, P.wc[ks], kr);
    }
    // keys 4g .. 4g+3 of feature 16 tc + r: 8 contiguous bytes of the [feature][key] image
    *reinterpret_cast<bf16x4*>(TT + ((int64_t)b * D + 16 * tc + r) * MQ + 4 * g) = pack4(kr);
    // features 32 (tc >> 1) + perm32(8 g + 4 (tc & 1) + e) of key r: half of the lane's 16 bytes of the
    // K-permuted image (k_mid_fwd writes pack8 of the head's two tiles)
    *reinterpret_cast<bf16x4*>(PP + ((int64_t)b * MQ + r) * D + 32 * (tc >> 1) + 8 * g + 4 * (tc & 1)) =
        pack4(fr);
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// many-queries block mab1(X, H) (modules.py:19-33) on the set's rows in sY; wave = (quad, head):
//   Qp_h = fc_q(x) ; A = softmax(Qp_h Kp_h^T / sqrt d) ; O_h = Qp_h + A Vp_h        per unit, in registers
//   barrier ; O slices -> sY (in place) ; barrier ; Z_h = fc_o(O) ; Y_h = O_h + relu(Z_h)
//   barrier ; Y slices -> sY ; barrier ; coalesced stores of the saved O and of Y from sY
// SMALL: layer 1 (dq = din <= 4: fc_q on the vector ALU from the points in sX, nothing in sY yet)
// ---------------------------------------------------------------------------------------------
template <bool SMALL>
__device__ __forceinline__ void mab1_phase(const Set128Layer& L, const Ctx& c, char* sY,
                                           const float* sX, int dq, float scale_log2e) {
  const int r = c.r, g = c.g, j = c.j, b = c.b, N = c.N;
  constexpr int KS = 4;
  // this head's weight slices as A operands [row = feature 32 j + 16 t + r][k-slots 32 s + 8 g ..]
  bf16x8 wa[KS][2];
  float wqs[2][4][4];
  if (!SMALL) {
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int t = 0; t < 2; ++t)
        wa[s][t] = gload8(L.WqB + (int64_t)(32 * j + 16 * t + r) * D + 32 * s + 8 * g);
  } else {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
          wqs[t][e][cc] = cc < dq ? L.WqF[(32 * j + 16 * t + 4 * g + e) * dq + cc] : 0.f;
  }
  f32x4 bqv[2], bov[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const float4 q4 = *reinterpret_cast<const float4*>(L.bq1 + 32 * j + 16 * t + 4 * g);
    const float4 o4 = *reinterpret_cast<const float4*>(L.bo1 + 32 * j + 16 * t + 4 * g);
    bqv[t] = f32x4{q4.x, q4.y, q4.z, q4.w};
    bov[t] = f32x4{o4.x, o4.y, o4.z, o4.w};
  }
  // the head's 16 keys: A operand [key r][k-slots 8 g .. of the head's 32 features] and
  // V^T: A operand [feature 16 t + r][keys 4 g ..]
  const bf16x8 kpa = gload8(L.KpP + ((int64_t)b * MQ + r) * D + 32 * j + 8 * g);
  bf16x4 vta[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
    vta[t] = *reinterpret_cast<const bf16x4*>(L.Vt + ((int64_t)b * D + 32 * j + 16 * t + r) * MQ + 4 * g);

  // per-lane byte offsets inside a 16-row block of sY (y_off depends on row & 15 only)
  int oB[KS], oD[2], oP[KS][2];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    oB[s] = y_off(r, 4 * s + g);
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) oP[s][hh] = y_off(r, 4 * s + 2 * hh + (g >> 1)) + 8 * (g & 1);
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) oD[t] = y_off(r, 4 * j + 2 * t + (g >> 1)) + 8 * (g & 1);

  bf16x4 opk[4][2][2];                // [unit][t][nb]: the wave's O slices, then its Y slices
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    if (u >= c.UPQ) break;
    const int n0 = c.qn0 + 32 * u;
    f32x4 acc[2][2];
    if (SMALL) {
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const float4 x4 = *reinterpret_cast<const float4*>(sX + (n0 + 16 * nb + r) * 4);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc[t][nb][e] = bqv[t][e] + wqs[t][e][0] * x4.x + wqs[t][e][1] * x4.y +
                            wqs[t][e][2] * x4.z + wqs[t][e][3] * x4.w;
      }
    } else {
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        acc[0][nb] = bqv[0];
        acc[1][nb] = bqv[1];
      }
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          const bf16x8 bx = *reinterpret_cast<const bf16x8*>(sY + (n0 + 16 * nb) * ROWB + oB[s]);
          acc[0][nb] = mfma32(wa[s][0], bx, acc[0][nb]);
          acc[1][nb] = mfma32(wa[s][1], bx, acc[1][nb]);
        }
      // saved for the backward: straight from the accumulators (8-byte pieces of 16 rows)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int t = 0; t < 2; ++t)
          *reinterpret_cast<bf16x4*>(L.QpS + ((int64_t)b * N + n0 + 16 * nb + r) * D + 32 * j + 16 * t +
                                     4 * g) = pack4(acc[t][nb]);
    }
    // attention over the 16 inducing keys, all inside the wave
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const bf16x8 qb = pack8(acc[0][nb], acc[1][nb]);
      const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
      f32x4 s0 = mfma32(kpa, qb, z4);            // [key 4 g + e][point r]
      float mx = fmaxf(fmaxf(s0[0], s0[1]), fmaxf(s0[2], s0[3]));
      mx = wave16_max(mx);
      float sum = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s0[e] = __builtin_amdgcn_exp2f((s0[e] - mx) * scale_log2e);
        sum += s0[e];
      }
      sum = wave16_sum(sum);
      const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
      for (int e = 0; e < 4; ++e) s0[e] *= inv;
      const bf16x4 pb = pack4(s0);               // B operand [k = key 4 g + e][point r]
      acc[0][nb] = mfma16(vta[0], pb, acc[0][nb]);
      acc[1][nb] = mfma16(vta[1], pb, acc[1][nb]);
    }
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int t = 0; t < 2; ++t) opk[u][t][nb] = pack4(acc[t][nb]);
  }
  // fc_o's weight slice replaces fc_q's (K-permuted image: the O tiles come back from sY in
  // accumulator order); requested before the barriers
#pragma unroll
  for (int s = 0; s < KS; ++s)
#pragma unroll
    for (int t = 0; t < 2; ++t)
      wa[s][t] = gload8(L.WoP + (int64_t)(32 * j + 16 * t + r) * D + 32 * s + 8 * g);
  if (!SMALL) __syncthreads();          // every head has read the unit's input rows
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    if (u >= c.UPQ) break;
    const int n0 = c.qn0 + 32 * u;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int t = 0; t < 2; ++t)
        *reinterpret_cast<bf16x4*>(sY + (n0 + 16 * nb) * ROWB + oD[t]) = opk[u][t][nb];
  }
  __syncthreads();                      // O rows complete
  // saved O: coalesced 16-byte pieces of full rows, from sY
  for (int p = c.tid; p < N * 16; p += NT) {
    const int row = p >> 4, ch = p & 15;
    *reinterpret_cast<uint4*>(L.OS + ((int64_t)b * N + row) * D + ch * 8) =
        *reinterpret_cast<const uint4*>(sY + y_off(row, ch));
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    if (u >= c.UPQ) break;
    const int n0 = c.qn0 + 32 * u;
    f32x4 acc[2][2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      acc[0][nb] = bov[0];
      acc[1][nb] = bov[1];
    }
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const char* rowp = sY + (n0 + 16 * nb) * ROWB;
        const bf16x4 lo = *reinterpret_cast<const bf16x4*>(rowp + oP[s][0]);
        const bf16x4 hi = *reinterpret_cast<const bf16x4*>(rowp + oP[s][1]);
        bf16x8 ob;
#pragma unroll
        for (int e = 0; e < 4; ++e) { ob[e] = lo[e]; ob[4 + e] = hi[e]; }
        acc[0][nb] = mfma32(wa[s][0], ob, acc[0][nb]);
        acc[1][nb] = mfma32(wa[s][1], ob, acc[1][nb]);
      }
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      uint32_t bits = 0u;                // byte j of the lane's mask word: feature tiles 2 j, 2 j + 1
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const bf16x4 o4 = opk[u][t][nb];
        f32x4 y;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float zz = acc[t][nb][e];
          y[e] = (float)o4[e] + fmaxf(zz, 0.f);
          if (zz > 0.f) bits |= 1u << (4 * t + e);
        }
        opk[u][t][nb] = pack4(y);
      }
      const int64_t blk = ((int64_t)b * N + n0) / 16 + nb;
      reinterpret_cast<uint8_t*>(L.mask)[(blk * 64 + c.lane) * 4 + j] = (uint8_t)bits;
    }
  }
  __syncthreads();                      // every head has read the O rows (and they are stored)
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    if (u >= c.UPQ) break;
    const int n0 = c.qn0 + 32 * u;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int t = 0; t < 2; ++t)
        *reinterpret_cast<bf16x4*>(sY + (n0 + 16 * nb) * ROWB + oD[t]) = opk[u][t][nb];
  }
  __syncthreads();                      // Y rows complete
  for (int p = c.tid; p < N * 16; p += NT) {
    const int row = p >> 4, ch = p & 15;
    *reinterpret_cast<uint4*>(L.Y + ((int64_t)b * N + row) * D + ch * 8) =
        *reinterpret_cast<const uint4*>(sY + y_off(row, ch));
  }
}

__global__ __launch_bounds__(NT) void k_set128_fwd(const Set128FwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sY = smem;                               // [512][256 B]
  char* sS = smem + 512 * ROWB;                  // 32 KiB of phase scratch
  float* sX = reinterpret_cast<float*>(sS);      // [N][4] the set's points (layer 1)            0 ..  8 K
  char* sA = sS + 8192;                          // O image of the mid stage                     8 .. 12 K
  char* sH = sS + 12288;                         // H image                                     12 .. 16 K
  char* sT = sS + 16384;                         // bf16 image of the merged T (layer 2)        16 .. 32 K
  float* sTf = reinterpret_cast<float*>(sT);     // fp32 [64][4] (layer 1)
  float* sAl = reinterpret_cast<float*>(sA);     // layer-2 attention: 16 alphas per wave (1 KiB)

  Ctx c;
  c.b = blockIdx.x; c.N = a.N; c.tid = threadIdx.x; c.lane = c.tid & 63;
  c.wave = __builtin_amdgcn_readfirstlane(c.tid >> 6);
  c.q = c.wave >> 2; c.j = c.wave & 3; c.r = c.lane & 15; c.g = c.lane >> 4;
  c.UPQ = a.N >> 7; c.qn0 = c.q * (a.N >> 2);
  const int b = c.b, N = a.N, dk = a.din, tid = c.tid, lane = c.lane, r = c.r, g = c.g;

  // ================= layer 1, few-queries block: scores G x, online softmax, T = A x =============
  // (k_mab0_attn_small's arithmetic: exact fp32 on the vector ALU.)  lane = score row (head * 16 +
  // query), wave w walks the points [w N/16, (w+1) N/16) - their coordinates are wave-uniform LDS
  // broadcasts - and the 16 partial (m, l, t) per row are merged by wave 0.
  MidPre pre;
  mid_prefetch<true>(a.L[0], c, dk, pre);
  {
    for (int i = tid; i < N * 4; i += NT) {
      const int pt = i >> 2, cc = i & 3;
      sX[i] = cc < dk ? a.X[((int64_t)b * N + pt) * dk + cc] : 0.f;
    }
    float gk[4];
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) gk[cc] = cc < dk ? a.L[0].Gf[lane * dk + cc] : 0.f;
    __syncthreads();
    const int pw = N >> 4, p0 = c.wave * pw;       // 8 .. 32 points per wave
    float m = -INFINITY;
#pragma unroll 8
    for (int i = 0; i < pw; ++i) {
      const float4 x4 = *reinterpret_cast<const float4*>(sX + (p0 + i) * 4);
      m = fmaxf(m, gk[0] * x4.x + gk[1] * x4.y + gk[2] * x4.z + gk[3] * x4.w);
    }
    float l = 0.f, t4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
    for (int i = 0; i < pw; ++i) {
      const float4 x4 = *reinterpret_cast<const float4*>(sX + (p0 + i) * 4);
      const float p = __builtin_amdgcn_exp2f(gk[0] * x4.x + gk[1] * x4.y + gk[2] * x4.z + gk[3] * x4.w - m);
      l += p;
      t4[0] += p * x4.x; t4[1] += p * x4.y; t4[2] += p * x4.z; t4[3] += p * x4.w;
    }
    float* sM = reinterpret_cast<float*>(sS + 8192);       // [16][64]
    float* sL = reinterpret_cast<float*>(sS + 12288);      // [16][64]
    float* sTp = reinterpret_cast<float*>(sS + 16384);     // [16][64][4]
    sM[c.wave * 64 + lane] = m;
    sL[c.wave * 64 + lane] = l;
    *reinterpret_cast<float4*>(sTp + (c.wave * 64 + lane) * 4) = float4{t4[0], t4[1], t4[2], t4[3]};
    __syncthreads();
    float tt[4] = {0.f, 0.f, 0.f, 0.f};
    if (c.wave == 0) {
      float M = -INFINITY;
#pragma unroll
      for (int p = 0; p < 16; ++p) M = fmaxf(M, sM[p * 64 + lane]);
      float Ls = 0.f;
#pragma unroll
      for (int p = 0; p < 16; ++p) {
        const float f = __builtin_amdgcn_exp2f(sM[p * 64 + lane] - M);
        const float4 tp = *reinterpret_cast<const float4*>(sTp + (p * 64 + lane) * 4);
        Ls += sL[p * 64 + lane] * f;
        tt[0] += tp.x * f; tt[1] += tp.y * f; tt[2] += tp.z * f; tt[3] += tp.w * f;
      }
      const float inv = 1.f / Ls;
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) tt[cc] *= inv;
      for (int cc = 0; cc < dk; ++cc) a.L[0].T[((int64_t)b * 64 + lane) * dk + cc] = tt[cc];
      a.L[0].LSE[(int64_t)b * 64 + lane] = M + log2f(Ls);
    }
    __syncthreads();                      // the partial buffers are dead: sTf aliases them
    if (c.wave == 0) *reinterpret_cast<float4*>(sTf + lane * 4) = float4{tt[0], tt[1], tt[2], tt[3]};
    __syncthreads();
  }
  mid_stage<true>(a.L[0], c, pre, sT, sTf, sA, sH);

  // ================= layer 1, many-queries block ================================================
  mab1_phase<true>(a.L[0], c, sY, sX, dk, a.scale_log2e);

  // ================= layer 2, few-queries block over the rows in sY (k_mab0_attn_h4) ==============
  {
    const Set128Layer& L = a.L[1];
    bf16x8 gf[4];                     // this head's G rows: B operand of the score MFMAs
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) gf[ks] = gload8(L.Gb + (int64_t)(16 * c.j + r) * D + 32 * ks + 8 * g);
    float* myAl = sAl + c.wave * 16;
    float mrow = -INFINITY, lrow = 0.f;
    f32x4 T[8];
#pragma unroll
    for (int ft = 0; ft < 8; ++ft) T[ft] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int u = 0; u < c.UPQ; ++u) {
      const char* img = sY + (c.qn0 + 32 * u) * ROWB;
      f32x4 s[2];
#pragma unroll
      for (int pb = 0; pb < 2; ++pb) {
        s[pb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          s[pb] = mfma32(*reinterpret_cast<const bf16x8*>(img + y_off(16 * pb + r, 4 * ks + g)), gf[ks],
                         s[pb]);
      }
      // rows of s = points 16 pb + 4 g + e ; column = query row r of this head
      float mt = -INFINITY;
#pragma unroll
      for (int pb = 0; pb < 2; ++pb)
#pragma unroll
        for (int e = 0; e < 4; ++e) mt = fmaxf(mt, s[pb][e]);
      mt = wave16_max(mt);
      const float mnew = fmaxf(mrow, mt);
      const float alpha = exp2f(mrow - mnew);
      float ls = 0.f;
#pragma unroll
      for (int pb = 0; pb < 2; ++pb)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s[pb][e] = exp2f(s[pb][e] - mnew);
          ls += s[pb][e];
        }
      ls = wave16_sum(ls);
      lrow = lrow * alpha + ls;
      mrow = mnew;
      if (g == 0) myAl[r] = alpha;                    // T rows are query rows 4g+e
      const float4 a4 = *reinterpret_cast<const float4*>(&myAl[4 * g]);
      const bf16x8 pa = pack8(s[0], s[1]);
#pragma unroll
      for (int ft = 0; ft < 8; ++ft) {
        T[ft][0] *= a4.x; T[ft][1] *= a4.y; T[ft][2] *= a4.z; T[ft][3] *= a4.w;
        T[ft] = mfma32(pa, y_tr_frag(img, ft, lane), T[ft]);
      }
    }
    // the four quads' partials of a head meet in global memory (L2: the workgroup's own lines);
    // 8 KiB per wave - there is no room for them next to sY
    mid_prefetch<false>(L, c, D, pre);
    const int64_t pbase = ((int64_t)b * 4 + c.q) * 64 + 16 * c.j;
#pragma unroll
    for (int ft = 0; ft < 8; ++ft)
#pragma unroll
      for (int e = 0; e < 4; ++e) a.Tp2[(pbase + 4 * g + e) * D + 16 * ft + r] = T[ft][e];
    if (g == 0) {
      a.Mp2[pbase + r] = mrow;
      a.Lp2[pbase + r] = lrow;
    }
    __syncthreads();
    {                                  // merge -> T (global fp32, saved) and its bf16 image
      const int row = tid >> 4, ch = tid & 15;
      float msv[4], lpv[4];
      float4 lo[4], hi[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int64_t o = ((int64_t)b * 4 + s) * 64 + row;
        msv[s] = a.Mp2[o];
        lpv[s] = a.Lp2[o];
        const float4* tp = reinterpret_cast<const float4*>(a.Tp2 + o * D + ch * 8);
        lo[s] = tp[0];
        hi[s] = tp[1];
      }
      const float M = fmaxf(fmaxf(msv[0], msv[1]), fmaxf(msv[2], msv[3]));
      float Ls = 0.f, t[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float fs = exp2f(msv[s] - M);
        Ls += fs * lpv[s];
        t[0] += fs * lo[s].x; t[1] += fs * lo[s].y; t[2] += fs * lo[s].z; t[3] += fs * lo[s].w;
        t[4] += fs * hi[s].x; t[5] += fs * hi[s].y; t[6] += fs * hi[s].z; t[7] += fs * hi[s].w;
      }
      const float inv = 1.f / Ls;
      bf16x8 v;
#pragma unroll
      for (int k = 0; k < 8; ++k) { t[k] *= inv; v[k] = (__bf16)t[k]; }
      float4* tg = reinterpret_cast<float4*>(L.T + ((int64_t)b * 64 + row) * D + ch * 8);
      tg[0] = float4{t[0], t[1], t[2], t[3]};
      tg[1] = float4{t[4], t[5], t[6], t[7]};
      *reinterpret_cast<bf16x8*>(sT + swz(row, ch, ROWB)) = v;
      if (ch == 0) L.LSE[(int64_t)b * 64 + row] = M + log2f(Ls);
    }
    __syncthreads();
    mid_stage<false>(L, c, pre, sT, sTf, sA, sH);
  }

  // ================= layer 2, many-queries block ================================================
  mab1_phase<false>(a.L[1], c, sY, sX, D, a.scale_log2e);

  // ================= PMA attention partials over Y2 (k_mab0_attn<1>): wave (q, j) = unit j of quad q ===
  {
    float mrow = -INFINITY, lrow = 0.f;
    f32x4 T[8];
#pragma unroll
    for (int ft = 0; ft < 8; ++ft) T[ft] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (c.j < c.UPQ) {
      const char* img = sY + (c.qn0 + 32 * c.j) * ROWB;
      f32x4 s[2];
#pragma unroll
      for (int pb = 0; pb < 2; ++pb) {
        s[pb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          s[pb] = mfma32(*reinterpret_cast<const bf16x8*>(img + y_off(16 * pb + r, 4 * ks + g)),
                         gload8(a.Gpma + (int64_t)r * D + 32 * ks + 8 * g), s[pb]);
      }
      float mt = -INFINITY;
#pragma unroll
      for (int pb = 0; pb < 2; ++pb)
#pragma unroll
        for (int e = 0; e < 4; ++e) mt = fmaxf(mt, s[pb][e]);
      mrow = wave16_max(mt);
      float ls = 0.f;
#pragma unroll
      for (int pb = 0; pb < 2; ++pb)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s[pb][e] = exp2f(s[pb][e] - mrow);
          ls += s[pb][e];
        }
      lrow = wave16_sum(ls);
      const bf16x8 pa = pack8(s[0], s[1]);
#pragma unroll
      for (int ft = 0; ft < 8; ++ft) T[ft] = mfma32(pa, y_tr_frag(img, ft, lane), T[ft]);
    }
    __syncthreads();                   // sY is dead (its rows are on their way to memory): the slabs alias it
    float* slab = reinterpret_cast<float*>(sY) + c.wave * 520;     // [4 rows][128] + m[4] + l[4]
    if (g == 0) {
#pragma unroll
      for (int ft = 0; ft < 8; ++ft)
#pragma unroll
        for (int e = 0; e < 4; ++e) slab[e * D + 16 * ft + r] = T[ft][e];
      if (r < 4) {
        slab[512 + r] = mrow;
        slab[516 + r] = lrow;
      }
    }
    __syncthreads();
    const int Sp = a.Sp, qpp = 4 / Sp;           // quads per partial
    if (tid < Sp * 512) {
      const int p = tid >> 9, rr = (tid >> 7) & 3, f = tid & 127;
      const float* s0 = reinterpret_cast<const float*>(sY);
      float M = -INFINITY;
      for (int w = p * qpp * 4; w < (p + 1) * qpp * 4; ++w) M = fmaxf(M, s0[w * 520 + 512 + rr]);
      float Ls = 0.f, t = 0.f;
      for (int w = p * qpp * 4; w < (p + 1) * qpp * 4; ++w) {
        const float mw = s0[w * 520 + 512 + rr];
        if (mw == -INFINITY) continue;
        const float fs = exp2f(mw - M);
        Ls += fs * s0[w * 520 + 516 + rr];
        t += fs * s0[w * 520 + rr * D + f];
      }
      const int64_t o = ((int64_t)b * Sp + p) * 4 + rr;
      a.TpP[o * D + f] = t;
      if (f == 0) {
        a.MpP[o] = M;
        a.LpP[o] = Ls;
      }
    }
  }
}

}  // namespace

bool set128_shape_ok(int N, int din, int d, int h, int m, int k) {
  return d == 128 && h == 4 && m == 16 && k == 1 && din >= 1 && din <= 4 && N % 128 == 0 && N >= 128 &&
         N <= 512;
}

size_t set128_fwd_ws_bytes(int B) {
  return align256((size_t)B * 4 * 64 * 128 * sizeof(float)) + 2 * align256((size_t)B * 4 * 64 * sizeof(float));
}

int set128_fwd_launch(const Set128FwdArgs& a, hipStream_t st) {
  PCA_REQUIRE(set128_shape_ok(a.N, a.din, 128, 4, 16, 1), "set128_fwd: N=%d din=%d not built", a.N, a.din);
  PCA_REQUIRE(a.Sp == 1 || a.Sp == 2 || a.Sp == 4, "set128_fwd: %d PMA partials", a.Sp);
  static std::once_flag once;
  std::call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_set128_fwd),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  const size_t lds = (size_t)512 * ROWB + 32768;
  // reference-formulation FLOPs of the three blocks this launch covers (the PMA's
  // epilogue and the classifier run in k_pma_head1), algorithmic bytes: X in, two [N, d] bf16 tensors
  // written and read once each
  const double Nn = a.N, dd = 128, mm = 16;
  const double macs = Nn * (3.0 * a.din * dd + 7.0 * dd * dd + 8.0 * mm * dd + 2.0 * dd) + 6.0 * mm * dd * dd;
  ProfScope ps(PCA_K_SET_FWD, st, 2.0 * macs * a.B, (double)a.B * (4.0 * Nn * a.din + 8.0 * Nn * dd));
  hipLaunchKernelGGL(k_set128_fwd, dim3(a.B), dim3(NT), lds, st, a);
  ps.end();
  return check_launch("k_set128_fwd");
}

}  // namespace pca
