// Fused bf16-MFMA backward of ISAB's mab1(X, H) (adjoint of set_transformer-master/
// modules.py:19-33 in the "many queries, few keys" orientation; math in SURVEY.md 3c).
//
//   k_mab1_bwd   per 32 points of one wave, all in registers in the transposed layout:
//                dZ = dY.mask ; dO^T = dY^T + Wo^T.dZ^T ; per head: P^T recomputed from the
//                saved Qp, dA^T = Vp_h.dO_h^T, dS^T = P^T.(dA^T - rowsum) / sqrt(d),
//                dQp_h^T = dO_h^T + Kp_h^T.dS^T ; dX^T = Wq^T.dQp^T.
//                Writes dX and bf16 copies of dZ, dO, dQp, dS, P for the reductions over
//                points, which need the point index on the MFMA K axis:
//   k_wgrad      dW[DG x DA] += G^T.A over a row range (G = dZ|dQp, A = O|X): both operands
//                come from row-major [point][feature] LDS images through ds_read_tr16_b64
//                (hardware transpose), fp32 partials added with atomics; column sums of G
//                (bias gradients) ride along.
//   k_kv_grad    per (set, head): dKp = dS^T.Qp, dVp = P^T.dO (m x 32 outputs, N-long sums).
// The tiny [B*m, d] projections of H (fc_k, fc_v) are differentiated with the fp32 GEMMs.
#include "mab1_bf16.hpp"
#include "terminal_bodies.hpp"
#include "slab_sum_body.hpp"

#include <math.h>
#include <stdlib.h>

#include <mutex>

namespace pca {


namespace {

constexpr int TP = M1_TP;
constexpr int NB = M1_NB;

struct Mab1BwdArgs {
  const void* dY;           // [B, N, D] fp32, or bf16 when ABF
  const __bf16* QpS;        // [B*N][D]
  const uint32_t* mask;
  const __bf16 *KpP, *VpP;  // [B][MI][D]  (K-permuted features)
  const __bf16* Kt;         // [B][D][MI]
  const __bf16 *WoTP, *WqTP;
  __bf16 *dZ, *dQp, *dOs;   // [B*N][D]
  __bf16 *dS, *P;           // [B*N][H*MI]
  void* dX;                 // [B, N, D] (fp32 / bf16 when ABF) or null
  float *dKpG, *dVpG;       // [B][nparts][MI][D] fp32 partial K/V gradients (fused mode)
  const float* Xs;          // layer 1 (dq <= 3): the fp32 points [B, N, dq] ...
  float *dWqS, *dbqS;       // ... and fc_q gradients accumulated here (fused reduction)
  float* wq_slab;           // ... or, when set, per-workgroup partials [wg][D*dq + D] (fixed-order sum later)
  const float *WqF, *bqF;   // ... and fc_q itself: Qp is recomputed (2..3 FMAs per element)
                            //     instead of being saved by the forward and read back
  int dq;
  float* zero_ptr;          // optional: zero_n floats cleared by this launch (consumer's
  int zero_n;               //           accumulator, e.g. the dQs of k_mid_bwd)
  long long* dbg;           // PCA_DEBUG_CLOCKS builds: phase time stamps of workgroup 0
  int B, N, tiles_per_set;
  int tpw;                  // consecutive tiles of ONE set per workgroup (fused mode)
  float scale, scale_log2e;
  int dbg_wg;               // PCA_DEBUG_CLOCKS builds: the workgroup whose phase stamps are kept
};

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// transposed fragment (k = the 32 points of a wave tile) from a small row-major bf16 image
// with `rb` bytes per row, for the 16 columns starting at col0
__device__ __forceinline__ bf16x8 tr_frag_small(const char* img, int rb, int col0, int lane) {
  const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
  const int a0 = (4 * g + q) * rb + (col0 + 4 * p) * 2;
  const int a1 = (16 + 4 * g + q) * rb + (col0 + 4 * p) * 2;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + a0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + a1));
  const bf16x4 l4 = __builtin_bit_cast(bf16x4, lo), h4 = __builtin_bit_cast(bf16x4, hi);
  bf16x8 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) { r[e] = l4[e]; r[4 + e] = h4[e]; }
  return r;
}

#ifdef PCA_DEBUG_CLOCKS
// (kept in scalar registers and written once at the end: a store per stamp would sit in vmcnt and be
//  waited for by the next phase's loads)
#define PCA_STAMP(i) do { stamp_[i] = wall_clock64(); } while (0)
#else
#define PCA_STAMP(i) do {} while (0)
#endif
// Waves per workgroup: 8 for the d -> d variant with dX (its 100+ KiB of LDS images allow one
// workgroup per CU; eight waves sharing them give every SIMD two wavefronts to interleave, and
// waves 4..7 take the next 128-point tile), otherwise 4 with two workgroups per CU.
template <bool WANT_DX, bool FUSE_KV>
struct BwdWaves {
  static constexpr int value = (WANT_DX && FUSE_KV) ? 8 : 4;
  static constexpr int per_cu = WANT_DX ? 1 : 2;        // workgroups per CU
};

template <int D, int MI, bool WANT_DX, bool FUSE_KV, bool FUSE_WQ, bool ABF>
__global__ __launch_bounds__((64 * BwdWaves<WANT_DX, FUSE_KV>::value),
                             (BwdWaves<WANT_DX, FUSE_KV>::per_cu))
void k_mab1_bwd(const Mab1BwdArgs a) {
  constexpr int NW = BwdWaves<WANT_DX, FUSE_KV>::value, NT = 64 * NW, SUBS = NW / 4;
  constexpr int DT = D / 16, KS = D / 32, ROWB = D * 2, HM = KS * MI;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sWoT = smem;
  char* sKp = sWoT + D * ROWB;
  char* sVp = sKp + MI * ROWB;
  char* sKt = sVp + MI * ROWB;
  char* sWqT = sKt + D * MI * 2;
  // fused K/V gradients: per wave [32][16] dS, [32][16] P, [32][32] Qp_j, [32][32] dO_j images
  char* sKV = sWqT + (WANT_DX ? D * ROWB : 0);

  const int tid = threadIdx.x, lane = tid & 63, wave8 = tid >> 6;
  const int wave = wave8 & 3, sub = wave8 >> 2;
  const int r = lane & 15, g = lane >> 4;
#ifdef PCA_DEBUG_CLOCKS
  long long stamp_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  PCA_STAMP(0);

  {
    // all chunks of this thread first, then the LDS stores (one round trip, not one per chunk)
    constexpr int NC = D * (D / 8) / NT;
    uint4 wo[NC], wq[WANT_DX ? NC : 1];
#pragma unroll
    for (int e = 0; e < NC; ++e) {
      const int c = tid + NT * e, row = c / (D / 8), c16 = c % (D / 8);
      wo[e] = *reinterpret_cast<const uint4*>(a.WoTP + (int64_t)row * D + c16 * 8);
      if (WANT_DX) wq[e] = *reinterpret_cast<const uint4*>(a.WqTP + (int64_t)row * D + c16 * 8);
    }
#pragma unroll
    for (int e = 0; e < NC; ++e) {
      const int c = tid + NT * e, row = c / (D / 8), c16 = c % (D / 8);
      *reinterpret_cast<uint4*>(sWoT + swz(row, c16, ROWB)) = wo[e];
      if (WANT_DX) *reinterpret_cast<uint4*>(sWqT + swz(row, c16, ROWB)) = wq[e];
    }
  }

  PCA_STAMP(8);
  const int total_tiles = a.B * a.tiles_per_set;
  int cur_b = -1;
  if (a.zero_ptr != nullptr)
    for (int i = blockIdx.x * NT + tid; i < a.zero_n; i += gridDim.x * NT) a.zero_ptr[i] = 0.f;
  // row pitches padded by 8 bytes (40 / 72 instead of 32 / 64): with power-of-two pitches the
  // 16 point rows of a store or transposed read fall on 2 resp. 4 LDS banks repeatedly
  // (SQ_LDS_BANK_CONFLICT was 3x the LDS-active cycles of this kernel); 10 / 18 banks per row
  // spread them over all 64
  constexpr int PS = 40, PQ = 72, KVB = 2 * 32 * PS + 2 * 32 * PQ;
  // layer 1: fc_q rows as float4 (w0, w1, w2, bias) behind the per-wave images
  float4* sWq4 = reinterpret_cast<float4*>(sKV + NW * KVB);
  if (FUSE_WQ) {
    for (int f = tid; f < D; f += NT)
      sWq4[f] = float4{a.WqF[f * a.dq], a.dq > 1 ? a.WqF[f * a.dq + 1] : 0.f,
                       a.dq > 2 ? a.WqF[f * a.dq + 2] : 0.f, a.bqF[f]};
  }
  char* myDS = sKV + wave8 * KVB;
  char* myP = myDS + 32 * PS;
  char* myQ = myP + 32 * PS;
  char* myO = myQ + 32 * PQ;
  f32x4 dkp[FUSE_KV ? KS : 1][2], dvp[FUSE_KV ? KS : 1][2];
  if (FUSE_KV) {
#pragma unroll
    for (int j = 0; j < KS; ++j)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        dkp[j][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
        dvp[j][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
  }
  // layer 1: fc_q gradients on the matrix core.  wqa[j][tt] = Xa . dQp_j with the augmented
  // point matrix Xa[16][32 points]: rows 0..2 = bf16 high parts of x, row 3 = ones (bias),
  // rows 4..6 = bf16 low parts (x = hi + lo to ~2^-17), so that
  // dWq[f][c] = row c + row 4+c and dbq[f] = row 3 of column f.
  f32x4 wqa[FUSE_WQ ? KS : 1][2];
  if (FUSE_WQ) {
#pragma unroll
    for (int j = 0; j < KS; ++j)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) wqa[j][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // fused mode: a workgroup owns `tpw` consecutive tiles of one set; otherwise grid-stride
  const int t_first = FUSE_KV ? blockIdx.x * a.tpw : blockIdx.x;
  const int t_step = FUSE_KV ? 1 : gridDim.x;
  const int t_last = FUSE_KV ? t_first + a.tpw : total_tiles;
  // (fused mode with 8 waves: two consecutive tiles of the set per pass, one per wave quartet)
  for (int tile0 = t_first; tile0 < t_last && tile0 < total_tiles; tile0 += t_step * SUBS) {
    const int tile_id = tile0 + sub;
    const int b = tile0 / a.tiles_per_set, tile = tile_id - b * a.tiles_per_set;
    if (b != cur_b) {
      __syncthreads();
      for (int c = tid; c < MI * (D / 8); c += NT) {
        const int row = c / (D / 8), c16 = c % (D / 8);
        const int64_t src = ((int64_t)b * MI + row) * D + c16 * 8;
        *reinterpret_cast<uint4*>(sKp + swz(row, c16, ROWB)) =
            *reinterpret_cast<const uint4*>(a.KpP + src);
        *reinterpret_cast<uint4*>(sVp + swz(row, c16, ROWB)) =
            *reinterpret_cast<const uint4*>(a.VpP + src);
      }
      for (int c = tid; c < D * MI / 8; c += NT)
        reinterpret_cast<uint4*>(sKt)[c] =
            reinterpret_cast<const uint4*>(a.Kt + (int64_t)b * D * MI)[c];
      cur_b = b;
    }
    __syncthreads();
    PCA_STAMP(1);
    if (tile_id >= t_last || tile_id >= total_tiles) continue;   // odd count: waves 4..7 idle
                                                                 // (no barrier below)

    const int n_base = tile * TP + wave * 32;
    int nn[NB];
    bool live[NB];
    int64_t row[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      nn[nb] = n_base + 16 * nb + r;
      live[nb] = nn[nb] < a.N;
      row[nb] = (int64_t)b * a.N + (live[nb] ? nn[nb] : 0);
    }

    float xq[NB][3];                 // layer 1: this lane's points, for the Qp recomputation
    if (FUSE_WQ) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int c = 0; c < 3; ++c) xq[nb][c] = c < a.dq ? a.Xs[row[nb] * a.dq + c] : 0.f;
    }
    bf16x8 xaug;
    if (FUSE_WQ) {
      // A operand: lane (row c' = r, k-slots 8g..8g+7 <-> points perm32(8g + .)) of this tile
      const int c = r & 3, part = r >> 2;          // part 0: hi / ones, 1: lo
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int pt = n_base + (k < 4 ? 4 * g + k : 16 + 4 * g + k - 4);
        float v = 0.f;
        if (part <= 1 && r < 7 && r != 3 && c < a.dq) {
          const float x = a.Xs[((int64_t)b * a.N + (pt < a.N ? pt : 0)) * a.dq + c];
          const float hi = (float)(__bf16)x;
          v = part == 0 ? hi : x - hi;
        } else if (r == 3) {
          v = 1.f;
        }
        xaug[k] = (__bf16)v;
      }
    }
    // ---- dY^T tiles and dZ = dY . [Z > 0] ----
    f32x4 dO[DT][NB];
    bf16x8 dzb[KS][NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      uint32_t bits[D / 128];
#pragma unroll
      for (int w = 0; w < D / 128; ++w)
        bits[w] = a.mask[mab1_mask_index<D>(b, a.tiles_per_set, tile, wave, nb, w, lane)];
      f32x4 dz[DT];
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        // (unconditional - row[] is clamped for padding points - and zeroed afterwards: under the
        //  divergent `if (live)` each of the DT loads was waited for before the next was issued)
        float4 v;
        if (ABF) {
          const bf16x4 h4 = *reinterpret_cast<const bf16x4*>(
              reinterpret_cast<const __bf16*>(a.dY) + row[nb] * D + 16 * t + 4 * g);
          v = float4{(float)h4[0], (float)h4[1], (float)h4[2], (float)h4[3]};
        } else {
          v = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(a.dY) +
                                               row[nb] * D + 16 * t + 4 * g);
        }
        if (!live[nb]) v = float4{0.f, 0.f, 0.f, 0.f};
        dO[t][nb] = f32x4{v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
          dz[t][e] = ((bits[t / 8] >> ((t & 7) * 4 + e)) & 1u) ? dO[t][nb][e] : 0.f;
        if (live[nb] && a.dZ != nullptr)     // (null: the fc_o job masks dY itself, WgradJob::mask)
          *reinterpret_cast<bf16x4*>(a.dZ + row[nb] * D + 16 * t + 4 * g) = pack4(dz[t]);
      }
#pragma unroll
      for (int s = 0; s < KS; ++s) dzb[s][nb] = pack8(dz[2 * s], dz[2 * s + 1]);
    }

    PCA_STAMP(2);
    // ---- dO^T = dY^T + Wo^T . dZ^T ----
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        const bf16x8 wa =
            *reinterpret_cast<const bf16x8*>(sWoT + swz(16 * t + r, 4 * s + g, ROWB));
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) dO[t][nb] = mfma32(wa, dzb[s][nb], dO[t][nb]);
      }
    }
    if (!FUSE_KV) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
        if (live[nb]) {
#pragma unroll
          for (int t = 0; t < DT; ++t)
            *reinterpret_cast<bf16x4*>(a.dOs + row[nb] * D + 16 * t + 4 * g) = pack4(dO[t][nb]);
        }
    }

    PCA_STAMP(3);
    // ---- attention backward per head; dO tiles of the head become dQp in place ----
#pragma unroll
    for (int j = 0; j < KS; ++j) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        // saved Qp of this point, features 32j + perm32(8g + .)
        bf16x4 qlo, qhi;
        if (FUSE_WQ) {
          // Qp = x Wq^T + bq exactly as the forward computes it (fp32, then rounded to bf16)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float4 wl = sWq4[32 * j + 4 * g + e], wh = sWq4[32 * j + 16 + 4 * g + e];
            qlo[e] = (__bf16)(wl.w + wl.x * xq[nb][0] + wl.y * xq[nb][1] + wl.z * xq[nb][2] +
                              0.f * 0.f);
            qhi[e] = (__bf16)(wh.w + wh.x * xq[nb][0] + wh.y * xq[nb][1] + wh.z * xq[nb][2] +
                              0.f * 0.f);
          }
        } else {
          qlo = *reinterpret_cast<const bf16x4*>(a.QpS + row[nb] * D + 32 * j + 4 * g);
          qhi = *reinterpret_cast<const bf16x4*>(a.QpS + row[nb] * D + 32 * j + 16 + 4 * g);
        }
        bf16x8 qb;
#pragma unroll
        for (int e = 0; e < 4; ++e) { qb[e] = qlo[e]; qb[4 + e] = qhi[e]; }
        const bf16x8 dob = pack8(dO[2 * j][nb], dO[2 * j + 1][nb]);
        f32x4 p0 = {0.f, 0.f, 0.f, 0.f}, p1 = {0.f, 0.f, 0.f, 0.f};
        f32x4 da0 = {0.f, 0.f, 0.f, 0.f}, da1 = {0.f, 0.f, 0.f, 0.f};
        p0 = mfma32(*reinterpret_cast<const bf16x8*>(sKp + swz(r, 4 * j + g, ROWB)), qb, p0);
        da0 = mfma32(*reinterpret_cast<const bf16x8*>(sVp + swz(r, 4 * j + g, ROWB)), dob, da0);
        if (MI == 32) {
          p1 = mfma32(*reinterpret_cast<const bf16x8*>(sKp + swz(16 + r, 4 * j + g, ROWB)), qb,
                      p1);
          da1 = mfma32(*reinterpret_cast<const bf16x8*>(sVp + swz(16 + r, 4 * j + g, ROWB)), dob,
                       da1);
        }
        float mx = fmaxf(fmaxf(p0[0], p0[1]), fmaxf(p0[2], p0[3]));
        if (MI == 32) mx = fmaxf(mx, fmaxf(fmaxf(p1[0], p1[1]), fmaxf(p1[2], p1[3])));
        mx = wave16_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          p0[e] = exp2f((p0[e] - mx) * a.scale_log2e);
          sum += p0[e];
          if (MI == 32) {
            p1[e] = exp2f((p1[e] - mx) * a.scale_log2e);
            sum += p1[e];
          }
        }
        sum = wave16_sum(sum);
        const float inv = 1.0f / sum;
        float delta = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          p0[e] *= inv;
          p1[e] *= inv;
          delta += p0[e] * da0[e] + p1[e] * da1[e];
        }
        delta = wave16_sum(delta);
        f32x4 ds0, ds1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          ds0[e] = p0[e] * (da0[e] - delta) * a.scale;
          ds1[e] = p1[e] * (da1[e] - delta) * a.scale;
        }
        if (FUSE_KV) {
          // wave-private images [point][.] of this head: dS, P (16 keys), Qp_j, dO_j (32 feats);
          // padding points contribute zeros
          const int pt = 16 * nb + r;
          f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
          *reinterpret_cast<bf16x4*>(myDS + pt * PS + 8 * g) = pack4(live[nb] ? ds0 : zero4);
          *reinterpret_cast<bf16x4*>(myP + pt * PS + 8 * g) = pack4(live[nb] ? p0 : zero4);
          *reinterpret_cast<bf16x4*>(myQ + pt * PQ + 8 * g) = qlo;
          *reinterpret_cast<bf16x4*>(myQ + pt * PQ + 32 + 8 * g) = qhi;
          *reinterpret_cast<bf16x4*>(myO + pt * PQ + 8 * g) = pack4(dO[2 * j][nb]);
          *reinterpret_cast<bf16x4*>(myO + pt * PQ + 32 + 8 * g) = pack4(dO[2 * j + 1][nb]);
        } else if (live[nb]) {
          *reinterpret_cast<bf16x4*>(a.P + row[nb] * HM + j * MI + 4 * g) = pack4(p0);
          *reinterpret_cast<bf16x4*>(a.dS + row[nb] * HM + j * MI + 4 * g) = pack4(ds0);
          if (MI == 32) {
            *reinterpret_cast<bf16x4*>(a.P + row[nb] * HM + j * MI + 16 + 4 * g) = pack4(p1);
            *reinterpret_cast<bf16x4*>(a.dS + row[nb] * HM + j * MI + 16 + 4 * g) = pack4(ds1);
          }
        }
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          const int t = 2 * j + tt;
          const char* krow = sKt + (16 * t + r) * (MI * 2);
          if (MI == 16)
            dO[t][nb] = mfma16(*reinterpret_cast<const bf16x4*>(krow + 8 * g), pack4(ds0),
                               dO[t][nb]);
          else
            dO[t][nb] = mfma32(*reinterpret_cast<const bf16x8*>(krow + 16 * g), pack8(ds0, ds1),
                               dO[t][nb]);
        }
      }
      if (FUSE_KV) {
        // dKp_j[key][f] += sum_pt dS[key][pt] Qp[pt][f] ; dVp_j[key][f] += sum_pt P[key][pt] dO[pt][f]
        const bf16x8 ads = tr_frag_small(myDS, PS, 0, lane);
        const bf16x8 ap = tr_frag_small(myP, PS, 0, lane);
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          dkp[j][tt] = mfma32(ads, tr_frag_small(myQ, PQ, 16 * tt, lane), dkp[j][tt]);
          dvp[j][tt] = mfma32(ap, tr_frag_small(myO, PQ, 16 * tt, lane), dvp[j][tt]);
        }
        if (FUSE_WQ) {
          // the dO_j image is consumed: overwrite it with dQp_j (padding points carry zeros)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            const int pt = 16 * nb + r;
            *reinterpret_cast<bf16x4*>(myO + pt * PQ + 8 * g) = pack4(dO[2 * j][nb]);
            *reinterpret_cast<bf16x4*>(myO + pt * PQ + 32 + 8 * g) = pack4(dO[2 * j + 1][nb]);
          }
#pragma unroll
          for (int tt = 0; tt < 2; ++tt)
            wqa[j][tt] = mfma32(xaug, tr_frag_small(myO, PQ, 16 * tt, lane), wqa[j][tt]);
        }
      }
    }
    PCA_STAMP(4);
    // dO now holds dQp^T
    if (FUSE_WQ) {
      // reduced per head above
    } else {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
        if (live[nb]) {
#pragma unroll
          for (int t = 0; t < DT; ++t)
            *reinterpret_cast<bf16x4*>(a.dQp + row[nb] * D + 16 * t + 4 * g) = pack4(dO[t][nb]);
        }
    }

    if (WANT_DX) {
      f32x4 dx[DT][NB];
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) dx[t][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        bf16x8 qb2[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) qb2[nb] = pack8(dO[2 * s][nb], dO[2 * s + 1][nb]);
#pragma unroll
        for (int t = 0; t < DT; ++t) {
          const bf16x8 wa =
              *reinterpret_cast<const bf16x8*>(sWqT + swz(16 * t + r, 4 * s + g, ROWB));
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) dx[t][nb] = mfma32(wa, qb2[nb], dx[t][nb]);
        }
      }
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
        if (live[nb]) {
#pragma unroll
          for (int t = 0; t < DT; ++t) {
            const int64_t xo = row[nb] * D + 16 * t + 4 * g;
            if (ABF) *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(a.dX) + xo) =
                         pack4(dx[t][nb]);
            else *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.dX) + xo) =
                     float4{dx[t][nb][0], dx[t][nb][1], dx[t][nb][2], dx[t][nb][3]};
          }
        }
    }
  }
  PCA_STAMP(5);
  if (FUSE_WQ) {
    // wqa rows (4g + e) = augmented-point rows, columns = features 32j + 16tt + r: combine
    // the hi (g = 0) and lo (g = 1) rows, sum the waves through LDS, one atomic per element
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);             // [4 waves][D][4]
#pragma unroll
    for (int j = 0; j < KS; ++j)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float lo = __shfl(wqa[j][tt][e], 16 + r, 64);     // row 4 + e of column r
          v[e] = wqa[j][tt][e] + (e < 3 ? lo : 0.f);
        }
        if (g == 0) {
          float* dst = red + ((wave8 * D) + 32 * j + 16 * tt + r) * 4;
          dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];
        }
      }
    __syncthreads();
    for (int i = tid; i < D * 4; i += NT) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) v += red[w * D * 4 + i];
      const int f = i >> 2, c = i & 3;
      if (a.wq_slab != nullptr) {
        float* slab = a.wq_slab + (int64_t)blockIdx.x * (D * a.dq + D);
        if (c < a.dq) slab[f * a.dq + c] = v;
        else if (c == 3) slab[D * a.dq + f] = v;
      } else if (c < a.dq) atomicAdd(&a.dWqS[f * a.dq + c], v);
      else if (c == 3) atomicAdd(&a.dbqS[f], v);
    }
    __syncthreads();
  }
  PCA_STAMP(6);
  if (FUSE_KV && cur_b >= 0) {
    // reduce the four waves' [MI][D] partials in LDS (the weight images are dead), then one
    // atomic per element per workgroup
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);            // [NW][2][MI*D] = 64 / 128 KiB
#pragma unroll
    for (int j = 0; j < KS; ++j)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int o = (4 * g + e) * D + 32 * j + 16 * tt + r;
          red[(wave8 * 2 + 0) * MI * D + o] = dkp[j][tt][e];
          red[(wave8 * 2 + 1) * MI * D + o] = dvp[j][tt][e];
        }
    __syncthreads();
    const int nparts = a.tiles_per_set / a.tpw;
    const int part = (t_first - cur_b * a.tiles_per_set) / a.tpw;
    for (int i = tid; i < 2 * MI * D; i += NT) {
      const int which = i / (MI * D), o = i - which * MI * D;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) v += red[(w * 2 + which) * MI * D + o];
      (which ? a.dVpG : a.dKpG)[((int64_t)cur_b * nparts + part) * MI * D + o] = v;
    }
  }
  PCA_STAMP(7);
#ifdef PCA_DEBUG_CLOCKS
  if (a.dbg && (int)blockIdx.x == a.dbg_wg && threadIdx.x == 0)
    for (int i = 0; i < 10; ++i) a.dbg[i] = ((long long)i << 48) | (stamp_[i] & 0xffffffffffffLL);
  if (a.dbg && threadIdx.x == 0 && blockIdx.x < 1024) {
    a.dbg[64 + 2 * blockIdx.x] = stamp_[0];
    a.dbg[65 + 2 * blockIdx.x] = stamp_[7];
  }
#endif
}

// ---------------------------------------------------------------------------------
// dW[DG x DA] += G[rows, DG]^T . A[rows, DA]  (+ db[DG] += column sums of G)
// ---------------------------------------------------------------------------------
// LDS image of a 32-row tile with 256-byte rows; 16-byte chunk ch of row `row` sits at
// (cdna_hip_programming.md T10, image (b)): serves the transposed reads conflict-free
__device__ __forceinline__ int tr_off(int row, int ch) {
  return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}


// A/B fragment with k = the 32 points of the tile (k-slot (g, j): j<4 -> point 4g+j,
// else 16+4g+j-4) for the 16 features of tile t
__device__ __forceinline__ bf16x8 tr_frag(const char* img, int t, int lane) {
  const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
  const int a0 = tr_off(4 * g + q, 2 * t + (p >> 1)) + 8 * (p & 1);
  const int a1 = tr_off(16 + 4 * g + q, 2 * t + (p >> 1)) + 8 * (p & 1);
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + a0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + a1));
  const bf16x4 l4 = __builtin_bit_cast(bf16x4, lo), h4 = __builtin_bit_cast(bf16x4, hi);
  bf16x8 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) { r[e] = l4[e]; r[4 + e] = h4[e]; }
  return r;
}

__device__ __forceinline__ bf16x8 load8(const __bf16* p) {
  return *reinterpret_cast<const bf16x8*>(p);
}
__device__ __forceinline__ bf16x8 load8(const float* p) {
  const float4 lo = reinterpret_cast<const float4*>(p)[0], hi = reinterpret_cast<const float4*>(p)[1];
  bf16x8 v;
  v[0] = (__bf16)lo.x; v[1] = (__bf16)lo.y; v[2] = (__bf16)lo.z; v[3] = (__bf16)lo.w;
  v[4] = (__bf16)hi.x; v[5] = (__bf16)hi.y; v[6] = (__bf16)hi.z; v[7] = (__bf16)hi.w;
  return v;
}

// Software-pipelined: the next 32-row tile is fetched into registers while the current one is
// consumed from LDS; two LDS buffers -> one barrier per tile.  blockIdx.y selects the job.
// NG row groups of four waves each (NG = 2 for the long B*N-row jobs: twice the loads in flight
// per CU; the kernel runs on at most half the CUs because every workgroup costs 16384 atomics):
// group q takes the 32-row tiles q, q + NG, ...; the groups' [128][128] blocks are summed in LDS.
template <typename GT, typename AT, int NG>
__global__ __launch_bounds__(256 * NG) void k_wgrad128(const WgradJobs jobs, int rows_per_wg,
                                                       const SlabSumJobs riders) {
  constexpr int D = 128, NT = 256 * NG;
  // 64 KiB: 2 x 2 staging tiles per group during the loop, the fp32 [128][128] result afterwards
  __shared__ __attribute__((aligned(16))) char lds[4 * 32 * 256 * 2];
  if ((int)blockIdx.y >= jobs.n) {          // rider rows: partial sums that are due now
    slab_sum_body(riders.j[blockIdx.y - jobs.n], blockIdx.x, threadIdx.x,
                  reinterpret_cast<float4*>(lds));
    return;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, grp = tid >> 8;
  const int gtid = tid & 255;
  char (*sG)[32 * 256] = reinterpret_cast<char (*)[32 * 256]>(lds + grp * (4 * 32 * 256));
  char (*sA)[32 * 256] = reinterpret_cast<char (*)[32 * 256]>(lds + grp * (4 * 32 * 256) + 2 * 32 * 256);
  const WgradJob job = jobs.j[blockIdx.y];
  const GT* __restrict__ G = reinterpret_cast<const GT*>(job.G);
  const AT* __restrict__ A = reinterpret_cast<const AT*>(job.A);
  const int64_t M = job.M;
  const int r = lane & 15, g = lane >> 4;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_wg;
  if (r0 >= M) return;
  const int64_t r1 = (r0 + rows_per_wg < M) ? r0 + rows_per_wg : M;
  // wave w owns the 64 x 64 output block (G features 64*(w>>1).., A features 64*(w&1)..):
  // 4 + 4 transposed fragments feed 16 MFMAs per 32-row tile
  const int gt0 = 4 * (wave >> 1), at0 = 4 * (wave & 1);
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // column sums of G, columns 8*(tid&15)..

  bf16x8 vg[2], va[2];
  uint32_t mk[2][2] = {{~0u, ~0u}, {~0u, ~0u}};       // ReLU mask words of the fetched chunks (job.mask)
  auto fetch = [&](int64_t base) {          // rows at and beyond r1 read as zeros
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int c = gtid + e * 256;
      const int row = c >> 4, ch = c & 15;
      if (base + row < r1) {          // (guarded on purpose: the unconditional form measured +18 %)
        vg[e] = load8(G + (base + row) * D + ch * 8);
        va[e] = load8(A + (base + row) * D + ch * 8);
        if (job.mask != nullptr) {
          // features 8 ch .. 8 ch + 7 of row R: two nibbles (bit 4 t + e of lane (r, g) <-> feature
          // 16 t + 4 g + e, t = ch / 2) of the words of lanes g0 = 2 (ch & 1) and g0 + 1.  Only
          // requested here; applied when the tile goes to LDS (the loads stay in flight meanwhile)
          const int64_t R = base + row;
          const uint32_t* mw = job.mask + (R >> 4) * 64 + (R & 15) + 32 * (ch & 1);
          mk[e][0] = mw[0];
          mk[e][1] = mw[16];
        }
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) { vg[e][k] = (__bf16)0.f; va[e][k] = (__bf16)0.f; }
      }
    }
  };
  fetch(r0 + 32 * grp);
  int buf = 0;
  // uniform trip count for all groups (a group whose tile lies beyond r1 multiplies zeros)
  for (int64_t base0 = r0; base0 < r1; base0 += 32 * NG, buf ^= 1) {
    const int64_t base = base0 + 32 * grp;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int c = gtid + e * 256;
      const int row = c >> 4, ch = c & 15;
      if (job.mask != nullptr) {
        const uint32_t n0 = mk[e][0] >> (4 * (ch >> 1)), n1 = mk[e][1] >> (4 * (ch >> 1));
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (!((n0 >> k) & 1u)) vg[e][k] = (__bf16)0.f;
          if (!((n1 >> k) & 1u)) vg[e][4 + k] = (__bf16)0.f;
        }
      }
      *reinterpret_cast<bf16x8*>(sG[buf] + tr_off(row, ch)) = vg[e];
      *reinterpret_cast<bf16x8*>(sA[buf] + tr_off(row, ch)) = va[e];
      if (job.db != nullptr) {
#pragma unroll
        for (int k = 0; k < 8; ++k) bs[k] += (float)vg[e][k];
      }
    }
    __syncthreads();
    if (base0 + 32 * NG < r1) fetch(base + 32 * NG);
    bf16x8 ga[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) ga[i] = tr_frag(sG[buf], gt0 + i, lane);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const bf16x8 ab = tr_frag(sA[buf], at0 + t, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][t] = mfma32(ga[i], ab, acc[i][t]);
    }
  }
  // The atomics cost per cache-line transaction, not per lane: stage the [128][128] block in
  // LDS and add it with fully coalesced instructions (64 consecutive floats per wave) instead
  // of 16-float row fragments straight from the accumulator layout.
  __syncthreads();
  float* res = reinterpret_cast<float*>(lds);
#pragma unroll
  for (int q = NG - 1; q >= 0; --q) {       // last group stores, the others add on top
    if (grp == q) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int grow = 16 * (gt0 + i) + 4 * g + e;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            float* dst = &res[grow * D + 16 * (at0 + t) + r];
            *dst = (q == NG - 1) ? acc[i][t][e] : *dst + acc[i][t][e];
          }
        }
    }
    __syncthreads();
  }
  // slab mode: this workgroup's [rows][128] block (+ its 128 bias sums) as plain stores
  const int n1 = (job.g_hi - job.g_lo) * D;
  float* slab = job.slab == nullptr
                    ? nullptr
                    : job.slab + (int64_t)blockIdx.x * (n1 + (job.db != nullptr ? D : 0));
  if (slab != nullptr)
    for (int i = tid; i < n1; i += NT) slab[i] = res[job.g_lo * D + i];
  else
    for (int i = job.g_lo * D + tid; i < job.g_hi * D; i += NT) atomicAdd(&job.dW[i], res[i]);
  if (job.db != nullptr) {
    // threads with equal (tid & 15) hold partial sums of the same 8 columns
    __syncthreads();
    float* red = reinterpret_cast<float*>(lds);         // [16 NG groups][128 columns]
#pragma unroll
    for (int k = 0; k < 8; ++k) red[(tid >> 4) * D + (tid & 15) * 8 + k] = bs[k];
    __syncthreads();
    if (tid < D) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < 16 * NG; ++q) t += red[q * D + tid];
      if (slab != nullptr) slab[n1 + tid] = t;
      else atomicAdd(&job.db[tid], t);
    }
  }
}

// dH[q][c] (+)= sum_f dKp[q][f] Wk[f][c] + dVp[q][f] Wv[f][c]; thread = (column c, query half)
__global__ __launch_bounds__(256) void k_kv_dh(const float* __restrict__ dKp,
                                               const float* __restrict__ dVp,
                                               const float* __restrict__ Wk,
                                               const float* __restrict__ Wv,
                                               float* __restrict__ dH, int m, int d,
                                               int accumulate) {
  extern __shared__ float sm[];
  float* sK = sm;
  float* sV = sm + m * d;
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < m * d; i += 256) {
    sK[i] = dKp[(int64_t)b * m * d + i];
    sV[i] = dVp[(int64_t)b * m * d + i];
  }
  __syncthreads();
  const int c = tid % d, q0 = (tid / d) * 8;
  if (q0 >= m) return;
  float acc[8];
#pragma unroll
  for (int q = 0; q < 8; ++q)
    acc[q] = (accumulate && q0 + q < m) ? dH[(int64_t)b * m * d + (q0 + q) * d + c] : 0.f;
  col_gemm<8>(sK + q0 * d, d, Wk, d, d, c, acc);
  col_gemm<8>(sV + q0 * d, d, Wv, d, d, c, acc);
#pragma unroll
  for (int q = 0; q < 8; ++q)
    if (q0 + q < m) dH[(int64_t)b * m * d + (q0 + q) * d + c] = acc[q];
}

// layer 1: dW[D x dq] += dQp^T . X with dq <= 4 (fp32 X), db += colsum(dQp).
// 256 threads = 128 features x 2 row phases; 128 rows per workgroup, loads unrolled.
template <typename GT>
__global__ __launch_bounds__(256) void k_wgrad_small(const GT* __restrict__ G,
                                                     const float* __restrict__ X, int64_t M,
                                                     int dq, int rows_per_wg,
                                                     int64_t x_head_stride,   // A = X + (f/32)*stride
                                                     float* __restrict__ dW,
                                                     float* __restrict__ db) {
  wgrad_small_body<GT>(G, X, M, dq, rows_per_wg, x_head_stride, dW, db, blockIdx.x);
}

// per (set, head): dKp[b][key][32j + c] = sum_n dS[n][j*MI + key] Qp[n][32j + c]
//                  dVp[b][key][32j + c] = sum_n  P[n][j*MI + key] dO[n][32j + c]
template <int MI>
__global__ __launch_bounds__(256) void k_kv_grad(const __bf16* __restrict__ dS,
                                                 const __bf16* __restrict__ P,
                                                 const __bf16* __restrict__ Qp,
                                                 const __bf16* __restrict__ dO, int N, int D,
                                                 int H, float* __restrict__ dKp,
                                                 float* __restrict__ dVp) {
  constexpr int CH = 64;                   // points per staged chunk
  __shared__ float sS[CH][MI + 1], sP[CH][MI + 1], sQ[CH][33], sO[CH][33];
  const int b = blockIdx.x / H, j = blockIdx.x % H;
  const int HM = H * MI;
  const int tid = threadIdx.x;
  const int c = tid & 31, k0 = tid >> 5;         // outputs (key = k0 + 8*i, c)
  constexpr int KPT = MI / 8;
  float ak[KPT], av[KPT];
#pragma unroll
  for (int i = 0; i < KPT; ++i) ak[i] = av[i] = 0.f;
  for (int n0 = 0; n0 < N; n0 += CH) {
    for (int i = tid; i < CH * MI; i += 256) {
      const int pnt = i / MI, key = i % MI;
      const int64_t row = (int64_t)b * N + n0 + pnt;
      const bool ok = n0 + pnt < N;
      sS[pnt][key] = ok ? (float)dS[row * HM + j * MI + key] : 0.f;
      sP[pnt][key] = ok ? (float)P[row * HM + j * MI + key] : 0.f;
    }
    for (int i = tid; i < CH * 32; i += 256) {
      const int pnt = i >> 5, cc = i & 31;
      const int64_t row = (int64_t)b * N + n0 + pnt;
      const bool ok = n0 + pnt < N;
      sQ[pnt][cc] = ok ? (float)Qp[row * D + 32 * j + cc] : 0.f;
      sO[pnt][cc] = ok ? (float)dO[row * D + 32 * j + cc] : 0.f;
    }
    __syncthreads();
#pragma unroll 4
    for (int pnt = 0; pnt < CH; ++pnt) {
      const float q = sQ[pnt][c], o = sO[pnt][c];
#pragma unroll
      for (int i = 0; i < KPT; ++i) {
        ak[i] += sS[pnt][k0 + 8 * i] * q;
        av[i] += sP[pnt][k0 + 8 * i] * o;
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < KPT; ++i) {
    const int64_t o = ((int64_t)b * MI + k0 + 8 * i) * D + 32 * j + c;
    dKp[o] = ak[i];
    dVp[o] = av[i];
  }
}

__global__ void k_sum_parts(const float* __restrict__ kp, const float* __restrict__ vp,
                            float* __restrict__ dk, float* __restrict__ dv, int B, int nparts,
                            int n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)B * n) return;
  const int64_t b = i / n, o = i - b * n;
  float a = 0.f, c = 0.f;
  for (int p = 0; p < nparts; ++p) {
    a += kp[(b * nparts + p) * n + o];
    c += vp[(b * nparts + p) * n + o];
  }
  dk[i] = a;
  dv[i] = c;
}

#ifdef PCA_DEBUG_CLOCKS
// diagnostic build: the last launch's phase stamps of workgroup 0 are printed at exit
long long* debug_clock_buffer(int which) {
  static long long* buf[2] = {nullptr, nullptr};
  if (buf[which] == nullptr) {
    (void)hipHostMalloc(reinterpret_cast<void**>(&buf[which]), (64 + 2048) * sizeof(long long), 0);
    for (int i = 0; i < 64 + 2048; ++i) buf[which][i] = 0;
    static int reg[2] = {0, 1};
    struct P { static void dump(int w) {
      long long* b = buf[w]; long long t0 = b[0] & 0xffffffffffffLL;
      fprintf(stderr, "k_mab1_bwd[%d] stamps (phase:us):", w);
      for (int i = 0; i < 64 && (i == 0 || b[i]); ++i)
        fprintf(stderr, " %lld:%.2f", b[i] >> 48, ((b[i] & 0xffffffffffffLL) - t0) / 100.0);
      fprintf(stderr, "\n");
      long long s0 = 0; int n = 0;
      for (int i = 0; i < 1024; ++i) if (b[64 + 2 * i]) { if (!n || b[64 + 2 * i] < s0) s0 = b[64 + 2 * i]; ++n; }
      fprintf(stderr, "k_mab1_bwd[%d] per-workgroup start/end us (%d wgs):", w, n);
      for (int i = 0; i < n; i += (n > 64 ? n / 32 : 1))
        fprintf(stderr, " %d:%.1f/%.1f", i, (b[64 + 2 * i] - s0) / 100.0, (b[65 + 2 * i] - s0) / 100.0);
      fprintf(stderr, "\n"); } };
    if (which == 0) atexit([] { P::dump(0); }); else atexit([] { P::dump(1); });
    (void)reg;
  }
  return buf[which];
}
#endif
template <int D, int MI, bool DX, bool FUSE, bool FWQ, bool ABF>
int launch_bwd(const Mab1BwdArgs& a, hipStream_t st, double flops, double bytes) {
  constexpr int NW = BwdWaves<DX, FUSE>::value;
  size_t lds = (size_t)D * D * 2 + 2 * (size_t)MI * D * 2 + (size_t)D * MI * 2 +
               (DX ? (size_t)D * D * 2 : 0) + (FUSE ? NW * (2 * 32 * 40 + 2 * 32 * 72) : 0) +
               (FWQ ? (size_t)D * sizeof(float4) : 0);
  if (FUSE && lds < (size_t)2 * NW * MI * D * 4) lds = (size_t)2 * NW * MI * D * 4;   // flush buffer
  static std::once_flag once;
  std::call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_mab1_bwd<D, MI, DX, FUSE, FWQ, ABF>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  const int total = a.B * a.tiles_per_set;
  const int grid = FUSE ? (int)cdiv(total, a.tpw) : (total < 256 ? total : 256);
  ProfScope ps(PCA_K_MAB1_BWD, st, flops, bytes);
  hipLaunchKernelGGL((k_mab1_bwd<D, MI, DX, FUSE, FWQ, ABF>), dim3(grid), dim3(64 * NW), lds, st, a);
  ps.end();
  return check_launch("k_mab1_bwd");
}

}  // namespace
size_t mab1_carve_bwd_ws(const pca_mab_shape& s, Mab1BwdWs* out, void* base) {
  Carver c(base);
  Mab1BwdWs w;
  const size_t M = (size_t)s.B * s.nq, d = s.d;
  w.WoTP = c.take<__bf16>(d * d);
  w.WqTP = c.take<__bf16>(d * d);
  w.dZ = c.take<__bf16>(M * d);
  w.dQp = c.take<__bf16>(M * d);
  w.dOs = c.take<__bf16>(M * d);
  w.dS = c.take<__bf16>(M * s.h * s.nk);
  w.P = c.take<__bf16>(M * s.h * s.nk);
  w.dKp = c.take<float>((size_t)s.B * s.nk * d);
  w.dVp = c.take<float>((size_t)s.B * s.nk * d);
  const size_t tiles = (size_t)cdiv(s.nq, M1_TP);
  w.dKpPart = c.take<float>((size_t)s.B * tiles * s.nk * d);
  w.dVpPart = c.take<float>((size_t)s.B * tiles * s.nk * d);
  if (out) *out = w;
  return c.off;
}

size_t mab1_bf16_bwd_ws_bytes(const pca_mab_shape& s) {
  if (s.d == 256) return mab1_d256_bwd_ws_bytes(s);
  return mab1_carve_bwd_ws(s, nullptr, nullptr);
}

int wgrad128_launch(const WgradJobs& jobs_in, bool g_bf16, bool a_bf16, int rows_per_wg,
                    hipStream_t st, WgradSlabs* sl) {
  WgradJobs jobs = jobs_in;
  int64_t maxM = 0;
  for (int i = 0; i < jobs.n; ++i) maxM = jobs.j[i].M > maxM ? jobs.j[i].M : maxM;
  SlabSumJobs riders{};
  if (sl != nullptr && sl->riders != nullptr) riders = *sl->riders;
  if (maxM == 0 || jobs.n == 0) return slab_sum_jobs(riders, st);
  if (sl != nullptr && sl->ws != nullptr) {
    // one slab per workgroup and job; more rows per workgroup until they fit
    for (;;) {
      size_t need = 0;
      for (int i = 0; i < jobs.n; ++i) {
        const WgradJob& j = jobs.j[i];
        need += (size_t)cdiv(j.M, rows_per_wg) * ((j.g_hi - j.g_lo) * 128 + (j.db ? 128 : 0)) * 4;
      }
      if (need <= sl->cap) break;
      rows_per_wg *= 2;
    }
    float* at = sl->ws;
    for (int i = 0; i < jobs.n; ++i) {
      WgradJob& j = jobs.j[i];
      if (j.M <= 0) continue;
      const int nwg = (int)cdiv(j.M, rows_per_wg), n1 = (j.g_hi - j.g_lo) * 128;
      const int stride = n1 + (j.db ? 128 : 0);
      j.slab = at;
      PCA_REQUIRE(sl->sums_out->n + 2 <= 40, "wgrad128: slab-sum table full");
      sl->sums_out->j[sl->sums_out->n++] = SlabSumJob{at, j.dW + (int64_t)j.g_lo * 128, nwg, n1, 1, stride};
      if (j.db) sl->sums_out->j[sl->sums_out->n++] = SlabSumJob{at + n1, j.db, nwg, 128, 1, stride};
      at += (size_t)nwg * stride;
    }
    sl->used = (size_t)(at - sl->ws) * sizeof(float);
  }
  unsigned gx = (unsigned)cdiv(maxM, rows_per_wg);
  for (int i = 0; i < riders.n; ++i) {
    PCA_REQUIRE(slab_sum_job_ok(riders.j[i]), "wgrad128: rider alignment");
    const unsigned need = (unsigned)cdiv(riders.j[i].n, 256);
    gx = need > gx ? need : gx;
  }
  const dim3 grid(gx, (unsigned)(jobs.n + riders.n));
  // two row groups per workgroup when every workgroup has at least four tiles to share
  const bool two = rows_per_wg >= 128;
  if (g_bf16 && a_bf16) {
    if (two) hipLaunchKernelGGL((k_wgrad128<__bf16, __bf16, 2>), grid, dim3(512), 0, st, jobs, rows_per_wg, riders);
    else hipLaunchKernelGGL((k_wgrad128<__bf16, __bf16, 1>), grid, dim3(256), 0, st, jobs, rows_per_wg, riders);
  } else if (g_bf16) {
    if (two) hipLaunchKernelGGL((k_wgrad128<__bf16, float, 2>), grid, dim3(512), 0, st, jobs, rows_per_wg, riders);
    else hipLaunchKernelGGL((k_wgrad128<__bf16, float, 1>), grid, dim3(256), 0, st, jobs, rows_per_wg, riders);
  } else if (!a_bf16) {
    if (two) hipLaunchKernelGGL((k_wgrad128<float, float, 2>), grid, dim3(512), 0, st, jobs, rows_per_wg, riders);
    else hipLaunchKernelGGL((k_wgrad128<float, float, 1>), grid, dim3(256), 0, st, jobs, rows_per_wg, riders);
  } else {
    set_error("wgrad128: fp32 G with bf16 A is not instantiated");
    return PCA_EUNSUPPORTED;
  }
  return check_launch("k_wgrad128");
}

int wgrad128_defer(BwdDefer* defer, const WgradJobs& jobs, bool bf16, int rows_per_wg,
                   hipStream_t st) {
  if (defer == nullptr) return wgrad128_launch(jobs, bf16, bf16, rows_per_wg, st);
  WgradJobs& L = bf16 ? defer->wg_bf16 : defer->wg_f32;
  if (L.n + jobs.n > 16) {                 // table full: run what has been collected
    PCA_TRY(wgrad128_launch(L, bf16, bf16, bf16 ? 512 : 64, st));
    L.n = 0;
  }
  for (int i = 0; i < jobs.n; ++i) L.j[L.n++] = jobs.j[i];
  return PCA_OK;
}

bool wgrad_slabs_on() {          // (read per call: a test switches it between two engines)
  const char* e = getenv("PCA_WGRAD_SLABS");
  return !(e != nullptr && e[0] == '0');
}
int bwd_defer_flush(BwdDefer& D, hipStream_t st) {
  hipStream_t ts = terminal_stream(st);
  // The sums the post stages read (D.sums: dG of the few-queries blocks) ride in the first
  // weight-gradient launch as extra workgroup rows.  Slab mode (the default when the caller lent
  // room; PCA_WGRAD_SLABS=0 switches back to fp32 atomics): the weight gradients themselves use no
  // atomics either - per-workgroup partials, summed in a fixed order by rider rows of k_terminal1
  // (`late`: only the optimizer reads them).  With EVERY reduction of the step in this form
  // configs[1] measured 0.324 ms/step against 0.335 with the atomics (three same-box pairs), and the
  // step is bit-reproducible.  (With only k_wgrad128 converted it was 0.343 ... 0.361 against 0.348,
  // depending on where the partials happened to lie.)
  const bool slab_mode = wgrad_slabs_on() && D.slab_ws != nullptr && D.slab_cap > 0;
  SlabSumJobs late{};
  size_t used = 0;          // the two lists' slabs lie back to back
  if (D.wg_bf16.n > 0) {
    double rows = 0;
    for (int i = 0; i < D.wg_bf16.n; ++i) rows += (double)D.wg_bf16.j[i].M;
    ProfScope ps(PCA_K_WGRAD, ts, 2.0 * rows * 128 * 128, 4.0 * rows * 128);
    // every workgroup costs a 64 KiB slab (16384 atomics without the slabs): aim at ~200
    // workgroups over all jobs (512 rows for one B*N-row job, 1024 for three, ...)
    // (measured at 3 x 65536 rows: 512 -> 41 us, 768 -> 41, 1024 -> 31, 1536 -> 31, 2048 -> 39)
    int rpw = 512 * (int)((rows + 98303.0) / 98304.0);
    rpw = rpw < 512 ? 512 : (rpw > 1024 ? 1024 : rpw);
    WgradSlabs sl{slab_mode ? D.slab_ws : nullptr, D.slab_cap * 3 / 4, &late, &D.sums, 0};
    PCA_TRY(wgrad128_launch(D.wg_bf16, true, true, rpw, ts, &sl));
    used = (sl.used + 255) & ~(size_t)255;
    ps.end();
    D.wg_bf16.n = 0;
    D.sums.n = 0;
  }
  if (D.wg_f32.n > 0) {
    // 64: 18.7 us, 128: 14.2, 256: 15.3
    WgradSlabs sl{slab_mode ? D.slab_ws + used / sizeof(float) : nullptr, D.slab_cap - used, &late,
                  &D.sums, 0};
    PCA_TRY(wgrad128_launch(D.wg_f32, false, false, 128, ts, &sl));
    used += (sl.used + 255) & ~(size_t)255;
    D.wg_f32.n = 0;
    D.sums.n = 0;
  }
  if (D.wg256_n > 0) PCA_TRY(wgrad256_flush_deferred(D, ts));
  PCA_TRY(slab_sum_jobs(D.sums, ts));        // (nobody carried them)
  D.sums.n = 0;
  for (int i = 0; i < D.late.n; ++i) {
    PCA_REQUIRE(late.n < 40, "bwd_defer_flush: slab-sum table full");
    late.j[late.n++] = D.late.j[i];
  }
  D.late.n = 0;
  if (slab_mode && D.has_sw) {     // layer-1 fc_v gradient (rider of k_terminal1): slabs as well
    const int nwg = (int)cdiv(D.sw.M, D.sw.rows_per_wg), stride = 128 * D.sw.dq + 128;
    if ((128 * D.sw.dq) % 4 == 0 && used + (size_t)nwg * stride * 4 <= D.slab_cap)
      D.sw.slab = D.slab_ws + used / sizeof(float);
  }
  PCA_TRY(terminal_launch(D, st, &late));
  D.posts.n = 0;
  D.has_cls = D.has_sw = 0;
  return PCA_OK;
}

int wgrad_small_f32_launch(const float* G, const float* X, int64_t M, int dq,
                           int64_t x_head_stride, float* dW, float* db, hipStream_t st,
                           BwdDefer* defer) {
  if (defer != nullptr && !defer->has_sw) {
    defer->sw = SmallWgradArgs{G, X, M, dq, 64, x_head_stride, dW, db, nullptr};
    defer->has_sw = 1;
    return PCA_OK;
  }
  hipLaunchKernelGGL((k_wgrad_small<float>), dim3((unsigned)cdiv(M, 64)), dim3(256), 0, st, G, X, M,
                     dq, 64, x_head_stride, dW, db);
  return check_launch("k_wgrad_small<float>");
}

int kv_dh_launch(const float* dKp, const float* dVp, const float* Wk, const float* Wv, float* dH,
                 int B, int m, int d, int accumulate, hipStream_t st) {
  PCA_REQUIRE(d == 128 && m <= 16, "kv_dh: d=%d m=%d", d, m);
  hipLaunchKernelGGL(k_kv_dh, dim3(B), dim3(256), 2 * (size_t)m * d * sizeof(float), st, dKp, dVp,
                     Wk, Wv, dH, m, d, accumulate);
  return check_launch("k_kv_dh");
}

// dQ -> dX [B, nq, dq] (written; may be null), dK -> dH [B, nk, d] (written or accumulated)
int mab1_bf16_bwd(const pca_mab_shape& s, const void* X, const float* H,
                  const pca_mab_params& p, const void* saved, const void* dY, void* dX,
                  float* dH, int dk_accumulate, const pca_mab_grads& gr, void* ws,
                  hipStream_t st) {
  return mab1_bf16_bwd_ex(s, X, H, p, saved, dY, dX, dH, dk_accumulate, gr, ws, 0, st);
}
int mab1_bf16_bwd_ex(const pca_mab_shape& s, const void* X, const float* H,
                     const pca_mab_params& p, const void* saved, const void* dY, void* dX,
                     float* dH, int dk_accumulate, const pca_mab_grads& gr, void* ws, int flags,
                     hipStream_t st, const IsabImg* img, float* zero_ptr, int zero_n,
                     int* nparts_out, BwdDefer* defer) {
  if (s.d == 256)       // three launches + the 256-wide weight-gradient reduction (d256_host.hip)
    return mab1_d256_bwd(s, X, H, p, saved, dY, dX, dH, dk_accumulate, gr, ws, st, defer);
  Mab1Saved v;
  mab1_carve_saved(s, &v, const_cast<void*>(saved));
  Mab1BwdWs w;
  mab1_carve_bwd_ws(s, &w, ws);
  const int d = s.d, MI = s.nk;
  const int64_t M = (int64_t)s.B * s.nq;
  const bool small = s.dq <= 4;
  const bool want_dx = dX != nullptr && !small;
  const bool abf = s.y_dtype == PCA_BF16;
  if (dX != nullptr && small) {
    // the ST model never needs it (the set itself is the input of layer 1)
    set_error("mab1_bf16_bwd: dQ for dq <= 4 is not built");
    return PCA_EUNSUPPORTED;
  }

  if (img != nullptr) {
    w.WoTP = img->WoTP;
    w.WqTP = img->WqTP;
  } else {
    PCA_TRY(prep_weight(p.wo, w.WoTP, d, d, 2, st));
    if (want_dx) PCA_TRY(prep_weight(p.wq, w.WqTP, d, d, 2, st));
  }

  // dZ = dY . [Z > 0] is read by the fc_o weight-gradient job only: with bf16 gradients and whole
  // 128-point mask blocks per set that job takes dY and the mask words (WgradJob::mask) and dZ is
  // never written (PCA_D128_DZ_MASK=0: the materialised form)
  static const bool dzm_on = [] { const char* e = getenv("PCA_D128_DZ_MASK"); return !(e && e[0] == '0'); }();
  const bool dz_masked = dzm_on && abf && MI == 16 && s.nq % 128 == 0;
  Mab1BwdArgs a{};
  a.dY = dY; a.QpS = v.QpS; a.mask = v.mask; a.KpP = v.KpP; a.VpP = v.VpP; a.Kt = v.Kt;
  a.WoTP = w.WoTP; a.WqTP = w.WqTP; a.dZ = dz_masked ? nullptr : w.dZ; a.dQp = w.dQp; a.dOs = w.dOs; a.dS = w.dS;
  a.P = w.P; a.dX = want_dx ? dX : nullptr;
  a.B = s.B; a.N = s.nq; a.tiles_per_set = (int)cdiv(s.nq, TP);
  a.scale = 1.0f / sqrtf((float)d);
  a.scale_log2e = 1.4426950408889634f * a.scale;
#ifdef PCA_DEBUG_CLOCKS
  a.dbg_wg = getenv("PCA_DBG_WG") ? atoi(getenv("PCA_DBG_WG")) : 0;
#else
  a.dbg_wg = 0;
#endif
  // reference-formulation FLOPs of what THIS launch computes (round 3: not the whole block's
  // backward - dWo and, beyond layer 1, dWq run in k_wgrad128 and are charged there): fc_o adjoint
  // dO = dY + dZ Wo (2 M d^2), attention adjoint (recomputed scores, dP, dQp, dKp, dVp: 8 M m d),
  // dX = dQp Wq (2 M dq d) when asked for, layer 1's in-kernel dWq (2 M dq d)
  const double flops = (double)M * (2.0 * d * d + 8.0 * MI * d +
                                    (want_dx ? 2.0 * s.dq * d : 0.0) +
                                    ((small && s.dq <= 3 && MI == 16) ? 2.0 * s.dq * d : 0.0));
  // algorithmic HBM bytes (each operand once): dY in, X in, dX out
  const double eb = abf ? 2.0 : 4.0;
  const double bytes = (double)M * (eb * d + (small ? 4.0 * s.dq : eb * s.dq) +
                                    (want_dx ? eb * d : 0.0));
  int rc;
  const bool fuse = MI == 16;
  if (fuse) {
    // consecutive tiles per workgroup: the largest divisor of tiles_per_set that still
    // leaves >= 256 workgroups
    // (without the dX stage the kernel needs 72 KiB of LDS: two workgroups per CU)
    const int min_wg = want_dx ? 256 : 512;
    int tpw = 1;
    for (int c = 1; c <= a.tiles_per_set; ++c)
      if (a.tiles_per_set % c == 0 && (int64_t)a.B * a.tiles_per_set / c >= min_wg) tpw = c;
    a.tpw = tpw;
    a.dKpG = w.dKpPart;
    a.dVpG = w.dVpPart;
    a.zero_ptr = zero_ptr;
    a.zero_n = zero_n;
#ifdef PCA_DEBUG_CLOCKS
    a.dbg = debug_clock_buffer(want_dx ? 1 : 0);
#endif
    const bool fwq = small && s.dq <= 3;
    a.Xs = reinterpret_cast<const float*>(X); a.dWqS = gr.wq; a.dbqS = gr.bq; a.dq = s.dq;
    a.WqF = p.wq; a.bqF = p.bq;
    // slab mode: the (unused in this mode) dS block of the workspace holds the fc_q partials
    if (fwq && wgrad_slabs_on() && (128 * s.dq) % 4 == 0 &&
        (size_t)a.B * a.tiles_per_set / tpw * (128 * s.dq + 128) * 4 <= (size_t)M * s.h * s.nk * 2)
      a.wq_slab = reinterpret_cast<float*>(w.dS);
    if (abf)
      rc = want_dx ? launch_bwd<128, 16, true, true, false, true>(a, st, flops, bytes)
           : fwq   ? launch_bwd<128, 16, false, true, true, true>(a, st, flops, bytes)
                   : launch_bwd<128, 16, false, true, false, true>(a, st, flops, bytes);
    else
      rc = want_dx ? launch_bwd<128, 16, true, true, false, false>(a, st, flops, bytes)
           : fwq   ? launch_bwd<128, 16, false, true, true, false>(a, st, flops, bytes)
                   : launch_bwd<128, 16, false, true, false, false>(a, st, flops, bytes);
  } else {
    PCA_REQUIRE(!abf, "mab1_bf16_bwd: bf16 activations need m = 16");
    rc = want_dx ? launch_bwd<128, 32, true, false, false, false>(a, st, flops, bytes)
                 : launch_bwd<128, 32, false, false, false, false>(a, st, flops, bytes);
  }
  PCA_TRY(rc);
  if (a.wq_slab != nullptr) {        // fc_q of layer 1: the workgroups' partials, summed in a fixed order
    const int nwg = a.B * a.tiles_per_set / a.tpw, n1 = 128 * s.dq, stride = n1 + 128;
    SlabSumJobs two{};
    two.j[two.n++] = SlabSumJob{a.wq_slab, gr.wq, nwg, n1, 1, stride};
    two.j[two.n++] = SlabSumJob{a.wq_slab + n1, gr.bq, nwg, 128, 1, stride};
    if (defer != nullptr) {
      PCA_REQUIRE(defer->late.n + 2 <= 40, "mab1_bf16_bwd: slab-sum table full");
      defer->late.j[defer->late.n++] = two.j[0];
      defer->late.j[defer->late.n++] = two.j[1];
    } else {
      PCA_TRY(slab_sum_jobs(two, st));
    }
  }
  if (nparts_out != nullptr) *nparts_out = fuse ? a.tiles_per_set / a.tpw : 0;
  // ---- reductions over points (one launch for dWo / dWq when both operands are bf16) ----
  // 512 rows per workgroup: the fp32 atomics of the [128 x 128] result cost ~1 lane-op per
  // clock per L2 channel, so fewer, longer workgroups win (256 rows: +50 %, 64 rows: 4x)
  const int rows_per_wg = 512;
  const bool wq_big = !small;
  {
    WgradJobs jobs{};
    if (dz_masked)
      jobs.j[jobs.n++] = WgradJob{dY, v.OS, gr.wo, gr.bo, M, 0, 128, nullptr, v.mask};
    else
      jobs.j[jobs.n++] = WgradJob{w.dZ, v.OS, gr.wo, gr.bo, M, 0, 128};
    if (wq_big && abf) jobs.j[jobs.n++] = WgradJob{w.dQp, X, gr.wq, gr.bq, M, 0, 128};
    hipStream_t ts = terminal_stream(st);
    if (defer != nullptr) {
      PCA_TRY(wgrad128_defer(defer, jobs, true, rows_per_wg, ts));
    } else {
      ProfScope ps(PCA_K_WGRAD, ts, 2.0 * jobs.n * M * d * d, 4.0 * jobs.n * M * d);
      PCA_TRY(wgrad128_launch(jobs, true, true, rows_per_wg, ts));
      ps.end();
    }
  }
  if (small && fuse && s.dq <= 3) {
    // dWq / dbq were reduced inside the chain kernel
  } else if (small) {
    hipLaunchKernelGGL((k_wgrad_small<__bf16>), dim3((unsigned)cdiv(M, 128)), dim3(256), 0, st,
                       w.dQp, reinterpret_cast<const float*>(X), M, s.dq, 128, (int64_t)0, gr.wq,
                       gr.bq);
    PCA_TRY(check_launch("k_wgrad_small"));
  } else if (!abf) {
    WgradJobs jobs{};
    jobs.j[0] = WgradJob{w.dQp, X, gr.wq, gr.bq, M, 0, 128};
    jobs.n = 1;
    hipStream_t ts = terminal_stream(st);
    ProfScope ps(PCA_K_WGRAD, ts, 2.0 * M * d * d, 6.0 * M * d);
    PCA_TRY(wgrad128_launch(jobs, true, false, rows_per_wg, ts));
    ps.end();
  }
  if (!fuse) {
    hipLaunchKernelGGL((k_kv_grad<32>), dim3(s.B * s.h), dim3(256), 0, st, w.dS, w.P, v.QpS,
                       w.dOs, s.nq, d, s.h, w.dKp, w.dVp);
    PCA_TRY(check_launch("k_kv_grad"));
  }

  if (flags & PCA_F_SKIP_KV_TAIL) return PCA_OK;
  if (fuse) {        // stand-alone MAB: sum the per-workgroup partials
    const int nparts = a.tiles_per_set / a.tpw;
    hipLaunchKernelGGL(k_sum_parts, dim3((unsigned)cdiv((int64_t)s.B * MI * d, 256)), dim3(256), 0,
                       st, w.dKpPart, w.dVpPart, w.dKp, w.dVp, s.B, nparts, MI * d);
    PCA_TRY(check_launch("k_sum_parts"));
  }
  // ---- fc_k / fc_v of the m inducing-point outputs: [B*m]-row reductions, one launch ----
  const int64_t Mk = (int64_t)s.B * MI;
  {
    WgradJobs jobs{};
    jobs.j[0] = WgradJob{w.dKp, H, gr.wk, gr.bk, Mk, 0, 128};
    jobs.j[1] = WgradJob{w.dVp, H, gr.wv, gr.bv, Mk, 0, 128};
    jobs.n = 2;
    PCA_TRY(wgrad128_launch(jobs, false, false, 64, st));
  }
  if (dH != nullptr) {
    if (MI <= 16) {
      PCA_TRY(kv_dh_launch(w.dKp, w.dVp, p.wk, p.wv, dH, s.B, MI, d, dk_accumulate ? 1 : 0, st));
    } else {
      PCA_TRY(linear_dx_acc_f32(w.dKp, p.wk, dH, Mk, d, d, dk_accumulate ? 1 : 0, st));
      PCA_TRY(linear_dx_acc_f32(w.dVp, p.wv, dH, Mk, d, d, 1, st));
    }
  }
  return PCA_OK;
}

}  // namespace pca
