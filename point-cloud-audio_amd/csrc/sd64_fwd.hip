// Fused forward of the multihead attention block at the reference's SHIPPED shape: d = 64, 8 heads
// (head dim 8), up to 64 inducing points / seeds (Code/settransformer.py:81-83, Code/pceval.py:41-47;
// set_transformer-master/modules.py:19-33).  Inference only (saved == NULL): the models the paper's
// accuracies and sub-sampling sweeps belong to (Code/pceval.py:86-97,108-192: B = 8, N = 1 ... 10 240).
//
// Round 2 ran these shapes on the materialising GEMM chain (A [h B, nq, nk] through memory, one
// launch per product).  A head dim of 8 is a quarter of the K of the cheapest bf16 MFMA, the whole
// model is 80 k parameters, and a set costs 65 MMAC - so these kernels stay on the fp32 vector ALU
// (exact arithmetic: the "bf16" mode of this shape is NOT reduced precision) and fuse everything a
// block does into two launches (few queries) or one (many queries):
//
//   k_sd_fq      q_shared (ISAB mab0, PMA): blockIdx = (set, point range).  Wave h = head h, lane
//                q = query q.  Per tile of 64 points the wave projects ITS head's 8 features of
//                Kp / Vp for the 64 points (lane = point; the weight rows are wave-uniform:
//                scalar operands) into wave-private LDS, then every lane runs its query over the
//                tile (keys by LDS broadcast): scores, tile maximum, exp2, P V.  No barrier in the
//                loop, no score ever leaves registers.  Partials (m, l, acc[8]) per range.
//   k_sd_fq_epi  per set: merge the ranges, O = Qp + A V, H = O + relu(O Wo^T + bo).
//   k_sd_mq      many queries (ISAB mab1): the workgroup projects the set's m keys (Kp, Vp: 2 x 16
//                KiB of LDS) once, then lane = point: Qp = Wq x + bq, per head 64 scores in
//                registers, softmax, O_h = Qp_h + P V_h, Z = Wo O + bo, Y = O + relu(Z).
#include "mab1_bf16.hpp"

#include <math.h>

#include <mutex>

namespace pca {

namespace {

constexpr int D = 64, H = 8, DH = 8, MQ = 64, TP = 64;

struct SdArgs {
  const float* Qin;        // fq: I [m][dq] (shared) ; mq: X [B][N][dq]
  const float* Kin;        // fq: X [B][N][dk]       ; mq: Hk [B][nk][64]
  const float *wq, *bq, *wk, *bk, *wv, *bv, *wo, *bo;
  float* Y;                // fq_epi: [B][m][64] ; mq: [B][N][64]
  float *Op, *Mp, *Lp;     // fq partials [B][S][H][MQ][8], [B][S][H][MQ], same
  float* Qp;               // fq: [m][64] un-scaled projected query (written by range 0 of set 0)
  const int32_t* lengths;
  int B, N, m, dq, dk, S;
  float sl2e;
};

// ---------------------------------------------------------------------------------------------
// few shared queries over N keys
// ---------------------------------------------------------------------------------------------
template <int DK>       // DK = 64, or 4 for dk <= 4 (layer 1: the raw points)
__global__ __launch_bounds__(512) void k_sd_fq(const SdArgs a) {
  __shared__ float sKV[H][TP][2 * DH];           // wave-private: Kp_h | Vp_h of the tile's points
  const int tid = threadIdx.x, lane = tid & 63;
  const int h = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x, sp = blockIdx.y;
  const int q = lane;
  // ---- this lane's projected query, head h: Qp_h[q][0..7] = I[q] . Wq[8 h + f] + bq ----
  float qp[DH];
#pragma unroll
  for (int f = 0; f < DH; ++f) qp[f] = a.bq[DH * h + f];
  if (q < a.m) {
    for (int c = 0; c < a.dq; ++c) {
      const float x = a.Qin[q * a.dq + c];
#pragma unroll
      for (int f = 0; f < DH; ++f) qp[f] = fmaf(x, a.wq[(DH * h + f) * a.dq + c], qp[f]);
    }
  }
  if (b == 0 && sp == 0 && q < a.m) {
#pragma unroll
    for (int f = 0; f < DH; ++f) a.Qp[q * D + DH * h + f] = qp[f];
  }
#pragma unroll
  for (int f = 0; f < DH; ++f) qp[f] *= a.sl2e;          // scores in the log2 domain
  int len = a.N;
  if (a.lengths != nullptr) len = a.lengths[b] < a.N ? a.lengths[b] : a.N;
  const int per = (int)(((int64_t)(a.N + TP - 1) / TP + a.S - 1) / a.S) * TP;
  const int n_lo = sp * per;
  const int n_hi = (n_lo + per < len) ? n_lo + per : len;
  float mrun = -INFINITY, lrun = 0.f, acc[DH];
#pragma unroll
  for (int f = 0; f < DH; ++f) acc[f] = 0.f;
  const float* Xb = a.Kin + (int64_t)b * a.N * a.dk;
  for (int n0 = n_lo; n0 < n_hi; n0 += TP) {
    // ---- projection: lane = point n0 + lane, this head's 8 features of Kp and Vp ----
    {
      const int n = n0 + lane < a.N ? n0 + lane : a.N - 1;
      float kp[DH], vp[DH];
#pragma unroll
      for (int f = 0; f < DH; ++f) { kp[f] = a.bk[DH * h + f]; vp[f] = a.bv[DH * h + f]; }
      if (DK == 64) {
        const float4* xr = reinterpret_cast<const float4*>(Xb + (int64_t)n * 64);
#pragma unroll 4
        for (int c4 = 0; c4 < 16; ++c4) {
          const float4 x = xr[c4];
#pragma unroll
          for (int f = 0; f < DH; ++f) {
            const float* wkr = a.wk + (DH * h + f) * 64 + 4 * c4;
            const float* wvr = a.wv + (DH * h + f) * 64 + 4 * c4;
            kp[f] = fmaf(x.x, wkr[0], fmaf(x.y, wkr[1], fmaf(x.z, wkr[2], fmaf(x.w, wkr[3], kp[f]))));
            vp[f] = fmaf(x.x, wvr[0], fmaf(x.y, wvr[1], fmaf(x.z, wvr[2], fmaf(x.w, wvr[3], vp[f]))));
          }
        }
      } else {
        for (int c = 0; c < a.dk; ++c) {
          const float x = Xb[(int64_t)n * a.dk + c];
#pragma unroll
          for (int f = 0; f < DH; ++f) {
            kp[f] = fmaf(x, a.wk[(DH * h + f) * a.dk + c], kp[f]);
            vp[f] = fmaf(x, a.wv[(DH * h + f) * a.dk + c], vp[f]);
          }
        }
      }
      float4* dst = reinterpret_cast<float4*>(&sKV[h][lane][0]);
      dst[0] = float4{kp[0], kp[1], kp[2], kp[3]};
      dst[1] = float4{kp[4], kp[5], kp[6], kp[7]};
      dst[2] = float4{vp[0], vp[1], vp[2], vp[3]};
      dst[3] = float4{vp[4], vp[5], vp[6], vp[7]};
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): the wave's own LDS writes
    __builtin_amdgcn_wave_barrier();
    // ---- this lane's query over the tile's points ----
    const int cnt = n_hi - n0 < TP ? n_hi - n0 : TP;
    float sc[TP];
    float mt = -INFINITY;
#pragma unroll
    for (int i = 0; i < TP; ++i) {
      const float4* kr = reinterpret_cast<const float4*>(&sKV[h][i][0]);
      const float4 k0 = kr[0], k1 = kr[1];
      float s = qp[0] * k0.x;
      s = fmaf(qp[1], k0.y, s); s = fmaf(qp[2], k0.z, s); s = fmaf(qp[3], k0.w, s);
      s = fmaf(qp[4], k1.x, s); s = fmaf(qp[5], k1.y, s); s = fmaf(qp[6], k1.z, s);
      s = fmaf(qp[7], k1.w, s);
      s = i < cnt ? s : -INFINITY;
      sc[i] = s;
      mt = fmaxf(mt, s);
    }
    const float mnew = fmaxf(mrun, mt);
    const float alpha = exp2f(mrun - mnew);
    lrun *= alpha;
#pragma unroll
    for (int f = 0; f < DH; ++f) acc[f] *= alpha;
    mrun = mnew;
#pragma unroll
    for (int i = 0; i < TP; ++i) {
      const float p = exp2f(sc[i] - mnew);       // (masked points: exp2(-inf) = 0)
      lrun += p;
      const float4* vr = reinterpret_cast<const float4*>(&sKV[h][i][DH]);
      const float4 v0 = vr[0], v1 = vr[1];
      acc[0] = fmaf(p, v0.x, acc[0]); acc[1] = fmaf(p, v0.y, acc[1]);
      acc[2] = fmaf(p, v0.z, acc[2]); acc[3] = fmaf(p, v0.w, acc[3]);
      acc[4] = fmaf(p, v1.x, acc[4]); acc[5] = fmaf(p, v1.y, acc[5]);
      acc[6] = fmaf(p, v1.z, acc[6]); acc[7] = fmaf(p, v1.w, acc[7]);
    }
    __builtin_amdgcn_wave_barrier();             // the tile is consumed before it is overwritten
  }
  const int64_t o = (((int64_t)b * a.S + sp) * H + h) * MQ + q;
  a.Mp[o] = mrun;
  a.Lp[o] = lrun;
  float4* op = reinterpret_cast<float4*>(a.Op + o * DH);
  op[0] = float4{acc[0], acc[1], acc[2], acc[3]};
  op[1] = float4{acc[4], acc[5], acc[6], acc[7]};
}

// per set: merge the point ranges, O = Qp + A V (modules.py:29), H = O + relu(fc_o(O)) (:31)
__global__ __launch_bounds__(512) void k_sd_fq_epi(const SdArgs a) {
  __shared__ float sO[MQ][D + 1];
  const int tid = threadIdx.x, lane = tid & 63;
  const int h = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x, q = lane;
  float mx = -INFINITY;
  for (int s = 0; s < a.S; ++s) mx = fmaxf(mx, a.Mp[(((int64_t)b * a.S + s) * H + h) * MQ + q]);
  float l = 0.f, acc[DH];
#pragma unroll
  for (int f = 0; f < DH; ++f) acc[f] = 0.f;
  for (int s = 0; s < a.S; ++s) {
    const int64_t o = (((int64_t)b * a.S + s) * H + h) * MQ + q;
    const float w = a.Mp[o] == -INFINITY ? 0.f : exp2f(a.Mp[o] - mx);
    l = fmaf(w, a.Lp[o], l);
#pragma unroll
    for (int f = 0; f < DH; ++f) acc[f] = fmaf(w, a.Op[o * DH + f], acc[f]);
  }
  const float inv = 1.0f / l;
  float o8[DH];
#pragma unroll
  for (int f = 0; f < DH; ++f) {
    o8[f] = q < a.m ? a.Qp[q * D + DH * h + f] + acc[f] * inv : 0.f;
    sO[q][DH * h + f] = o8[f];
  }
  __syncthreads();
  if (q < a.m) {
    float z[DH];
#pragma unroll
    for (int f = 0; f < DH; ++f) z[f] = a.bo[DH * h + f];
    for (int c = 0; c < D; ++c) {
      const float x = sO[q][c];
#pragma unroll
      for (int f = 0; f < DH; ++f) z[f] = fmaf(x, a.wo[(DH * h + f) * D + c], z[f]);
    }
    float* y = a.Y + ((int64_t)b * a.m + q) * D + DH * h;
#pragma unroll
    for (int f = 0; f < DH; ++f) y[f] = o8[f] + fmaxf(z[f], 0.f);
  }
}

// ---------------------------------------------------------------------------------------------
// many queries (the points) over the m <= 64 keys of their set
// ---------------------------------------------------------------------------------------------
// (Code size: a lane's 64 Qp / O values would have to sit in registers with every index static,
//  i.e. all eight heads and both 64 x 64 projections fully unrolled - ~15 k instructions, twice the
//  instruction cache.  They live in LDS as sQO[feature][thread] (conflict-free: consecutive lanes,
//  consecutive banks) and the loops over heads and over output-feature groups stay rolled.)
template <int DQ>       // DQ = 64, or 4 for dq <= 4
__global__ __launch_bounds__(256) void k_sd_mq(const SdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem_mq[];
  float (*sK)[D] = reinterpret_cast<float (*)[D]>(smem_mq);                  // [MQ][D] (scaled)
  float (*sV)[D] = reinterpret_cast<float (*)[D]>(smem_mq + MQ * D);
  float (*sQO)[256] = reinterpret_cast<float (*)[256]>(smem_mq + 2 * MQ * D); // [D][256 threads]
  const int tid = threadIdx.x;
  const int b = blockIdx.x;
  const int nk = a.m;
  // ---- the set's keys: Kp = Hk Wk^T + bk, Vp = Hk Wv^T + bv (thread: key tid % 64, 16 features) ----
  {
    const int kk = tid & 63, f0 = __builtin_amdgcn_readfirstlane((tid >> 6) * 16);
    if (kk < nk) {
      const float* hr = a.Kin + ((int64_t)b * nk + kk) * D;
      float kp[16], vp[16];
#pragma unroll
      for (int f = 0; f < 16; ++f) { kp[f] = a.bk[f0 + f]; vp[f] = a.bv[f0 + f]; }
      for (int c = 0; c < D; ++c) {
        const float x = hr[c];
#pragma unroll
        for (int f = 0; f < 16; ++f) {
          kp[f] = fmaf(x, a.wk[(f0 + f) * D + c], kp[f]);
          vp[f] = fmaf(x, a.wv[(f0 + f) * D + c], vp[f]);
        }
      }
#pragma unroll
      for (int f = 0; f < 16; ++f) { sK[kk][f0 + f] = kp[f] * a.sl2e; sV[kk][f0 + f] = vp[f]; }
    } else {      // rows past the last key: finite, so that their zero probabilities stay zero
#pragma unroll
      for (int f = 0; f < 16; ++f) { sK[kk][f0 + f] = 0.f; sV[kk][f0 + f] = 0.f; }
    }
  }
  __syncthreads();
  const int n = blockIdx.y * 256 + tid;
  if (n >= a.N) return;
  // ---- Qp = Wq x + bq (the weight rows are uniform over the workgroup) ----
  if (DQ == 64) {
    float x[64];
    const float4* xr = reinterpret_cast<const float4*>(a.Qin + ((int64_t)b * a.N + n) * 64);
#pragma unroll
    for (int c4 = 0; c4 < 16; ++c4) {
      const float4 v = xr[c4];
      x[4 * c4] = v.x; x[4 * c4 + 1] = v.y; x[4 * c4 + 2] = v.z; x[4 * c4 + 3] = v.w;
    }
#pragma unroll 1
    for (int f4 = 0; f4 < D; f4 += 4) {
      float acc[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = a.bq[f4 + e];
#pragma unroll
      for (int c = 0; c < 64; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = fmaf(x[c], a.wq[(f4 + e) * 64 + c], acc[e]);
#pragma unroll
      for (int e = 0; e < 4; ++e) sQO[f4 + e][tid] = acc[e];
    }
  } else {
    float x[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) x[c] = c < a.dq ? a.Qin[((int64_t)b * a.N + n) * a.dq + c] : 0.f;
#pragma unroll 1
    for (int f = 0; f < D; ++f) {
      float acc = a.bq[f];
      for (int c = 0; c < a.dq; ++c) acc = fmaf(x[c], a.wq[f * a.dq + c], acc);
      sQO[f][tid] = acc;
    }
  }
  // ---- per head: scores over the keys, softmax, O_h = Qp_h + P V_h (in place) ----
#pragma unroll 1
  for (int h = 0; h < H; ++h) {
    float qh[DH];
#pragma unroll
    for (int f = 0; f < DH; ++f) qh[f] = sQO[DH * h + f][tid];
    float sc[MQ];
    float mt = -INFINITY;
#pragma unroll
    for (int kk = 0; kk < MQ; ++kk) {
      const float4* kr = reinterpret_cast<const float4*>(&sK[kk][DH * h]);
      const float4 k0 = kr[0], k1 = kr[1];
      float s = qh[0] * k0.x;
      s = fmaf(qh[1], k0.y, s); s = fmaf(qh[2], k0.z, s); s = fmaf(qh[3], k0.w, s);
      s = fmaf(qh[4], k1.x, s); s = fmaf(qh[5], k1.y, s); s = fmaf(qh[6], k1.z, s);
      s = fmaf(qh[7], k1.w, s);
      s = kk < nk ? s : -INFINITY;
      sc[kk] = s;
      mt = fmaxf(mt, s);
    }
    float l = 0.f, o[DH];
#pragma unroll
    for (int f = 0; f < DH; ++f) o[f] = 0.f;
#pragma unroll
    for (int kk = 0; kk < MQ; ++kk) {
      const float p = exp2f(sc[kk] - mt);
      l += p;
      const float4* vr = reinterpret_cast<const float4*>(&sV[kk][DH * h]);
      const float4 v0 = vr[0], v1 = vr[1];
      o[0] = fmaf(p, v0.x, o[0]); o[1] = fmaf(p, v0.y, o[1]); o[2] = fmaf(p, v0.z, o[2]);
      o[3] = fmaf(p, v0.w, o[3]); o[4] = fmaf(p, v1.x, o[4]); o[5] = fmaf(p, v1.y, o[5]);
      o[6] = fmaf(p, v1.z, o[6]); o[7] = fmaf(p, v1.w, o[7]);
    }
    const float inv = 1.0f / l;
#pragma unroll
    for (int f = 0; f < DH; ++f) sQO[DH * h + f][tid] = fmaf(o[f], inv, qh[f]);
  }
  // ---- Y = O + relu(Wo O + bo) ----
  float ov[D];
#pragma unroll
  for (int c = 0; c < D; ++c) ov[c] = sQO[c][tid];
  float* y = a.Y + ((int64_t)b * a.N + n) * D;
#pragma unroll 1
  for (int f4 = 0; f4 < D; f4 += 4) {
    float acc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = a.bo[f4 + e];
#pragma unroll
    for (int c = 0; c < D; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = fmaf(ov[c], a.wo[(f4 + e) * D + c], acc[e]);
    *reinterpret_cast<float4*>(y + f4) =
        float4{sQO[f4][tid] + fmaxf(acc[0], 0.f), sQO[f4 + 1][tid] + fmaxf(acc[1], 0.f),
               sQO[f4 + 2][tid] + fmaxf(acc[2], 0.f), sQO[f4 + 3][tid] + fmaxf(acc[3], 0.f)};
  }
}

inline bool all_f32(const pca_mab_shape& s) {
  return s.q_dtype == PCA_F32 && s.k_dtype == PCA_F32 && s.y_dtype == PCA_F32;
}
inline int fq_splits(const pca_mab_shape& s) {
  int S = 1;
  const int tiles = (int)cdiv(s.nk, TP);
  while (S * 2 <= tiles && s.B * S < 512 && S < 64) S *= 2;
  return S;
}

}  // namespace

// 1 = many queries (mab1 of an ISAB), 2 = few shared queries (mab0 / PMA), 0 = not this family
int sd64_kind(const pca_mab_shape& s) {
  if (!(s.d == D && s.h == H && s.ln == 0 && all_f32(s))) return 0;
  if (s.q_shared == 0)
    return (s.nk >= 1 && s.nk <= MQ && s.dk == D && (s.dq == D || s.dq <= 4) &&
            s.k_lengths == nullptr) ? 1 : 0;
  return (s.nq >= 1 && s.nq <= MQ && s.dq == D && (s.dk == D || s.dk <= 4)) ? 2 : 0;
}
size_t sd64_fwd_ws_bytes(const pca_mab_shape& s) {
  if (sd64_kind(s) != 2) return 256;
  const size_t rows = (size_t)s.B * fq_splits(s) * H * MQ;
  return align256(rows * DH * sizeof(float)) + 2 * align256(rows * sizeof(float)) +
         align256((size_t)MQ * D * sizeof(float));
}
int sd64_fwd(const pca_mab_shape& s, const float* Q, const float* K, const pca_mab_params& p,
             float* Y, void* ws, hipStream_t st) {
  const int kind = sd64_kind(s);
  PCA_REQUIRE(kind != 0, "sd64_fwd: unsupported shape");
  SdArgs a{};
  a.Qin = Q; a.Kin = K;
  a.wq = p.wq; a.bq = p.bq; a.wk = p.wk; a.bk = p.bk; a.wv = p.wv; a.bv = p.bv; a.wo = p.wo; a.bo = p.bo;
  a.Y = Y; a.B = s.B; a.dq = s.dq; a.dk = s.dk; a.lengths = s.k_lengths;
  a.sl2e = 1.4426950408889634f / sqrtf((float)D);
  if (kind == 1) {
    a.N = s.nq; a.m = s.nk;
    const dim3 grid((unsigned)s.B, (unsigned)cdiv(s.nq, 256));
    PCA_REQUIRE(grid.y <= 65535, "sd64_fwd: too many points per set (%d)", s.nq);
    const size_t lds = (size_t)(2 * MQ * D + D * 256) * sizeof(float);       // 96 KiB
    static std::once_flag once;
    std::call_once(once, [] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_sd_mq<64>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_sd_mq<4>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    if (s.dq == D) hipLaunchKernelGGL(k_sd_mq<64>, grid, dim3(256), lds, st, a);
    else hipLaunchKernelGGL(k_sd_mq<4>, grid, dim3(256), lds, st, a);
    return check_launch("k_sd_mq");
  }
  PCA_REQUIRE(ws != nullptr, "sd64_fwd: scratch required");
  a.N = s.nk; a.m = s.nq; a.S = fq_splits(s);
  const size_t rows = (size_t)s.B * a.S * H * MQ;
  Carver c(ws);
  a.Op = c.take<float>(rows * DH);
  a.Mp = c.take<float>(rows);
  a.Lp = c.take<float>(rows);
  a.Qp = c.take<float>((size_t)MQ * D);
  if (s.dk == D) hipLaunchKernelGGL(k_sd_fq<64>, dim3(s.B, a.S), dim3(512), 0, st, a);
  else hipLaunchKernelGGL(k_sd_fq<4>, dim3(s.B, a.S), dim3(512), 0, st, a);
  PCA_TRY(check_launch("k_sd_fq"));
  hipLaunchKernelGGL(k_sd_fq_epi, dim3(s.B), dim3(512), 0, st, a);
  return check_launch("k_sd_fq_epi");
}

}  // namespace pca
