// Fused bf16-MFMA forward of the "few queries, many keys" MAB: ISAB's mab0(I, X)
// (set_transformer-master/modules.py:52) and PMA's mab(S, X) (modules.py:63), where the
// query is a learned [m, d] tensor shared by all sets.
//
// Reassociation (exact algebra, SURVEY.md 8d "reassociation caveat"): with Qp = fc_q(I),
//   scores_h = Qp_h (X Wk_h^T + bk_h)^T = (Qp_h Wk_h) X^T + const_per_query
//   A_h (X Wv_h^T + bv_h)               = (A_h X) Wv_h^T + bv_h        (rows of A sum to 1)
// and the per-query constant cancels in the softmax, so the N keys are never projected:
//   G = [Qp_h Wk_h]_h   (R = h*m rows, batch invariant)   S = G X^T   A = softmax_N(S/sqrt d)
//   T = A X  [R, dk]    O_h = Qp_h + T_h Wv_h^T + bv_h     H = O + relu(fc_o(O))
// This is flash attention whose keys AND values are the same X tile: X is read once.
//
//   k_mab0_prep   Qp, G (bf16, scale*log2e folded in) -- once per call, tiny
//   k_mab0_attn   (PMA: 4 score rows) each wave streams its own 32-point tiles: S = X G^T on
//                 the MFMA (points on accumulator rows, so the softmax statistics of a query
//                 row are lane-local + 2 cross-lane steps), online softmax, T += P^T X with the
//                 probability tile fed back as the A operand straight from the accumulators
//                 and X^T read from the same LDS tile through ds_read_tr16_b64; the waves'
//                 partials are merged through per-wave LDS slabs
//   k_mab0_attn_h4 (ISAB: 4 heads x 16 queries) same arithmetic, but every wave owns one head
//                 over X tiles shared by the workgroup: no cross-wave merge
//   k_mab0_attn_small   layer 1 (dk = din <= 4): exact fp32 on the vector ALU, the rows of a
//                 set shared by two workgroups
//   k_mab0_epi    per set: O, Z, H (fp32 VALU; 2*m*d*(dk+d) MACs)
#include "mab1_bf16.hpp"
#include "pack_body.hpp"
#include "pma_head_bodies.hpp"

#include <math.h>

#include <mutex>

namespace pca {

namespace {

__device__ __forceinline__ int tr_off(int row, int ch) {
  return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
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

// ---------------------------------------------------------------------------------
// Query-side preparation of up to 3 MABs in ONE launch (they only depend on parameters):
//   Qp[m][d] = I Wq^T + bq (fp32) ; G[h*m][dk] = scale*log2(e) * Qp_h Wk_h (fp32 + bf16, padding
//   rows zero) ; GtP[dk][Rp] = K-permuted transpose of G for the backward.
// blockIdx.y = MAB, blockIdx.x = query row q (blocks beyond m exit).
// ---------------------------------------------------------------------------------
constexpr int PREP_SPARE = 16;
__device__ __forceinline__ void mab0_prep_body(const Mab0PrepJob& a, float* sq, int q);

__global__ __launch_bounds__(256) void k_mab0_prep(const Mab0PrepJobs jobs) {
  extern __shared__ float sq[];            // Qp row [d]
  mab0_prep_body(jobs.j[blockIdx.y], sq, blockIdx.x);
}

// weight images (as k_prep_jobs, mab1_bf16.hip) and query-side tensors in one grid:
// rows [0, Q.n) of blockIdx.y are query-side jobs, then the weight-image jobs, then (rider rows)
// the step's deferred point-set pack - independent of everything else in this launch
__global__ __launch_bounds__(256) void k_prep_all(const PrepJobs W, const Mab0PrepJobs Q,
                                                  const PackJob P) {
  extern __shared__ float sq[];
  if ((int)blockIdx.y >= Q.n + W.n) {
    const int blk = ((int)blockIdx.y - Q.n - W.n) * gridDim.x + blockIdx.x;
    if (blk < P.bx * P.B) pack_body(P, blk);
    return;
  }
  if ((int)blockIdx.y < Q.n) {
    mab0_prep_body(Q.j[blockIdx.y], sq, blockIdx.x);
    return;
  }
  const PrepJob jb = W.j[blockIdx.y - Q.n];
  const int rows = jb.rows, cols = jb.cols, mode = jb.mode;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * cols) return;
  float v;
  if (mode == 4) {                     // clear job: rows * cols 16-bit words of zeros (no source)
    v = 0.f;
  } else if (mode <= 1) {
    const int n = idx / cols, k = idx - n * cols;
    const int kk = mode == 1 ? (k & ~31) + perm32(k & 31) : k;
    v = jb.src[(int64_t)n * cols + kk];
  } else {
    const int n = idx / rows, k = idx - n * rows;
    const int kk = mode == 2 ? (k & ~31) + perm32(k & 31) : k;
    v = jb.src[(int64_t)kk * cols + n];
  }
  jb.dst[idx] = (__bf16)v;
}

// d > 128: blockIdx.x = q * h + j - one workgroup per (query row, head).  A thread per output
// feature walks Wq[f][:] with a 1 KiB stride between lanes in 16 dependent batches, then 8 more
// for the rows of G: 45 us for PMA's single query row.  Here a wave owns 8 features of the head
// (lanes over the contraction: coalesced, all 8 rows in flight), and the head's G row follows
// from the 32 Qp values in LDS with coalesced reads of Wk.
__device__ __forceinline__ void mab0_prep_head(const Mab0PrepJob& a, float* sq, int q, int j) {
  const int m = a.m, d = a.d, dq = a.dq, dk = a.dk, dh = a.d / a.h;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int f0 = wv * 8; f0 < dh; f0 += 32) {
    float acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = 0.f;
    for (int c = lane; c < dq; c += 64) {
      const float x = a.I[q * dq + c];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (f0 + u < dh) acc[u] = fmaf(x, a.Wq[(int64_t)(j * dh + f0 + u) * dq + c], acc[u]);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int sh = 1; sh < 64; sh <<= 1) acc[u] += __shfl_xor(acc[u], sh);
      if (lane == 0 && f0 + u < dh) {
        const float v = acc[u] + a.bq[j * dh + f0 + u];
        sq[f0 + u] = v;
        a.Qp[q * d + j * dh + f0 + u] = v;
      }
    }
  }
  __syncthreads();
  if (a.Gf != nullptr) {
    const int r = j * m + q;
    for (int c = threadIdx.x; c < dk; c += 256) {
      float acc = 0.f;
      for (int f = 0; f < dh; f += 16) {
        float wv16[16];
#pragma unroll
        for (int u = 0; u < 16; ++u)
          wv16[u] = (f + u < dh) ? a.Wk[(int64_t)(j * dh + f + u) * dk + c] : 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u)
          if (f + u < dh) acc = fmaf(sq[f + u], wv16[u], acc);
      }
      acc *= a.sl2e;
      a.Gf[r * dk + c] = acc;
      if (a.Gb != nullptr) a.Gb[r * dk + c] = (__bf16)acc;
      if (a.GtP != nullptr) {
        const int rb32 = r & ~31, ro = r & 31;
        int pos = 0;
#pragma unroll
        for (int p = 0; p < 32; ++p)
          if (perm32(p) == ro) pos = p;
        a.GtP[c * a.Rp + rb32 + pos] = (__bf16)acc;
      }
    }
    if (q == 0 && j == 0 && a.Gb != nullptr) {         // zero the padding rows (PMA: R = h < 32)
      const int R = a.h * m;
      for (int o = threadIdx.x; o < (a.Rp - R) * dk; o += 256) {
        const int rr = R + o / dk, c = o % dk;
        a.Gb[rr * dk + c] = (__bf16)0.f;
        if (a.GtP != nullptr) {
          const int rb32 = rr & ~31, ro = rr & 31;
          int pos = 0;
#pragma unroll
          for (int p = 0; p < 32; ++p)
            if (perm32(p) == ro) pos = p;
          a.GtP[c * a.Rp + rb32 + pos] = (__bf16)0.f;
        }
      }
    }
  }
}

__device__ __forceinline__ void mab0_prep_body(const Mab0PrepJob& a, float* sq, int q) {
  if (a.d > 128) {
    const int nx = a.m * a.h;
    if (q < nx) {
      mab0_prep_head(a, sq, q / a.h, q % a.h);
      return;
    }
    q = q - nx + a.m;                        // spare workgroups: as below
  }
  if (q >= a.m) {
    // spare workgroups: WvT[c][f] = Wv[f][c], WoT[c][f] = Wo[f][c]
    const int sp = q - a.m;
    if (sp >= PREP_SPARE || a.WvT == nullptr) return;
    const int nv = a.d * a.dk, no = a.d * a.d;
    // strided (uncoalesced) reads: keep 8 of them in flight per thread
    for (int o0 = sp * 256 + threadIdx.x; o0 < nv + no; o0 += 8 * PREP_SPARE * 256) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int o = o0 + u * PREP_SPARE * 256;
        v[u] = 0.f;
        if (o < nv) {
          const int c = o / a.d, f = o - c * a.d;
          v[u] = a.Wv[f * a.dk + c];
        } else if (o < nv + no) {
          const int oo = o - nv;
          const int c = oo / a.d, f = oo - c * a.d;
          v[u] = a.Wo[f * a.d + c];
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int o = o0 + u * PREP_SPARE * 256;
        if (o < nv) a.WvT[o] = v[u];
        else if (o < nv + no) a.WoT[o - nv] = v[u];
      }
    }
    return;
  }
  const int m = a.m, d = a.d, dq = a.dq, dk = a.dk, h = a.h;
  const int R = h * m;
  // The Wk values of the G stage do not depend on the Qp stage: request them now (head dim 32, at
  // most two outputs per thread), so that the launch is two L2 round trips deep instead of six
  const int dh_ = d / h;
  const bool pre = a.Gf != nullptr && dh_ == 32 && h * dk <= 512;
  float wk_pre[2][32];
  if (pre) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      int o = threadIdx.x + 256 * it;
      o = o < h * dk ? o : h * dk - 1;
      const int j = o / dk, c = o - j * dk;
#pragma unroll
      for (int u = 0; u < 32; ++u) wk_pre[it][u] = a.Wk[(j * 32 + u) * dk + c];
    }
  }
  if (d <= 128 && dq == 128) {
    // as below, the thread's 64 weights and the 64 inputs of its half in one batch
    __shared__ float part[128];
    const int f = threadIdx.x & 127, kh = threadIdx.x >> 7;
    const int fc = f < d ? f : d - 1;
    float4 w4[16], x4[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      w4[u] = *reinterpret_cast<const float4*>(a.Wq + fc * dq + kh * 64 + 4 * u);
      x4[u] = *reinterpret_cast<const float4*>(a.I + q * dq + kh * 64 + 4 * u);
    }
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      acc = fmaf(x4[u].x, w4[u].x, acc);
      acc = fmaf(x4[u].y, w4[u].y, acc);
      acc = fmaf(x4[u].z, w4[u].z, acc);
      acc = fmaf(x4[u].w, w4[u].w, acc);
    }
    if (kh == 1 && f < d) part[f] = acc;
    __syncthreads();
    if (kh == 0 && f < d) {
      acc += part[f] + a.bq[f];
      sq[f] = acc;
      a.Qp[q * d + f] = acc;
    }
  } else if (d <= 128 && dq % 32 == 0) {
    // both halves of the workgroup work on the d outputs: half kh takes half of the
    // contraction (the dot products are chains of dependent L2 round trips)
    __shared__ float part[128];
    const int f = threadIdx.x & 127, kh = threadIdx.x >> 7;
    float acc = 0.f;
    if (f < d) {
      const int c0 = kh * (dq / 2), c1 = c0 + dq / 2;
      for (int c = c0; c < c1; c += 16) {
        float wv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) wv[u] = a.Wq[f * dq + c + u];
#pragma unroll
        for (int u = 0; u < 16; ++u) acc = fmaf(a.I[q * dq + c + u], wv[u], acc);
      }
      if (kh == 1) part[f] = acc;
    }
    __syncthreads();
    if (kh == 0 && f < d) {
      acc += part[f] + a.bq[f];
      sq[f] = acc;
      a.Qp[q * d + f] = acc;
    }
  } else {
  for (int f = threadIdx.x; f < d; f += 256) {
    float acc = a.bq[f];
    int c = 0;
    for (; c + 16 <= dq; c += 16) {          // 16 independent loads in flight
      float wv[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) wv[u] = a.Wq[f * dq + c + u];
#pragma unroll
      for (int u = 0; u < 16; ++u) acc = fmaf(a.I[q * dq + c + u], wv[u], acc);
    }
    for (; c < dq; ++c) acc += a.I[q * dq + c] * a.Wq[f * dq + c];
    sq[f] = acc;
    a.Qp[q * d + f] = acc;
  }
  }
  __syncthreads();
  if (a.Gf == nullptr) return;             // caller only wants Qp (keys are projected: d = 256)
  const int dh = d / h;
#pragma unroll 2
  for (int it = 0; it * 256 < h * dk; ++it) {
    const int o = threadIdx.x + 256 * it;
    if (o >= h * dk) break;
    const int j = o / dk, c = o - j * dk;
    float acc = 0.f;
    if (pre) {
#pragma unroll
      for (int u = 0; u < 32; ++u) acc = fmaf(sq[j * 32 + u], wk_pre[it & 1][u], acc);
    } else {
    for (int f = 0; f < dh; f += 16) {
      float wv[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) wv[u] = (f + u < dh) ? a.Wk[(j * dh + f + u) * dk + c] : 0.f;
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (f + u < dh) acc = fmaf(sq[j * dh + f + u], wv[u], acc);
    }
    }
    acc *= a.sl2e;
    const int r = j * m + q;
    a.Gf[r * dk + c] = acc;
    if (a.Gb != nullptr) a.Gb[r * dk + c] = (__bf16)acc;
    if (a.GtP != nullptr) {
      const int rb32 = r & ~31, ro = r & 31;
      int pos = 0;
#pragma unroll
      for (int p = 0; p < 32; ++p)
        if (perm32(p) == ro) pos = p;
      a.GtP[c * a.Rp + rb32 + pos] = (__bf16)acc;
    }
  }
  if (q == 0 && a.Gb != nullptr) {         // zero the padding rows (PMA: R = h < 32)
    for (int o = threadIdx.x; o < (a.Rp - R) * dk; o += 256) {
      const int r = R + o / dk, c = o % dk;
      a.Gb[r * dk + c] = (__bf16)0.f;
      if (a.GtP != nullptr) {
        const int rb32 = r & ~31, ro = r & 31;
        int pos = 0;
#pragma unroll
        for (int p = 0; p < 32; ++p)
          if (perm32(p) == ro) pos = p;
        a.GtP[c * a.Rp + rb32 + pos] = (__bf16)0.f;
      }
    }
  }
}

// ---------------------------------------------------------------------------------
// attention over the points, dk == 128
// ---------------------------------------------------------------------------------
struct Mab0AttnArgs {
  const void* X;        // [B, N, 128] fp32, or bf16 when ABF
  const __bf16* Gb;     // [16*RB][128]  (scale*log2e folded in; padding rows zero)
  float* Tp;            // [B][S][R][128] fp32 partial sums  sum_n 2^(s_n - M) x_n
  float* Mp;            // [B][S][R]      partial maxima M (log2 domain)
  float* Lp;            // [B][S][R]      partial sums    sum_n 2^(s_n - M)
  int B, N, R, S;       // S = point splits per set (gridDim.y); merged by k_mab0_epi
  const int32_t* lengths;   // [B] valid points per set, or null
};

template <int RB, bool ABF>
__global__ __launch_bounds__(256, 1) void k_mab0_attn(const Mab0AttnArgs a) {
  constexpr int DK = 128, FT = DK / 16, KS = DK / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sG = smem;                                    // [16 RB][256 B] tr_off image rows
  char* sX = sG + 16 * RB * 256;                      // 4 waves x 32 rows x 256 B
  // X tiles during the loop; afterwards one fp32 [16 RB][128] slab per wave for the merge
  constexpr int XBYTES = (4 * 32 * 256 > 4 * 16 * RB * DK * 4) ? 4 * 32 * 256 : 4 * 16 * RB * DK * 4;
  float* sAl = reinterpret_cast<float*>(sX + XBYTES);   // 4 waves x 16 RB
  float* sM = sAl + 4 * 16 * RB;                      // merge: [4][16 RB] m, then l
  float* sL = sM + 4 * 16 * RB;
  float* sT = reinterpret_cast<float*>(sX);           // merge buffer aliases the X tiles

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int b = blockIdx.x, sp = blockIdx.y;
  // this workgroup's point range (multiples of 128 so that tiles stay aligned)
  const int per = (int)(((int64_t)(a.N + 127) / 128 + a.S - 1) / a.S) * 128;
  // variable-size sets: points at and beyond len never enter; a range without any point
  // leaves the (-inf, 0, 0) partial, which the merge skips
  int len = a.N;
  if (a.lengths != nullptr) len = a.lengths[b] < a.N ? a.lengths[b] : a.N;
  const int n_lo = sp * per, n_hi = (n_lo + per < len) ? n_lo + per : len;

  for (int c = tid; c < 16 * RB * 16; c += 256) {
    const int row = c >> 4, ch = c & 15;
    *reinterpret_cast<uint4*>(sG + tr_off(row, ch)) =
        *reinterpret_cast<const uint4*>(a.Gb + (int64_t)row * DK + ch * 8);
  }
  __syncthreads();

  char* myX = sX + wave * 32 * 256;
  float* myAl = sAl + wave * 16 * RB;
  float mrow[RB], lrow[RB];
  f32x4 T[RB][FT];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    mrow[rb] = -INFINITY;
    lrow[rb] = 0.f;
#pragma unroll
    for (int ft = 0; ft < FT; ++ft) T[rb][ft] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // bf16 activations: the wave's next X tile is fetched into registers while the current one is worked
  // on (as k_mab0_bwd does; every tile used to start with an exposed round trip to memory)
  bf16x8 nx[8];
  auto fetch_tile = [&](int n0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = lane + 64 * e;
      nx[e] = ld_x8_clamped(a.X, (int64_t)b * a.N, n0 + (c >> 4), n_hi, DK, c & 15);
    }
  };
  if (ABF && n_lo + wave * 32 < n_hi) fetch_tile(n_lo + wave * 32);
  for (int n0 = n_lo + wave * 32; n0 < n_hi; n0 += 128) {
    // stage 32 rows of X (fp32 -> bf16)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = lane + 64 * e;
      const int row = c >> 4, ch = c & 15;
      const bf16x8 v = ABF ? zero_unless(n0 + row < n_hi, nx[e])
                           : ld_x8_guard<ABF>(a.X, (int64_t)b * a.N, n0 + row, n_hi, DK, ch);
      *reinterpret_cast<bf16x8*>(myX + tr_off(row, ch)) = v;
    }
    if (ABF && n0 + 128 < n_hi) fetch_tile(n0 + 128);
    bf16x8 xrow[2][KS], xtr[FT];
#pragma unroll
    for (int pb = 0; pb < 2; ++pb)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        xrow[pb][ks] = *reinterpret_cast<const bf16x8*>(myX + tr_off(16 * pb + r, 4 * ks + g));
#pragma unroll
    for (int ft = 0; ft < FT; ++ft) xtr[ft] = tr_frag(myX, ft, lane);

#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      f32x4 s[2];
#pragma unroll
      for (int pb = 0; pb < 2; ++pb) {
        s[pb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
          s[pb] = mfma32(xrow[pb][ks],
                         *reinterpret_cast<const bf16x8*>(sG + tr_off(16 * rb + r, 4 * ks + g)),
                         s[pb]);
      }
      // rows of s = points 16 pb + 4 g + e ; column = query row 16 rb + r
      float mt = -INFINITY;
#pragma unroll
      for (int pb = 0; pb < 2; ++pb)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (n0 + 16 * pb + 4 * g + e >= n_hi) s[pb][e] = -INFINITY;
          mt = fmaxf(mt, s[pb][e]);
        }
      mt = wave16_max(mt);
      const float mnew = fmaxf(mrow[rb], mt);      // finite: every tile has >= 1 live point
      const float alpha = exp2f(mrow[rb] - mnew);
      float ls = 0.f;
#pragma unroll
      for (int pb = 0; pb < 2; ++pb)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s[pb][e] = exp2f(s[pb][e] - mnew);
          ls += s[pb][e];
        }
      ls = wave16_sum(ls);
      lrow[rb] = lrow[rb] * alpha + ls;
      mrow[rb] = mnew;
      // the T tiles hold query rows 4g+e on their accumulator rows: fetch their alphas
      if (g == 0) myAl[16 * rb + r] = alpha;
      const float4 a4 = *reinterpret_cast<const float4*>(&myAl[16 * rb + 4 * g]);
      const bf16x8 pa = pack8(s[0], s[1]);
#pragma unroll
      for (int ft = 0; ft < FT; ++ft) {
        T[rb][ft][0] *= a4.x; T[rb][ft][1] *= a4.y; T[rb][ft][2] *= a4.z; T[rb][ft][3] *= a4.w;
        T[rb][ft] = mfma32(pa, xtr[ft], T[rb][ft]);
      }
    }
  }

  // ---- merge the four waves' partial (m, l, T): per-wave slabs, plain LDS stores ----
  __syncthreads();                       // X tiles are dead: the slabs alias them
  if (g == 0) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      sM[wave * 16 * RB + 16 * rb + r] = mrow[rb];
      sL[wave * 16 * RB + 16 * rb + r] = lrow[rb];
    }
  }
  __syncthreads();
  float* mySlab = sT + wave * 16 * RB * DK;
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    // factor of this wave for query rows 4g+e of block rb (accumulator-row layout)
    float f4[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int rr = 16 * rb + 4 * g + e;
      const float M = fmaxf(fmaxf(sM[rr], sM[16 * RB + rr]),
                            fmaxf(sM[2 * 16 * RB + rr], sM[3 * 16 * RB + rr]));
      const float mine = sM[wave * 16 * RB + rr];
      f4[e] = (mine == -INFINITY) ? 0.f : exp2f(mine - M);
    }
#pragma unroll
    for (int ft = 0; ft < FT; ++ft)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        mySlab[(16 * rb + 4 * g + e) * DK + 16 * ft + r] = T[rb][ft][e] * f4[e];
  }
  __syncthreads();
  const int64_t pbase = ((int64_t)b * a.S + sp) * a.R;
  for (int i = tid; i < a.R * DK; i += 256) {
    const int rr = i / DK;
    a.Tp[(pbase + rr) * DK + (i - rr * DK)] = sT[i] + sT[16 * RB * DK + i] +
                                              sT[2 * 16 * RB * DK + i] + sT[3 * 16 * RB * DK + i];
    if (i == rr * DK) {
      float M = -INFINITY;
#pragma unroll
      for (int w = 0; w < 4; ++w) M = fmaxf(M, sM[w * 16 * RB + rr]);
      float L = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const float mw = sM[w * 16 * RB + rr];
        if (mw != -INFINITY) L += sL[w * 16 * RB + rr] * exp2f(mw - M);
      }
      a.Mp[pbase + rr] = M;
      a.Lp[pbase + rr] = L;
    }
  }
}

// ---------------------------------------------------------------------------------
// ISAB mab0 (R = 4 heads x 16 queries = 64 score rows): WAVE = HEAD.  The four waves of a
// workgroup share every X tile (staged once in LDS, double-buffered, the next tile's global
// loads in flight during the MFMAs) and each owns the 16 query rows of one head, so there is
// no cross-wave merge at all: 32 accumulator registers per lane instead of 128, 48 KiB of LDS
// instead of 150 KiB (three workgroups per CU instead of one), and the partial (T, M, L) of a
// point range leaves straight from registers.  (The one-wave-per-32-points variant above spent
// most of its 29 us on the LDS slab merge and on a single exposed load per iteration:
// 0.55 TB/s on a kernel whose only traffic is reading X once.)
// ---------------------------------------------------------------------------------
template <bool ABF>
__global__ __launch_bounds__(256) void k_mab0_attn_h4(const Mab0AttnArgs a) {
  constexpr int DK = 128, FT = DK / 16, KS = DK / 32, TILE = 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sG = smem;                                    // [64][256 B] tr_off image rows
  char* sX = sG + 64 * 256;                           // [2 buffers][2 halves][32 rows][256 B]
  float* sAl = reinterpret_cast<float*>(sX + 2 * 2 * 32 * 256);   // 4 waves x 16 alphas
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int b = blockIdx.x, sp = blockIdx.y;
  const int per = (int)(((int64_t)(a.N + 127) / 128 + a.S - 1) / a.S) * 128;
  int len = a.N;
  if (a.lengths != nullptr) len = a.lengths[b] < a.N ? a.lengths[b] : a.N;
  const int n_lo = sp * per, n_hi = (n_lo + per < len) ? n_lo + per : len;

  for (int c = tid; c < 64 * 16; c += 256) {
    const int row = c >> 4, ch = c & 15;
    *reinterpret_cast<uint4*>(sG + tr_off(row, ch)) =
        *reinterpret_cast<const uint4*>(a.Gb + (int64_t)row * DK + ch * 8);
  }
  bf16x8 v[4];
  auto fetch = [&](int n0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = tid + 256 * e;
      const int row = c >> 4, ch = c & 15;
      const int n = n0 + row;
      // (bf16: unconditional clamped loads, rows past the range zeroed by stash() - zeroed here, or
      //  loaded under the divergent `if`, every load is waited for where it is issued and nothing
      //  arrives under the current tile's arithmetic)
      if (ABF) {
        v[e] = ld_x8_clamped(a.X, (int64_t)b * a.N, n, n_hi, DK, ch);
      } else if (n < n_hi) {
        const float4* src = reinterpret_cast<const float4*>(
            reinterpret_cast<const float*>(a.X) + ((int64_t)b * a.N + n) * DK + ch * 8);
        const float4 lo = src[0], hi = src[1];
        v[e][0] = (__bf16)lo.x; v[e][1] = (__bf16)lo.y; v[e][2] = (__bf16)lo.z; v[e][3] = (__bf16)lo.w;
        v[e][4] = (__bf16)hi.x; v[e][5] = (__bf16)hi.y; v[e][6] = (__bf16)hi.z; v[e][7] = (__bf16)hi.w;
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[e][k] = (__bf16)0.f;
      }
    }
  };
  auto stash = [&](int buf, int n0) {          // n0: first point of the tile fetch() was given
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = tid + 256 * e;
      const int row = c >> 4, ch = c & 15;
      *reinterpret_cast<bf16x8*>(sX + ((buf * 2 + (row >> 5)) * 32) * 256 + tr_off(row & 31, ch)) =
          ABF ? zero_unless(n0 + row < n_hi, v[e]) : v[e];
    }
  };
  fetch(n_lo);
  stash(0, n_lo);
  __syncthreads();
  bf16x8 gf[KS];                    // this head's G rows: B operand of the score MFMAs
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
    gf[ks] = *reinterpret_cast<const bf16x8*>(sG + tr_off(16 * wave + r, 4 * ks + g));
  float* myAl = sAl + wave * 16;
  float mrow = -INFINITY, lrow = 0.f;
  f32x4 T[FT];
#pragma unroll
  for (int ft = 0; ft < FT; ++ft) T[ft] = f32x4{0.f, 0.f, 0.f, 0.f};

  int buf = 0;
  for (int n0 = n_lo; n0 < n_hi; n0 += TILE, buf ^= 1) {
    const bool more = n0 + TILE < n_hi;
    if (more) fetch(n0 + TILE);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int base = n0 + 32 * half;
      if (base >= n_hi) break;                        // uniform over the workgroup
      const char* img = sX + ((buf * 2 + half) * 32) * 256;
      f32x4 s[2];
#pragma unroll
      for (int pb = 0; pb < 2; ++pb) {
        s[pb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
          s[pb] = mfma32(*reinterpret_cast<const bf16x8*>(img + tr_off(16 * pb + r, 4 * ks + g)),
                         gf[ks], s[pb]);
      }
      // rows of s = points 16 pb + 4 g + e ; column = query row r of this head
      float mt = -INFINITY;
#pragma unroll
      for (int pb = 0; pb < 2; ++pb)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (base + 16 * pb + 4 * g + e >= n_hi) s[pb][e] = -INFINITY;
          mt = fmaxf(mt, s[pb][e]);
        }
      mt = wave16_max(mt);
      const float mnew = fmaxf(mrow, mt);             // finite: the half has >= 1 live point
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
      for (int ft = 0; ft < FT; ++ft) {
        T[ft][0] *= a4.x; T[ft][1] *= a4.y; T[ft][2] *= a4.z; T[ft][3] *= a4.w;
        T[ft] = mfma32(pa, tr_frag(img, ft, lane), T[ft]);
      }
    }
    if (more) stash(buf ^ 1, n0 + TILE);
    __syncthreads();
  }
  const int64_t pbase = ((int64_t)b * a.S + sp) * a.R + 16 * wave;
#pragma unroll
  for (int ft = 0; ft < FT; ++ft)
#pragma unroll
    for (int e = 0; e < 4; ++e) a.Tp[(pbase + 4 * g + e) * DK + 16 * ft + r] = T[ft][e];
  if (g == 0) {
    a.Mp[pbase + r] = mrow;
    a.Lp[pbase + r] = lrow;
  }
}

// ---------------------------------------------------------------------------------
// layer 1: dk = din <= 4, exact fp32.  One workgroup per set; thread = (query row r,
// point partition); R <= 256.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_mab0_attn_small(const float* __restrict__ X,
                                                         const float* __restrict__ Gf, int N,
                                                         int R, int dk, float* __restrict__ T,
                                                         float* __restrict__ LSE,
                                                         const int32_t* __restrict__ lengths) {
  constexpr int CH = PCA_POINT_CHUNK;         // points staged per chunk (16 KiB of LDS)
  __shared__ float sM[256], sL[256], sT[256][4];
  __shared__ __attribute__((aligned(16))) float sX[CH * 4];
  const int b = blockIdx.x, tid = threadIdx.x;
  // gridDim.y workgroups share the rows of a set (fills the chip when B < #CUs and halves the
  // points each thread walks): this one owns rows [row0, row0 + Rb)
  const int Rb = R / gridDim.y, row0 = blockIdx.y * Rb;
  const int parts = 256 / Rb;                // Rb in {32, 64, 128, 256}
  const int rl = tid % Rb, part = tid / Rb;
  const int r = row0 + rl;
  float gk[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) gk[c] = c < dk ? Gf[r * dk + c] : 0.f;
  float m = -INFINITY, l = 0.f, t[4] = {0.f, 0.f, 0.f, 0.f};
  f32x2 l2 = {0.f, 0.f}, t2[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
  int len = N;
  if (lengths != nullptr) len = lengths[b] < N ? lengths[b] : N;
  float xr[16];
  if (len > 0) fetch_points(X + (int64_t)b * N * dk, len < CH ? len : CH, dk, xr);
  for (int n0 = 0; n0 < len; n0 += CH) {
    const int cn = (len - n0 < CH) ? len - n0 : CH;
    // the chunk as [pair of points][component][2]: the loop below works on two points per
    // packed-fp32 instruction (v_pk_fma_f32 / v_pk_mul_f32 run two fp32 lanes per issue slot), whose
    // operands are the register PAIRS {x_a[c], x_b[c]} this layout delivers; the next chunk's loads
    // are in flight while this one is worked on
    const int npairs = (cn + 1) >> 1;
    commit_points(xr, cn, dk, sX);
    if (n0 + CH < len)
      fetch_points(X + ((int64_t)b * N + n0 + CH) * dk, (len - n0 - CH < CH) ? len - n0 - CH : CH, dk, xr);
    // online softmax over this thread's pairs of the chunk, four pairs at a time (one rescale per 8 points)
    for (int q = part; q < npairs; q += 4 * parts) {
      f32x2 sv[4], xv[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int pq = q + u * parts;
        const bool ok = pq < npairs;
        const float4 lo = ok ? *reinterpret_cast<const float4*>(&sX[pq * 8]) : float4{0.f, 0.f, 0.f, 0.f};
        const float4 hi = ok ? *reinterpret_cast<const float4*>(&sX[pq * 8 + 4]) : float4{0.f, 0.f, 0.f, 0.f};
        xv[u][0] = f32x2{lo.x, lo.y}; xv[u][1] = f32x2{lo.z, lo.w};
        xv[u][2] = f32x2{hi.x, hi.y}; xv[u][3] = f32x2{hi.z, hi.w};
        sv[u] = gk[0] * xv[u][0] + gk[1] * xv[u][1] + gk[2] * xv[u][2] + gk[3] * xv[u][3];
        if (!ok) sv[u] = f32x2{-INFINITY, -INFINITY};
        else if (2 * pq + 1 >= cn) sv[u][1] = -INFINITY;           // the odd point of the chunk
      }
      const float mn = fmaxf(fmaxf(fmaxf(m, fmaxf(sv[0][0], sv[0][1])), fmaxf(sv[1][0], sv[1][1])),
                             fmaxf(fmaxf(sv[2][0], sv[2][1]), fmaxf(sv[3][0], sv[3][1])));
      // (raw v_exp_f32: exp2f adds a range test and a rescale per call for results below 2^-126,
      //  which are zero for every purpose here)
      const float al = __builtin_amdgcn_exp2f(m - mn);
      l2 *= al;
#pragma unroll
      for (int c = 0; c < 4; ++c) t2[c] *= al;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const f32x2 p = {__builtin_amdgcn_exp2f(sv[u][0] - mn),          // exp2(-inf) = 0 for padding
                         __builtin_amdgcn_exp2f(sv[u][1] - mn)};
        l2 += p;
#pragma unroll
        for (int c = 0; c < 4; ++c) t2[c] += p * xv[u][c];
      }
      m = mn;
    }
  }
  l = l2[0] + l2[1];
#pragma unroll
  for (int c = 0; c < 4; ++c) t[c] = t2[c][0] + t2[c][1];
  if (part >= parts) { m = -INFINITY; l = 0.f; }
  sM[tid] = m; sL[tid] = l;
#pragma unroll
  for (int c = 0; c < 4; ++c) sT[tid][c] = t[c];
  __syncthreads();
  if (tid < Rb) {
    float M = -INFINITY;
    for (int p = 0; p < parts; ++p) M = fmaxf(M, sM[p * Rb + tid]);
    float L = 0.f, tt[4] = {0.f, 0.f, 0.f, 0.f};
    for (int p = 0; p < parts; ++p) {
      const float mw = sM[p * Rb + tid];
      if (mw == -INFINITY) continue;
      const float f = exp2f(mw - M);
      L += sL[p * Rb + tid] * f;
#pragma unroll
      for (int c = 0; c < 4; ++c) tt[c] += sT[p * Rb + tid][c] * f;
    }
    for (int c = 0; c < dk; ++c) T[((int64_t)b * R + row0 + tid) * dk + c] = tt[c] / L;
    LSE[(int64_t)b * R + row0 + tid] = M + log2f(L);
  }
}

// ---------------------------------------------------------------------------------
// per-set epilogue: O[q][32j+f] = Qp[q][.] + T[j*m+q][:] . Wv[32j+f][:] + bv ; Z = O Wo^T + bo
// H = O + relu(Z).  O and Z are saved for the backward.
// ---------------------------------------------------------------------------------
// thread = (output feature f, half of the queries); weights pre-transposed ([in][out]) so
// that consecutive threads read consecutive addresses; T / O rows broadcast from LDS.
template <int MQ>   // queries per thread: m/2 for ISAB (m = 16), 1 for PMA
__global__ __launch_bounds__(256) void k_mab0_epi(const float* __restrict__ Tp,   // [B][S][R][dk]
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
                                                  float* __restrict__ Zsave) {
  mab0_epi_body<MQ>(Tp, Mp, Lp, S, T, LSE, Qp, WvT, bv, WoT, bo, m, d, dk, h, H, Osave, Zsave, blockIdx.x);
}

}  // namespace

// ---- host side --------------------------------------------------------------------
bool mab0_bf16_supported(const pca_mab_shape& s) {
  if (s.d == 256) return mab0_d256_supported(s);
  const int R = s.h * s.nq;
  if (!(s.nq == 16 || s.nq <= 2)) return false;      // epilogue: 8 or 1 queries per thread
  // the keys X may be bf16 when they are a hidden tensor (dk == d)
  const bool dt_ok = s.q_dtype == PCA_F32 && s.y_dtype == PCA_F32 &&
                     (s.k_dtype == PCA_F32 || (s.k_dtype == PCA_BF16 && s.dk == s.d));
  return s.q_shared == 1 && s.d == 128 && s.h * 32 == s.d && s.dq == s.d &&
         ((s.dk == s.d && (R <= 16 || R == 64)) ||
          (s.dk <= 4 && (R == 64 || R == 128 || R == 256))) && dt_ok;
}

void mab0_collect_prep(const pca_mab_shape& s, const float* I, const pca_mab_params& p,
                       const Mab0Saved& v, bool training, bool epilogue_images,
                       Mab0PrepJobs* J) {
  Mab0PrepJob a{};
  a.I = I; a.Wq = p.wq; a.bq = p.bq; a.Wk = p.wk;
  a.m = s.nq; a.d = s.d; a.dq = s.dq; a.dk = s.dk; a.h = s.h;
  a.Rp = (int)cdiv(s.h * s.nq, 32) * 32;
  a.sl2e = 1.4426950408889634f / sqrtf((float)s.d);
  a.Qp = v.Qp; a.Gf = v.Gf;
  a.Gb = s.dk <= 4 ? nullptr : v.Gb;
  a.GtP = (s.dk <= 4 || !training) ? nullptr : v.GtP;
  if (epilogue_images) { a.Wv = p.wv; a.Wo = p.wo; a.WvT = v.WvT; a.WoT = v.WoT; }
  J->j[J->n++] = a;
}
int prep_all_launch(const PrepJobs& W, const Mab0PrepJobs& Q, hipStream_t st) {
  if (W.n == 0) return mab0_prep_launch(Q, st);
  if (Q.n == 0) return prep_jobs_launch(W, st);
  int maxm = 0, maxd = 0, maxe = 0;
  bool spare = false;
  for (int i = 0; i < Q.n; ++i) {
    maxm = Q.j[i].m > maxm ? Q.j[i].m : maxm;
    maxd = Q.j[i].d > maxd ? Q.j[i].d : maxd;
    spare = spare || Q.j[i].WvT != nullptr;
  }
  for (int i = 0; i < W.n; ++i) {
    const int e = W.j[i].rows * W.j[i].cols;
    maxe = e > maxe ? e : maxe;
  }
  for (int i = 0; i < Q.n; ++i)              // d > 128: one workgroup per (query row, head)
    if (Q.j[i].d > 128) maxm = Q.j[i].m * Q.j[i].h > maxm ? Q.j[i].m * Q.j[i].h : maxm;
  int gx = maxm + (spare ? PREP_SPARE : 0);
  const int gw = (int)cdiv(maxe, 256);
  gx = gx > gw ? gx : gw;
  PCA_REQUIRE(pack_stream_ok(st), "deferred pack: submitted on another stream than the engine call's");
  PackJob P{};
  const int prows = pack_take(&P) ? (int)cdiv(pack_blocks(P), gx) : 0;
  hipLaunchKernelGGL(k_prep_all, dim3(gx, Q.n + W.n + prows), dim3(256), maxd * sizeof(float), st,
                     W, Q, P);
  return check_launch("k_prep_all");
}

int mab0_prep_launch(const Mab0PrepJobs& J, hipStream_t st) {
  if (J.n == 0) return PCA_OK;
  int maxm = 0, maxd = 0;
  for (int i = 0; i < J.n; ++i) {
    maxm = J.j[i].m > maxm ? J.j[i].m : maxm;
    maxd = J.j[i].d > maxd ? J.j[i].d : maxd;
  }
  bool spare = false;
  for (int i = 0; i < J.n; ++i) spare = spare || J.j[i].WvT != nullptr;
  for (int i = 0; i < J.n; ++i)              // d > 128: one workgroup per (query row, head)
    if (J.j[i].d > 128) maxm = J.j[i].m * J.j[i].h > maxm ? J.j[i].m * J.j[i].h : maxm;
  hipLaunchKernelGGL(k_mab0_prep, dim3(maxm + (spare ? PREP_SPARE : 0), J.n), dim3(256),
                     maxd * sizeof(float), st, J);
  return check_launch("k_mab0_prep");
}

// layer-1 attention (dk <= 4) of R = h*m score rows, stand-alone (the d = 256 path calls it)
// workgroups per set of the layer-1 kernels (each owns R / y >= 32 score rows and re-stages the
// set's few-column points, which costs nothing): the work is vector-ALU arithmetic per (row,
// point), so the launch should put four waves on every SIMD, not one
int small_row_split(int B, int R) {
  int y = 2;
  while (y * 2 <= R / 32 && (int64_t)B * y < 1024) y *= 2;
  return y;
}

int mab0_attn_small_launch(const float* X, const float* Gf, int B, int N, int R, int dk, float* T,
                           float* LSE, const int32_t* lengths, hipStream_t st) {
  PCA_REQUIRE(R == 64 || R == 128 || R == 256, "mab0_attn_small: %d score rows", R);
  hipLaunchKernelGGL(k_mab0_attn_small, dim3(B, small_row_split(B, R)), dim3(256), 0, st, X, Gf, N,
                     R, dk, T, LSE, lengths);
  return check_launch("k_mab0_attn_small");
}

// point-range splits per set so that B*S workgroups cover the 256 CUs
int mab0_splits(const pca_mab_shape& s) {
  if (s.dk <= 4) return 1;
  int S = 1;
  const int tiles = (int)cdiv(s.nk, 128);
  // PMA (a handful of score rows): the per-range partials are tiny and the forward needs
  // 40 KiB of LDS, so aim for two workgroups per CU; ISAB: one per CU
  const int want = s.h * s.nq <= 16 ? 512 : 256;
  while (S * 2 <= tiles && s.B * S < want && S < 8) S *= 2;
  return S;
}

size_t mab0_carve_saved(const pca_mab_shape& s, Mab0Saved* out, void* base) {
  Carver c(base);
  Mab0Saved v;
  const int R = s.h * s.nq;
  const int Rpad = (int)cdiv(R, 32) * 32;     // the backward contracts 32 rows per MFMA
  v.Qp = c.take<float>((size_t)s.nq * s.d);
  v.Gf = c.take<float>((size_t)Rpad * s.dk);
  v.Gb = c.take<__bf16>((size_t)Rpad * s.dk);
  v.GtP = c.take<__bf16>((size_t)Rpad * s.dk);
  v.T = c.take<float>((size_t)s.B * R * s.dk);
  v.LSE = c.take<float>((size_t)s.B * R);
  v.O = c.take<float>((size_t)s.B * s.nq * s.d);
  v.Z = c.take<float>((size_t)s.B * s.nq * s.d);
  v.WvT = c.take<float>((size_t)s.dk * s.d);
  v.WoT = c.take<float>((size_t)s.d * s.d);
  const int S = mab0_splits(s);
  v.Tp = c.take<float>((size_t)s.B * S * R * s.dk);
  v.Mp = c.take<float>((size_t)s.B * S * R);
  v.Lp = c.take<float>((size_t)s.B * S * R);
  if (out) *out = v;
  return c.off;
}

size_t mab0_bf16_saved_bytes(const pca_mab_shape& s) {
  if (s.d == 256) return mab0_d256_saved_bytes(s);
  return mab0_carve_saved(s, nullptr, nullptr);
}
size_t mab0_bf16_fwd_ws_bytes(const pca_mab_shape& s) {
  if (s.d == 256) return mab0_d256_fwd_ws_bytes(s);
  return mab0_carve_saved(s, nullptr, nullptr);
}

int mab0_bf16_fwd(const pca_mab_shape& s, const float* I, const void* X,
                  const pca_mab_params& p, float* H, void* saved, void* ws, hipStream_t st) {
  return mab0_bf16_fwd_ex(s, I, X, p, H, saved, ws, 0, st);
}
int mab0_bf16_fwd_ex(const pca_mab_shape& s, const float* I, const void* X,
                     const pca_mab_params& p, float* H, void* saved, void* ws, int flags,
                     hipStream_t st) {
  if (s.d == 256) return mab0_d256_fwd(s, I, X, p, H, saved, ws, st);
  PCA_REQUIRE(mab0_bf16_supported(s), "mab0_bf16_fwd: unsupported shape");
  const bool training = saved != nullptr;
  PCA_REQUIRE(training || ws != nullptr, "mab0_bf16_fwd: scratch required");
  Mab0Saved v;
  mab0_carve_saved(s, &v, training ? saved : ws);
  const int d = s.d, m = s.nq, h = s.h, dk = s.dk, R = h * m;
  const int Rpad = (int)cdiv(R, 16) * 16;
  const int Rpad32 = (int)cdiv(R, 32) * 32;
  const bool small = dk <= 4;

  if (!(flags & PCA_F_PREP_DONE)) {
    Mab0PrepJobs J{};
    mab0_collect_prep(s, I, p, v, training, !(flags & PCA_F_SKIP_EPILOGUE), &J);
    PCA_TRY(mab0_prep_launch(J, st));
  }
  (void)Rpad32;

  const double pts = (double)s.B * s.nk;
  if (small) {
    // two workgroups per set when that still leaves 32 rows each
    hipLaunchKernelGGL(k_mab0_attn_small, dim3(s.B, (R % 64 == 0 && R <= 512) ? 2 : 1), dim3(256), 0, st,
                       reinterpret_cast<const float*>(X), v.Gf, s.nk, R, dk, v.T, v.LSE,
                       s.k_lengths);
    PCA_TRY(check_launch("k_mab0_attn_small"));
  } else {
    const int S = mab0_splits(s);
    Mab0AttnArgs a{X, v.Gb, v.Tp, v.Mp, v.Lp, s.B, s.nk, R, S, s.k_lengths};
    const int RB = Rpad / 16;
    const size_t xbytes = (size_t)4 * Rpad * 128 * 4 > 4 * 32 * 256 ? (size_t)4 * Rpad * 128 * 4
                                                                     : (size_t)4 * 32 * 256;
    const size_t lds = (size_t)Rpad * 256 + xbytes + 3 * 4 * Rpad * sizeof(float);
    // reference-formulation FLOPs of the block: fc_k, fc_v over the keys + QK^T + AV
    ProfScope ps(PCA_K_MAB0_FWD, st, 2.0 * pts * (2.0 * dk * d + 2.0 * m * d),
                 pts * 4.0 * dk);
    const dim3 grid(s.B, S);
    const bool abf = s.k_dtype == PCA_BF16;
    PCA_REQUIRE(RB == 1 || RB == 4, "mab0_bf16_fwd: %d score rows not built", Rpad);
    const size_t lds_h4 = 64 * 256 + 2 * 2 * 32 * 256 + 4 * 16 * sizeof(float);
    if (RB == 1 && abf) hipLaunchKernelGGL((k_mab0_attn<1, true>), grid, dim3(256), lds, st, a);
    else if (RB == 1) hipLaunchKernelGGL((k_mab0_attn<1, false>), grid, dim3(256), lds, st, a);
    else if (abf) hipLaunchKernelGGL((k_mab0_attn_h4<true>), grid, dim3(256), lds_h4, st, a);
    else hipLaunchKernelGGL((k_mab0_attn_h4<false>), grid, dim3(256), lds_h4, st, a);
    ps.end();
    PCA_TRY(check_launch("k_mab0_attn"));
  }
  if (flags & PCA_F_SKIP_EPILOGUE) return PCA_OK;
  // WvT / WoT were written by the spare workgroups of k_mab0_prep
  const size_t el = ((size_t)R * dk + (size_t)m * d) * sizeof(float);
  float* Os = training ? v.O : nullptr;
  float* Zs = training ? v.Z : nullptr;
  const int Sm = small ? 0 : mab0_splits(s);
  if (m > 2)
    hipLaunchKernelGGL((k_mab0_epi<8>), dim3(s.B), dim3(256), el, st, v.Tp, v.Mp, v.Lp, Sm, v.T,
                       v.LSE, v.Qp, v.WvT, p.bv, v.WoT, p.bo, m, d, dk, h, H, Os, Zs);
  else
    hipLaunchKernelGGL((k_mab0_epi<1>), dim3(s.B), dim3(256), el, st, v.Tp, v.Mp, v.Lp, Sm, v.T,
                       v.LSE, v.Qp, v.WvT, p.bv, v.WoT, p.bo, m, d, dk, h, H, Os, Zs);
  return check_launch("k_mab0_epi");
}

}  // namespace pca
