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
  float* part;                 // S > 1: per-split partial results (layouts at the kernels)
  const int32_t* lengths;
  int B, nq, nk, d, dh, h;
  int nt, gt, S, cps;          // Plan of the launch
  int64_t qb;                  // batch stride of Qp (0: shared)
  float scale, c;              // 1 / sqrt(d), scale * log2(e)
};

constexpr int CH = 64;         // streamed rows per chunk
// bf16 elements of one staged chunk: row form has one spare row (the last head's upper lanes read on
// into the next row), transposed form 16 spare feature rows
__host__ __device__ constexpr int row_elems(int d) { return (CH + 1) * (d + 4); }
__host__ __device__ constexpr int tr_elems(int d) { return (d + 16) * (CH + 4); }

// How a launch is cut.  The side that stays in registers (queries; keys in k_attnc_bwd_kv) has nt tiles of
// 16 rows; gt workgroups per set share them (workgroup xt takes tiles xt, xt + gt, ...).  Where that alone
// leaves the chip short of workgroups - few tiles and few sets: nq = 64 or 1 at B = 16 - the streamed
// side is cut as well, into S ranges of cps chunks whose partial results a small second kernel merges.
struct Plan { int nt, gt, S, cps; };
inline Plan plan_of(int B, int reg_rows, int str_rows) {
  Plan p;
  p.nt = (int)cdiv(reg_rows, 16);
  const int nch = (int)cdiv(str_rows, CH);
  const int want = (int)cdiv(512, B);                 // workgroups per set for two per CU
  p.gt = p.nt < want ? p.nt : want;
  int S = 1;
  if (2 * p.gt <= want && nch >= 4) {
    S = want / p.gt;
    if (S > nch / 2) S = nch / 2;                     // >= 2 chunks per range
  }
  p.cps = (int)cdiv(nch, S);
  p.S = (int)cdiv(nch, p.cps);                        // no empty range
  return p;
}

// workgroup id -> set b, tile slot xt, range sp: sets b and b + 8 k sit on the same XCD (id mod 8), the
// workgroups of one set in consecutive slots of it
__device__ __forceinline__ void wg_coords(const AttnCoreArgs& a, int& b, int& xt, int& sp) {
  const int id = blockIdx.x, slot = id >> 3, gx = a.gt * a.S;
  const int x = slot % gx;
  b = (slot / gx) * 8 + (id & 7);
  xt = x % a.gt;
  sp = x / a.gt;
}

// Staging of a chunk, all heads at once: thread -> rows 4 q .. 4 q + 3, features 4 c4 .. 4 c4 + 3 of
// X[nrows][d] (CH / 4 x d / 4 items <= the workgroup's 64 h threads because d / h <= 16); rows >= nrows
// are zeros.  Synchronous - the other workgroups of the CU (four fit) cover the loads: holding the next
// chunk in registers under the products was tried and costs more than it hides (32 VGPRs per operand
// take the kernels to the 128-register cap, two workgroups per CU, and hipcc copies parts of the loaded
// quads right behind the loads, i.e. waits for them where they were issued).
struct StageId {
  int q, c4;
  unsigned off;                                        // (4 q) d + 4 c4: the item's first float in a chunk
  bool act;
};
__device__ __forceinline__ StageId stage_id(int d, int tid) {
  const int cg = d >> 2, total = (CH / 4) * cg;
  const int i = tid < total ? tid : total - 1;
  StageId s;
  s.q = i / cg; s.c4 = i - s.q * cg; s.act = tid < total;
  s.off = (unsigned)(4 * s.q * d + 4 * s.c4);
  return s;
}
struct Staged { float4 v[4]; };
// X: the set's [nrows][d] array (a uniform pointer: the loads take it as their scalar base)
__device__ __forceinline__ Staged stage_load(const float* __restrict__ X, int nrows, int r0, int d,
                                             const StageId& id) {
  Staged s;
  if (id.act) {
    if (r0 + CH <= nrows) {                            // (uniform) the whole chunk exists
      const float* p = X + (size_t)r0 * d;
#pragma unroll
      for (int e = 0; e < 4; ++e) s.v[e] = *reinterpret_cast<const float4*>(p + (id.off + e * d));
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = r0 + 4 * id.q + e;
        const int rc = row < nrows ? row : nrows - 1;
        s.v[e] = *reinterpret_cast<const float4*>(X + (size_t)rc * d + 4 * id.c4);
      }
    }
  }
  return s;                                            // (rows past the end are zeroed at stage_store: a
}                                                      //  select here would wait for the loads at once)
template <bool ROW, bool TR>
__device__ __forceinline__ void stage_store(const Staged& s, int nrows, int r0, int d, const StageId& id,
                                            __bf16* __restrict__ Rw, __bf16* __restrict__ Tr) {
  if (!id.act) return;
  f32x4 v[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = f32x4{s.v[e].x, s.v[e].y, s.v[e].z, s.v[e].w};
  if (r0 + CH > nrows) {                               // (uniform) the ragged last chunk
    const int live = nrows - (r0 + 4 * id.q);
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (e >= live) v[e] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  if constexpr (ROW) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      *reinterpret_cast<bf16x4*>(Rw + (4 * id.q + e) * (d + 4) + 4 * id.c4) =
          __builtin_convertvector(v[e], bf16x4);
  }
  if constexpr (TR) {
    __bf16* dst = Tr + (4 * id.c4) * (CH + 4) + 4 * id.q;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 t = {v[0][i], v[1][i], v[2][i], v[3][i]};
      *reinterpret_cast<bf16x4*>(dst + i * (CH + 4)) = __builtin_convertvector(t, bf16x4);
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

// S > 1: part = { ml [B][S][h][nq][2] , acc [B][S][nq][d] } (unnormalised), merged by k_attnc_fwd_merge
template <int NW>   // waves (= heads) the launch may have
__global__ __launch_bounds__(64 * NW, 4) void k_attnc_fwd(const AttnCoreArgs a) {
  extern __shared__ __attribute__((aligned(16))) __bf16 lds[];
  const int tid = threadIdx.x, lane = tid & 63, j = tid >> 6, r = lane & 15, g = lane >> 4;
  const int nthr = blockDim.x, d = a.d, dh = a.dh, nq = a.nq, nk = a.nk;
  int b, xt, sp;
  wg_coords(a, b, xt, sp);
  if (b >= a.B) return;
  __bf16* Krow = lds;
  __bf16* Vtr = lds + row_elems(d);
  lds_clear(lds, row_elems(d) + tr_elems(d), tid, nthr);
  int len = nk;
  if (a.lengths != nullptr) len = a.lengths[b] < nk ? a.lengths[b] : nk;
  const int nlive = (len + CH - 1) / CH;               // chunks that hold live keys
  const int cb = sp * a.cps, ce = cb + a.cps < nlive ? cb + a.cps : nlive;
  const bool single = ce - cb == 1;                    // one chunk: staged once for all the tiles
  const StageId id = stage_id(d, tid);
  const float* Kb = a.Kp + (int64_t)b * nk * d;
  const float* Vb = a.Vp + (int64_t)b * nk * d;
  bool staged = false;
  // the tile's query rows: 4 features per lane (fo = the lane's piece, clamped into the head), loaded
  // a tile ahead - a tile of the many-queries blocks is four key tiles of arithmetic
  const int fo = 4 * g < dh ? 4 * g : 0;
  const float* Qb = a.Qp + (int64_t)b * a.qb + j * dh + fo;
  auto q_of = [&](int t) {
    const int qi = t * 16 + r < nq ? t * 16 + r : nq - 1;
    return *reinterpret_cast<const float4*>(Qb + (int64_t)qi * d);
  };
  float4 qv = q_of(xt < a.nt ? xt : 0);
  for (int t = xt; t < a.nt; t += a.gt) {
    const int q0 = t * 16;
    const float4 q4 = qv;
    qv = q_of(t + a.gt < a.nt ? t + a.gt : t);
    bf16x4 qb4;
    qb4[0] = (__bf16)(4 * g < dh ? q4.x : 0.f); qb4[1] = (__bf16)(4 * g < dh ? q4.y : 0.f);
    qb4[2] = (__bf16)(4 * g < dh ? q4.z : 0.f); qb4[3] = (__bf16)(4 * g < dh ? q4.w : 0.f);
    float m = -INFINITY, l = 0.f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const bool fill = !(single && staged);
    for (int c = cb; c < ce; ++c) {
      if (fill) {
        __syncthreads();                               // the previous chunk is consumed (first: cleared)
        const Staged kr = stage_load(Kb, nk, c * CH, d, id);
        const Staged vr = stage_load(Vb, nk, c * CH, d, id);
        stage_store<true, false>(kr, nk, c * CH, d, id, Krow, nullptr);
        stage_store<false, true>(vr, nk, c * CH, d, id, nullptr, Vtr);
        __syncthreads();
      }
      bf16x4 ka[4], va[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        ka[u] = *reinterpret_cast<const bf16x4*>(Krow + (16 * u + r) * (d + 4) + j * dh + 4 * g);
        va[u] = *reinterpret_cast<const bf16x4*>(Vtr + (j * dh + r) * (CH + 4) + 16 * u + 4 * g);
      }
      if ((c + 1) * CH <= len) fwd_keys64<false>(ka, va, qb4, a.c, 64, g, m, l, acc);
      else                     fwd_keys64<true>(ka, va, qb4, a.c, len - c * CH, g, m, l, acc);
    }
    staged = true;
    const float lt = wave16_sum(l);
    if (q0 + r < nq) {
      if (a.S == 1) {
        if (4 * g < dh) {
          const float inv = 1.f / lt;
          *reinterpret_cast<float4*>(a.Oout + ((int64_t)b * nq + q0 + r) * d + j * dh + 4 * g) =
              float4{q4.x + acc[0] * inv, q4.y + acc[1] * inv, q4.z + acc[2] * inv, q4.w + acc[3] * inv};
        }
        if (g == 0) a.LSE[((int64_t)b * a.h + j) * nq + q0 + r] = m + log2f(lt);
      } else {
        float* ml = a.part;
        float* pa = a.part + (int64_t)a.B * a.S * a.h * nq * 2;
        const int64_t bs = (int64_t)b * a.S + sp;
        if (4 * g < dh)
          *reinterpret_cast<float4*>(pa + (bs * nq + q0 + r) * d + j * dh + 4 * g) =
              float4{acc[0], acc[1], acc[2], acc[3]};
        if (g == 0)
          *reinterpret_cast<float2*>(ml + ((bs * a.h + j) * nq + q0 + r) * 2) = float2{m, lt};
      }
    }
  }
}

// one thread per (set, query, head, 4 features): O = Q_ + sum_s 2^(m_s - M) acc_s / L, LSE = M + log2 L
__global__ __launch_bounds__(256) void k_attnc_fwd_merge(const AttnCoreArgs a) {
  const int f4 = a.dh >> 2;
  const int64_t n = (int64_t)a.B * a.nq * a.h * f4;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int g = (int)(i % f4), j = (int)((i / f4) % a.h);
  const int q = (int)((i / ((int64_t)f4 * a.h)) % a.nq), b = (int)(i / ((int64_t)f4 * a.h * a.nq));
  const float* ml = a.part;
  const float* pa = a.part + (int64_t)a.B * a.S * a.h * a.nq * 2;
  float M = -INFINITY;
  for (int s = 0; s < a.S; ++s)
    M = fmaxf(M, ml[((((int64_t)b * a.S + s) * a.h + j) * a.nq + q) * 2]);
  float L = 0.f;
  float4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < a.S; ++s) {
    const int64_t bs = (int64_t)b * a.S + s;
    const float2 v = *reinterpret_cast<const float2*>(ml + ((bs * a.h + j) * a.nq + q) * 2);
    const float w = __builtin_amdgcn_exp2f(v.x - M);   // a range without live keys: 2^(-inf) = 0
    const float4 p = *reinterpret_cast<const float4*>(pa + (bs * a.nq + q) * a.d + j * a.dh + 4 * g);
    L = fmaf(v.y, w, L);
    acc.x = fmaf(p.x, w, acc.x); acc.y = fmaf(p.y, w, acc.y);
    acc.z = fmaf(p.z, w, acc.z); acc.w = fmaf(p.w, w, acc.w);
  }
  const float4 q4 = *reinterpret_cast<const float4*>(a.Qp + (int64_t)b * a.qb + (int64_t)q * a.d +
                                                     j * a.dh + 4 * g);
  const float inv = 1.f / L;
  *reinterpret_cast<float4*>(a.Oout + ((int64_t)b * a.nq + q) * a.d + j * a.dh + 4 * g) =
      float4{q4.x + acc.x * inv, q4.y + acc.y * inv, q4.z + acc.z * inv, q4.w + acc.w * inv};
  if (g == 0) a.LSE[((int64_t)b * a.h + j) * a.nq + q] = M + log2f(L);
}

// dst[b][row][:] = (res ? res[b][row][:] : 0) + sum_s part[b][s][row][:]      (rows x d floats per set)
__global__ __launch_bounds__(256) void k_attnc_sum(float* __restrict__ dst, const float* __restrict__ res,
                                                   const float* __restrict__ part, int B, int S,
                                                   int64_t per_set4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)B * per_set4) return;
  const int64_t b = i / per_set4, o = i - b * per_set4;
  float4 v = res != nullptr ? reinterpret_cast<const float4*>(res)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  for (int s = 0; s < S; ++s) {
    const float4 p = reinterpret_cast<const float4*>(part)[(b * S + s) * per_set4 + o];
    v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
  }
  reinterpret_cast<float4*>(dst)[i] = v;
}

// S > 1: part = dQ partials [B][S][nq][d] (without the residual), summed onto dO by k_attnc_sum
template <int NW>   // waves (= heads) the launch may have
__global__ __launch_bounds__(64 * NW, 4) void k_attnc_bwd_q(const AttnCoreArgs a) {
  extern __shared__ __attribute__((aligned(16))) __bf16 lds[];
  const int tid = threadIdx.x, lane = tid & 63, j = tid >> 6, r = lane & 15, g = lane >> 4;
  const int nthr = blockDim.x, d = a.d, dh = a.dh, nq = a.nq, nk = a.nk;
  int b, xt, sp;
  wg_coords(a, b, xt, sp);
  if (b >= a.B) return;
  __bf16* Krow = lds;
  __bf16* Vrow = Krow + row_elems(d);
  __bf16* Ktr = Vrow + row_elems(d);
  lds_clear(lds, 2 * row_elems(d) + tr_elems(d), tid, nthr);
  int len = nk;
  if (a.lengths != nullptr) len = a.lengths[b] < nk ? a.lengths[b] : nk;
  const int nlive = (len + CH - 1) / CH;
  const int cb = sp * a.cps, ce = cb + a.cps < nlive ? cb + a.cps : nlive;
  const bool single = ce - cb == 1;
  const StageId id = stage_id(d, tid);
  const float* Kb = a.Kp + (int64_t)b * nk * d;
  const float* Vb = a.Vp + (int64_t)b * nk * d;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  bool staged = false;
  // the tile's rows of Q_, dO, O (the lane's 4 features) and its LSE, loaded a tile ahead (see k_attnc_fwd)
  const int fo = 4 * g < dh ? 4 * g : 0;
  const bool on = 4 * g < dh;
  struct TileIn { float4 q, dO, o; float lse; };
  auto tile_in = [&](int t) {
    const int qi = t * 16 + r < nq ? t * 16 + r : nq - 1;
    const int64_t orow = ((int64_t)b * nq + qi) * d + j * dh + fo;
    TileIn x;
    x.q = *reinterpret_cast<const float4*>(a.Qp + (int64_t)b * a.qb + (int64_t)qi * d + j * dh + fo);
    x.dO = *reinterpret_cast<const float4*>(a.dO + orow);
    x.o = *reinterpret_cast<const float4*>(a.O + orow);
    x.lse = a.LSE[((int64_t)b * a.h + j) * nq + qi];
    return x;
  };
  TileIn nx = tile_in(xt < a.nt ? xt : 0);
  for (int t = xt; t < a.nt; t += a.gt) {
    const int q0 = t * 16;
    const TileIn cur = nx;
    nx = tile_in(t + a.gt < a.nt ? t + a.gt : t);
    const float4 do4 = cur.dO, o4 = cur.o, q4 = cur.q;
    bf16x4 qb4, dob;
    qb4[0] = (__bf16)(on ? q4.x : 0.f); qb4[1] = (__bf16)(on ? q4.y : 0.f);
    qb4[2] = (__bf16)(on ? q4.z : 0.f); qb4[3] = (__bf16)(on ? q4.w : 0.f);
    dob[0] = (__bf16)(on ? do4.x : 0.f); dob[1] = (__bf16)(on ? do4.y : 0.f);
    dob[2] = (__bf16)(on ? do4.z : 0.f); dob[3] = (__bf16)(on ? do4.w : 0.f);
    const float nlse = -cur.lse;
    float dl = do4.x * (o4.x - q4.x) + do4.y * (o4.y - q4.y) + do4.z * (o4.z - q4.z) +
               do4.w * (o4.w - q4.w);
    if (!on) dl = 0.f;
    const float delta = wave16_sum(dl);                // rowdot(dO_j, A V_j) of query r
    const float nds = -delta * a.scale;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const bool fill = !(single && staged);
    for (int c = cb; c < ce; ++c) {
      if (fill) {
        __syncthreads();
        const Staged kr = stage_load(Kb, nk, c * CH, d, id);
        const Staged vr = stage_load(Vb, nk, c * CH, d, id);
        stage_store<true, true>(kr, nk, c * CH, d, id, Krow, Ktr);
        stage_store<true, false>(vr, nk, c * CH, d, id, Vrow, nullptr);
        __syncthreads();
      }
      const int klim = len - c * CH;                   // live keys of the chunk (may exceed 64)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (16 * u >= klim) break;
        const int ro = (16 * u + r) * (d + 4) + j * dh + 4 * g;
        const bf16x4 ka = *reinterpret_cast<const bf16x4*>(Krow + ro);
        const bf16x4 vv = *reinterpret_cast<const bf16x4*>(Vrow + ro);
        const bf16x4 kt = *reinterpret_cast<const bf16x4*>(Ktr + (j * dh + r) * (CH + 4) + 16 * u + 4 * g);
        const f32x4 s = mfma16(ka, qb4, z4);            // [key 4 g + e][query r]
        const f32x4 dp = mfma16(vv, dob, z4);           // dP^T = V dO^T
        f32x4 ds;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float pe = __builtin_amdgcn_exp2f(fmaf(s[e], a.c, nlse));
          if (16 * u + 16 > klim) pe = 16 * u + 4 * g + e < klim ? pe : 0.f;
          ds[e] = pe * fmaf(dp[e], a.scale, nds);
        }
        acc = mfma16(kt, pack4(ds), acc);               // dQ^T += K^T dS^T
      }
    }
    staged = true;
    if (q0 + r < nq) {
      if (4 * g < dh) {
        if (a.S == 1)
          *reinterpret_cast<float4*>(a.dQp + ((int64_t)b * nq + q0 + r) * d + j * dh + 4 * g) =
              float4{do4.x + acc[0], do4.y + acc[1], do4.z + acc[2], do4.w + acc[3]};  // + the residual Q_
        else
          *reinterpret_cast<float4*>(a.part + (((int64_t)b * a.S + sp) * nq + q0 + r) * d + j * dh +
                                     4 * g) = float4{acc[0], acc[1], acc[2], acc[3]};
      }
      if (g == 0 && sp == 0) a.Delta[((int64_t)b * a.h + j) * nq + q0 + r] = delta;
    }
  }
}

// S > 1: part = { dK partials [B][S][nk][d], dV partials [B][S][nk][d] }, summed by k_attnc_sum
template <int NW>   // waves (= heads) the launch may have
__global__ __launch_bounds__(64 * NW, 4) void k_attnc_bwd_kv(const AttnCoreArgs a) {
  extern __shared__ __attribute__((aligned(16))) __bf16 lds[];
  const int tid = threadIdx.x, lane = tid & 63, j = tid >> 6, r = lane & 15, g = lane >> 4;
  const int nthr = blockDim.x, d = a.d, dh = a.dh, nq = a.nq, nk = a.nk, h = a.h;
  int b, xt, sp;
  wg_coords(a, b, xt, sp);
  if (b >= a.B) return;
  __bf16* Qrow = lds;
  __bf16* Drow = Qrow + row_elems(d);
  __bf16* Qtr = Drow + row_elems(d);
  __bf16* Dtr = Qtr + tr_elems(d);
  const int nb16 = 2 * row_elems(d) + 2 * tr_elems(d);
  float* lseS = reinterpret_cast<float*>(lds + ((nb16 + 7) & ~7));   // [h][CH] each; one entry per thread
  float* delS = lseS + h * CH;
  lds_clear(lds, nb16, tid, nthr);
  int len = nk;
  if (a.lengths != nullptr) len = a.lengths[b] < nk ? a.lengths[b] : nk;
  const int nch = (nq + CH - 1) / CH;
  const int cb = sp * a.cps, ce = cb + a.cps < nch ? cb + a.cps : nch;
  const bool single = ce - cb == 1;
  const StageId id = stage_id(d, tid);
  const float* Qb = a.Qp + (int64_t)b * a.qb;
  const float* dOb = a.dO + (int64_t)b * nq * d;
  const float* lseb = a.LSE + ((int64_t)b * h + j) * nq;
  const float* delb = a.Delta + ((int64_t)b * h + j) * nq;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  bool staged = false;
  // the tile's rows of K_ and V_ (the lane's 4 features), loaded a tile ahead (see k_attnc_fwd)
  const int fo = 4 * g < dh ? 4 * g : 0;
  const bool on = 4 * g < dh;
  struct TileIn { float4 k, v; };
  auto tile_in = [&](int t) {
    const int ki = t * 16 + r < nk ? t * 16 + r : nk - 1;
    const int64_t o = ((int64_t)b * nk + ki) * d + j * dh + fo;
    TileIn x;
    x.k = *reinterpret_cast<const float4*>(a.Kp + o);
    x.v = *reinterpret_cast<const float4*>(a.Vp + o);
    return x;
  };
  TileIn nx = tile_in(xt < a.nt ? xt : 0);
  for (int t = xt; t < a.nt; t += a.gt) {
    const int k0 = t * 16;
    const TileIn cur = nx;
    nx = tile_in(t + a.gt < a.nt ? t + a.gt : t);
    bf16x4 kb4, vb4;                                   // B operands [k = feature][col = key r]
    kb4[0] = (__bf16)(on ? cur.k.x : 0.f); kb4[1] = (__bf16)(on ? cur.k.y : 0.f);
    kb4[2] = (__bf16)(on ? cur.k.z : 0.f); kb4[3] = (__bf16)(on ? cur.k.w : 0.f);
    vb4[0] = (__bf16)(on ? cur.v.x : 0.f); vb4[1] = (__bf16)(on ? cur.v.y : 0.f);
    vb4[2] = (__bf16)(on ? cur.v.z : 0.f); vb4[3] = (__bf16)(on ? cur.v.w : 0.f);
    const bool ragged = k0 + 16 > len, dead = k0 + r >= len;
    f32x4 dk = {0.f, 0.f, 0.f, 0.f}, dv = {0.f, 0.f, 0.f, 0.f};
    const bool fill = !(single && staged);
    for (int c = cb; c < ce; ++c) {
      if (fill) {
        __syncthreads();
        const Staged qr = stage_load(Qb, nq, c * CH, d, id);
        const Staged dr = stage_load(dOb, nq, c * CH, d, id);
        const int qq = c * CH + lane;                  // this thread's entry: head j (its wave), query qq
        const int qc = qq < nq ? qq : nq - 1;
        const float lr = lseb[qc], er = delb[qc];
        stage_store<true, true>(qr, nq, c * CH, d, id, Qrow, Qtr);
        stage_store<true, true>(dr, nq, c * CH, d, id, Drow, Dtr);
        lseS[j * CH + lane] = qq < nq ? -lr : -INFINITY;      // queries past the end: P = 2^(-inf) = 0
        delS[j * CH + lane] = qq < nq ? -er * a.scale : 0.f;
        __syncthreads();
      }
#pragma unroll
      for (int u = 0; u < CH / 16; ++u) {
        if (c * CH + 16 * u >= nq) break;
        const int ro = (16 * u + r) * (d + 4) + j * dh + 4 * g;
        const int to = (j * dh + r) * (CH + 4) + 16 * u + 4 * g;
        const bf16x4 qa = *reinterpret_cast<const bf16x4*>(Qrow + ro);    // A operand [query r][k = feature]
        const bf16x4 doa = *reinterpret_cast<const bf16x4*>(Drow + ro);
        const bf16x4 qt = *reinterpret_cast<const bf16x4*>(Qtr + to);     // Q^T, dO^T: [feature][k = query]
        const bf16x4 dot = *reinterpret_cast<const bf16x4*>(Dtr + to);
        const float4 nl = *reinterpret_cast<const float4*>(lseS + j * CH + 16 * u + 4 * g);
        const float4 nd = *reinterpret_cast<const float4*>(delS + j * CH + 16 * u + 4 * g);
        const float nl4[4] = {nl.x, nl.y, nl.z, nl.w}, nd4[4] = {nd.x, nd.y, nd.z, nd.w};
        const f32x4 s = mfma16(qa, kb4, z4);           // [query 4 g + e][key r]
        const f32x4 dp = mfma16(doa, vb4, z4);         // dP = dO V^T
        f32x4 p, ds;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          p[e] = __builtin_amdgcn_exp2f(fmaf(s[e], a.c, nl4[e]));
          if (ragged) p[e] = dead ? 0.f : p[e];
          ds[e] = p[e] * fmaf(dp[e], a.scale, nd4[e]);
        }
        dv = mfma16(dot, pack4(p), dv);                // dV^T += dO^T P     [feature 4 g + e][key r]
        dk = mfma16(qt, pack4(ds), dk);                // dK^T += Q^T dS
      }
    }
    staged = true;
    if (k0 + r < nk && 4 * g < dh) {
      float *pk = a.dKp, *pv = a.dVp;
      int64_t o = ((int64_t)b * nk + k0 + r) * d + j * dh + 4 * g;
      if (a.S > 1) {
        pk = a.part;
        pv = a.part + (int64_t)a.B * a.S * nk * d;
        o = (((int64_t)b * a.S + sp) * nk + k0 + r) * d + j * dh + 4 * g;
      }
      *reinterpret_cast<float4*>(pk + o) = float4{dk[0], dk[1], dk[2], dk[3]};
      *reinterpret_cast<float4*>(pv + o) = float4{dv[0], dv[1], dv[2], dv[3]};
    }
  }
}

inline AttnCoreArgs core_args(const pca_mab_shape& s, const Plan& p) {
  AttnCoreArgs a{};
  a.B = s.B; a.nq = s.nq; a.nk = s.nk; a.d = s.d; a.dh = s.d / s.h; a.h = s.h;
  a.qb = s.q_shared ? 0 : (int64_t)s.nq * s.d;
  a.scale = 1.0f / sqrtf((float)s.d);                 // modules.py:28: sqrt(dim_V)
  a.c = a.scale * 1.4426950408889634f;
  a.lengths = s.k_lengths;
  a.nt = p.nt; a.gt = p.gt; a.S = p.S; a.cps = p.cps;
  return a;
}
inline unsigned grid_of(const pca_mab_shape& s, const Plan& p) {
  return (unsigned)(8 * p.gt * p.S * cdiv(s.B, 8));
}
inline size_t fwd_part_elems(const pca_mab_shape& s, const Plan& p) {
  return p.S > 1 ? (size_t)s.B * p.S * s.nq * (2 * s.h + s.d) : 0;
}

}  // namespace

// head dims whose 16-byte row pieces the K = 16 MFMA serves, d small enough for the staged chunks
// (44 KB of LDS at d = 64); the exact fp32 mode keeps its chain (the parity path materialises A like
// the reference)
bool attn_core_ok(const pca_mab_shape& s) {
  const int dh = s.d / s.h;
  return s.mode != PCA_MODE_F32 && !s.ln && dh <= 16 && dh % 4 == 0 && s.d % 4 == 0 && s.d <= 64 &&
         s.h <= 16;
}

// floats behind `LSE` (forward) / `Delta` (backward): the statistics [B][h][nq], then the partial
// results of a launch whose streamed side is cut (Plan::S > 1)
size_t attn_core_fwd_elems(const pca_mab_shape& s) {
  return (size_t)s.B * s.h * s.nq + fwd_part_elems(s, plan_of(s.B, s.nq, s.nk));
}
size_t attn_core_bwd_elems(const pca_mab_shape& s) {
  const Plan pq = plan_of(s.B, s.nq, s.nk), pk = plan_of(s.B, s.nk, s.nq);
  const size_t eq = pq.S > 1 ? (size_t)s.B * pq.S * s.nq * s.d : 0;
  const size_t ek = pk.S > 1 ? 2 * (size_t)s.B * pk.S * s.nk * s.d : 0;
  return (size_t)s.B * s.h * s.nq + (eq > ek ? eq : ek);
}

int attn_core_fwd(const pca_mab_shape& s, const float* Qp, const float* Kp, const float* Vp, float* O,
                  float* LSE, hipStream_t st) {
  const Plan p = plan_of(s.B, s.nq, s.nk);
  AttnCoreArgs a = core_args(s, p);
  a.Qp = Qp; a.Kp = Kp; a.Vp = Vp; a.Oout = O; a.LSE = LSE;
  a.part = LSE + (size_t)s.B * s.h * s.nq;
  const size_t ldsb = (size_t)(row_elems(s.d) + tr_elems(s.d)) * sizeof(__bf16);
  if (s.h <= 8) hipLaunchKernelGGL(k_attnc_fwd<8>, dim3(grid_of(s, p)), dim3(64 * s.h), ldsb, st, a);
  else hipLaunchKernelGGL(k_attnc_fwd<16>, dim3(grid_of(s, p)), dim3(64 * s.h), ldsb, st, a);
  PCA_TRY(check_launch("k_attnc_fwd"));
  if (p.S > 1) {
    const int64_t n = (int64_t)s.B * s.nq * s.h * (a.dh / 4);
    hipLaunchKernelGGL(k_attnc_fwd_merge, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, a);
    PCA_TRY(check_launch("k_attnc_fwd_merge"));
  }
  return PCA_OK;
}

// Delta: attn_core_bwd_elems(s) floats of scratch
int attn_core_bwd(const pca_mab_shape& s, const float* Qp, const float* Kp, const float* Vp,
                  const float* O, const float* LSE, const float* dO, float* dQp, float* dKp, float* dVp,
                  float* Delta, hipStream_t st) {
  float* part = Delta + (size_t)s.B * s.h * s.nq;
  {
    const Plan p = plan_of(s.B, s.nq, s.nk);
    AttnCoreArgs a = core_args(s, p);
    a.Qp = Qp; a.Kp = Kp; a.Vp = Vp; a.O = O; a.LSE = const_cast<float*>(LSE); a.dO = dO;
    a.dQp = dQp; a.Delta = Delta; a.part = part;
    const size_t ldsb = (size_t)(2 * row_elems(s.d) + tr_elems(s.d)) * sizeof(__bf16);
    if (s.h <= 8) hipLaunchKernelGGL(k_attnc_bwd_q<8>, dim3(grid_of(s, p)), dim3(64 * s.h), ldsb, st, a);
  else hipLaunchKernelGGL(k_attnc_bwd_q<16>, dim3(grid_of(s, p)), dim3(64 * s.h), ldsb, st, a);
    PCA_TRY(check_launch("k_attnc_bwd_q"));
    if (p.S > 1) {
      const int64_t per4 = (int64_t)s.nq * s.d / 4;
      hipLaunchKernelGGL(k_attnc_sum, dim3((unsigned)cdiv(s.B * per4, 256)), dim3(256), 0, st, dQp, dO,
                         part, s.B, p.S, per4);
      PCA_TRY(check_launch("k_attnc_sum"));
    }
  }
  const Plan p = plan_of(s.B, s.nk, s.nq);
  AttnCoreArgs a = core_args(s, p);
  a.Qp = Qp; a.Kp = Kp; a.Vp = Vp; a.LSE = const_cast<float*>(LSE); a.dO = dO;
  a.dKp = dKp; a.dVp = dVp; a.Delta = Delta; a.part = part;
  const size_t nb16 = 2 * (size_t)row_elems(s.d) + 2 * (size_t)tr_elems(s.d);
  const size_t ldsb = ((nb16 + 7) & ~(size_t)7) * sizeof(__bf16) + 2 * (size_t)s.h * CH * sizeof(float);
  if (s.h <= 8) hipLaunchKernelGGL(k_attnc_bwd_kv<8>, dim3(grid_of(s, p)), dim3(64 * s.h), ldsb, st, a);
  else hipLaunchKernelGGL(k_attnc_bwd_kv<16>, dim3(grid_of(s, p)), dim3(64 * s.h), ldsb, st, a);
  PCA_TRY(check_launch("k_attnc_bwd_kv"));
  if (p.S > 1) {
    const int64_t per4 = (int64_t)s.nk * s.d / 4;
    const unsigned grid = (unsigned)cdiv(s.B * per4, 256);
    hipLaunchKernelGGL(k_attnc_sum, dim3(grid), dim3(256), 0, st, dKp, (const float*)nullptr, part, s.B,
                       p.S, per4);
    PCA_TRY(check_launch("k_attnc_sum"));
    hipLaunchKernelGGL(k_attnc_sum, dim3(grid), dim3(256), 0, st, dVp, (const float*)nullptr,
                       part + (size_t)s.B * p.S * s.nk * s.d, s.B, p.S, per4);
    PCA_TRY(check_launch("k_attnc_sum"));
  }
  return PCA_OK;
}

}  // namespace pca
