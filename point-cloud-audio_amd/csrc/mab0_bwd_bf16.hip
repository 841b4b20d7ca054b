// Fused bf16-MFMA backward of the "few shared queries, many keys" MAB (ISAB mab0 / PMA) in
// the reassociated form of mab0_bf16.hip.  With G' = sl2e * Qp_h Wk_h (sl2e = log2(e)/sqrt d),
// P = exp2(G' X^T - lse), T = P X and the forward's epilogue O_h = Qp_h + T_h Wv_h^T + bv_h,
// H = O + relu(O Wo^T + bo):
//
//   k_mab0_epi_bwd (per set, fp32 VALU)   dZ = dH.[Z>0] ; dO = dH + dZ Wo ; dT_h = dO_h Wv_h ;
//                                         Delta = rowdot(dT, T)
//   k_mab0_bwd     (per set, MFMA)        per 32-point tile of a wave, transposed layout
//        S^T = G' X^T, dA^T = dT X^T  ->  P^T, dS^T = ln2 . P^T (dA^T - Delta)
//        dX^T += dT^T P^T + G'^T dS^T        (both sums run over accumulator ROWS: in registers)
//        dG   += dS X                         (sum over points: dS^T goes through a wave-private
//                                              LDS tile and comes back transposed, X^T through
//                                              ds_read_tr16_b64 of the X tile)
//   k_mab0_bwd_small                      layer 1 (dk <= 4): fp32 VALU, no dX needed
//   parameter gradients of the epilogue (dWo, dWv, dbo, dbv) are [B*m]-row reductions done with
//   the fp32 GEMM; k_mab0_post turns sum_b dO and dG into dWk, dWq, dbq, dI.  d(bk) is
//   identically zero (softmax shift invariance) and is left untouched.
#include "mab1_bf16.hpp"
#include "terminal_bodies.hpp"
#include "pma_head_bodies.hpp"
#include "slab_sum_body.hpp"

#include <math.h>
#include <stdlib.h>

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
// [rows][RP] bf16 image (RP*2 bytes per row), 16-byte chunks XOR-swizzled by the row
template <int RP>
__device__ __forceinline__ int rp_off(int row, int ch) {
  constexpr int NCH = RP / 8;
  return row * RP * 2 + ((ch ^ (row & (NCH - 1))) << 4);
}

// ---------------------------------------------------------------------------------
// per-set epilogue adjoint (fp32)
// ---------------------------------------------------------------------------------
// thread = (output column, half of the queries); weights are read in their natural nn.Linear
// layout, which is already coalesced for these products (sum over the OUTPUT index).
template <int MQ>
__global__ __launch_bounds__(256) void k_mab0_epi_bwd(
    const float* __restrict__ dH, const float* __restrict__ Z, const float* __restrict__ T,
    const float* __restrict__ LSE, const float* __restrict__ Wo, const float* __restrict__ Wv,
    int m, int d, int dk, int h, int Rp, float* __restrict__ dZ, float* __restrict__ dO,
    float* __restrict__ Th,      // [h][B*m][dk] head-major copy of T (for dWv)
    float* __restrict__ dTf,     // [B][R][dk] fp32 (small-dk path)
    __bf16* __restrict__ dTb,    // [B][Rp][dk] natural rows
    __bf16* __restrict__ dTt,    // [B][dk][Rp] r-permuted
    float* __restrict__ Delta,   // [B][Rp]
    float* __restrict__ LSEp,    // [B][Rp] padded with +1e30
    int B, float* __restrict__ zero_ptr, int zero_n) {
  mab0_epi_bwd_body<MQ>(dH, Z, T, LSE, Wo, Wv, m, d, dk, h, Rp, dZ, dO, Th, dTf, dTb, dTt, Delta, LSEp, B, zero_ptr, zero_n, blockIdx.x);
}

// G'^T image shared by all sets: GtP[c][32 s + p] = G'[32 s + perm32(p)][c]  (zero padding)
__global__ void k_mab0_gt(const float* __restrict__ Gf, int R, int Rp, int dk,
                          __bf16* __restrict__ GtP) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= dk * Rp) return;
  const int c = idx / Rp, k = idx - c * Rp;
  const int r = (k & ~31) + perm32(k & 31);
  GtP[idx] = (__bf16)(r < R ? Gf[r * dk + c] : 0.f);
}

// ---------------------------------------------------------------------------------
// main backward over the points, dk == 128
// ---------------------------------------------------------------------------------
struct Mab0BwdArgs {
  const void* X;         // [B, N, 128] fp32, or bf16 when ABF
  const __bf16* Gb;      // [Rp][128] natural rows (sl2e folded in)
  const __bf16* GtP;     // [128][Rp]
  const __bf16* dTb;     // [B][Rp][128]
  const __bf16* dTt;     // [B][128][Rp]
  const float* LSEp;     // [B][Rp]
  const float* Delta;    // [B][Rp]
  void* dX;              // [B, N, 128] (fp32 / bf16 when ABF) or null
  float* DG;             // [Rp][128] fp32, accumulated over sets (ln2-scaled dS units)
  int B, N, accumulate_dx, S;
  const int32_t* lengths;   // [B] valid points per set, or null
  int R;                    // real score rows (<= RP): only these rows of dG are non-zero
  float* slabs;             // [B*S][R][128]: per-workgroup dG (summed in a fixed order afterwards)
};

template <int RP, bool ABF>
__global__ __launch_bounds__(256, 1) void k_mab0_bwd(const Mab0BwdArgs a) {
  constexpr int DK = 128, FT = DK / 16, KS = DK / 32, RB = RP / 16, RS = RP / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sG = smem;                           // [RP][256]   tr_off rows
  char* sdT = sG + RP * 256;                 // [RP][256]
  char* sGt = sdT + RP * 256;                // [128][RP*2] rp_off
  char* sdTt = sGt + DK * RP * 2;            // [128][RP*2]
  char* sX = sdTt + DK * RP * 2;             // 4 x 32 x 256
  char* sDS = sX + 4 * 32 * 256;             // 4 x 32 x 256 (first RP*2 bytes of a row used)
  float* sLSE = reinterpret_cast<float*>(sDS + 4 * 32 * 256);
  float* sDel = sLSE + RP;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int b = blockIdx.x, sp = blockIdx.y;
  const int per = (int)(((int64_t)(a.N + 127) / 128 + a.S - 1) / a.S) * 128;
  const int n_lo = sp * per, n_hi = (n_lo + per < a.N) ? n_lo + per : a.N;
  // variable-size sets: points at and beyond len have P = 0 (so dS = 0 and their dX rows are
  // written as exact zeros); the loops still cover all N rows
  int len = a.N;
  if (a.lengths != nullptr) len = a.lengths[b] < a.N ? a.lengths[b] : a.N;

  {
    // all image chunks of this thread are fetched before the first LDS store: one round trip
    // instead of one per loop iteration (the stores would order the loads behind them)
    constexpr int NA = RP * 16 / 256, NB2 = DK * (RP / 8) / 256;
    uint4 g1[NA], t1[NA], g2[NB2], t2[NB2];
#pragma unroll
    for (int e = 0; e < NA; ++e) {
      const int c = tid + 256 * e, row = c >> 4, ch = c & 15;
      g1[e] = *reinterpret_cast<const uint4*>(a.Gb + (int64_t)row * DK + ch * 8);
      t1[e] = *reinterpret_cast<const uint4*>(a.dTb + ((int64_t)b * RP + row) * DK + ch * 8);
    }
#pragma unroll
    for (int e = 0; e < NB2; ++e) {
      const int c = tid + 256 * e, row = c / (RP / 8), ch = c % (RP / 8);
      g2[e] = *reinterpret_cast<const uint4*>(a.GtP + (int64_t)row * RP + ch * 8);
      t2[e] = *reinterpret_cast<const uint4*>(a.dTt + ((int64_t)b * DK + row) * RP + ch * 8);
    }
#pragma unroll
    for (int e = 0; e < NA; ++e) {
      const int c = tid + 256 * e, row = c >> 4, ch = c & 15;
      *reinterpret_cast<uint4*>(sG + tr_off(row, ch)) = g1[e];
      *reinterpret_cast<uint4*>(sdT + tr_off(row, ch)) = t1[e];
    }
#pragma unroll
    for (int e = 0; e < NB2; ++e) {
      const int c = tid + 256 * e, row = c / (RP / 8), ch = c % (RP / 8);
      *reinterpret_cast<uint4*>(sGt + rp_off<RP>(row, ch)) = g2[e];
      *reinterpret_cast<uint4*>(sdTt + rp_off<RP>(row, ch)) = t2[e];
    }
  }
  for (int i = tid; i < RP; i += 256) {
    sLSE[i] = a.LSEp[(int64_t)b * RP + i];
    sDel[i] = a.Delta[(int64_t)b * RP + i];
  }
  __syncthreads();

  char* myX = sX + wave * 32 * 256;
  char* myDS = sDS + wave * 32 * 256;
  f32x4 dG[RB][FT];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int ft = 0; ft < FT; ++ft) dG[rb][ft] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr float LN2 = 0.6931471805599453f;

  // the wave's next X tile is fetched into registers while the current one is worked on (the kernel
  // runs one wave per SIMD: registers are free, and every tile used to start with an exposed round trip)
  bf16x8 nx[8];
  auto fetch_tile = [&](int n0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = lane + 64 * e;
      // (bf16 activations: zeroed when the tile goes to LDS, see ld_x8_clamped)
      nx[e] = ABF ? ld_x8_clamped(a.X, (int64_t)b * a.N, n0 + (c >> 4), n_hi, DK, c & 15)
                  : ld_x8_guard<ABF>(a.X, (int64_t)b * a.N, n0 + (c >> 4), n_hi, DK, c & 15);
    }
  };
  if (n_lo + wave * 32 < n_hi) fetch_tile(n_lo + wave * 32);
  for (int n0 = n_lo + wave * 32; n0 < n_hi; n0 += 128) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = lane + 64 * e;
      const int row = c >> 4, ch = c & 15;
      *reinterpret_cast<bf16x8*>(myX + tr_off(row, ch)) = zero_unless(n0 + row < n_hi, nx[e]);
    }
    if (n0 + 128 < n_hi) fetch_tile(n0 + 128);
    // the rows this tile's dX is added onto (mab1's dQ part) are fetched now, a whole tile of work
    // ahead of their use: read at the end they were an exposed round trip per tile (5 of 37 us)
    bf16x8 od[8];
    if (ABF && a.dX != nullptr && a.accumulate_dx) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = lane + 64 * e, n = n0 + (c >> 4);
        od[e] = *reinterpret_cast<const bf16x8*>(
            reinterpret_cast<const __bf16*>(a.dX) +
            ((int64_t)b * a.N + (n < n_hi ? n : n_hi - 1)) * DK + (c & 15) * 8);
      }
    }
    bf16x8 xrow[2][KS];
#pragma unroll
    for (int pb = 0; pb < 2; ++pb)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        xrow[pb][ks] = *reinterpret_cast<const bf16x8*>(myX + tr_off(16 * pb + r, 4 * ks + g));
    bool live[2], key[2];          // row exists / row is a key of the (possibly shorter) set
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) {
      live[pb] = n0 + 16 * pb + r < n_hi;
      key[pb] = n0 + 16 * pb + r < len;
    }

    f32x4 dx[FT][2];
#pragma unroll
    for (int ft = 0; ft < FT; ++ft)
#pragma unroll
      for (int pb = 0; pb < 2; ++pb) dx[ft][pb] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int s = 0; s < RS; ++s) {
      f32x4 pt[2][2], dst[2][2];          // [row block within the 32][point block]
#pragma unroll
      for (int rbi = 0; rbi < 2; ++rbi) {
        const int rb = 2 * s + rbi;
        const float4 l4 = *reinterpret_cast<const float4*>(&sLSE[16 * rb + 4 * g]);
        const float4 d4 = *reinterpret_cast<const float4*>(&sDel[16 * rb + 4 * g]);
        const float lse[4] = {l4.x, l4.y, l4.z, l4.w}, del[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
          f32x4 sv = {0.f, 0.f, 0.f, 0.f}, da = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            sv = mfma32(*reinterpret_cast<const bf16x8*>(sG + tr_off(16 * rb + r, 4 * ks + g)),
                        xrow[pb][ks], sv);
            da = mfma32(*reinterpret_cast<const bf16x8*>(sdT + tr_off(16 * rb + r, 4 * ks + g)),
                        xrow[pb][ks], da);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float p = key[pb] ? exp2f(sv[e] - lse[e]) : 0.f;
            pt[rbi][pb][e] = p;
            dst[rbi][pb][e] = LN2 * p * (da[e] - del[e]);
          }
          // dS^T tile -> wave-private [point][r] image (for the sum over points below)
          *reinterpret_cast<bf16x4*>(myDS + tr_off(16 * pb + r, 2 * rb + (g >> 1)) + 8 * (g & 1)) =
              pack4(dst[rbi][pb]);
        }
      }
      if (a.dX != nullptr) {
        bf16x8 pf[2], df[2];
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
          pf[pb] = pack8(pt[0][pb], pt[1][pb]);
          df[pb] = pack8(dst[0][pb], dst[1][pb]);
        }
#pragma unroll
        for (int ft = 0; ft < FT; ++ft) {
          const bf16x8 ta =
              *reinterpret_cast<const bf16x8*>(sdTt + rp_off<RP>(16 * ft + r, 4 * s + g));
          const bf16x8 ga =
              *reinterpret_cast<const bf16x8*>(sGt + rp_off<RP>(16 * ft + r, 4 * s + g));
#pragma unroll
          for (int pb = 0; pb < 2; ++pb) {
            dx[ft][pb] = mfma32(ta, pf[pb], dx[ft][pb]);
            dx[ft][pb] = mfma32(ga, df[pb], dx[ft][pb]);
          }
        }
      }
    }

    // dG[r][c] += sum_points dS[r][pt] X[pt][c]
    bf16x8 xtr[FT];
#pragma unroll
    for (int ft = 0; ft < FT; ++ft) xtr[ft] = tr_frag(myX, ft, lane);
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const bf16x8 da = tr_frag(myDS, rb, lane);
#pragma unroll
      for (int ft = 0; ft < FT; ++ft) dG[rb][ft] = mfma32(da, xtr[ft], dG[rb][ft]);
    }

    if (a.dX != nullptr && ABF) {
      // bf16 dX: the [32 points][128] tile is assembled in the wave's own LDS tile (X is no longer
      // needed) and stored / accumulated in 16-byte pieces of full rows, the loads unconditional -
      // from the accumulator layout it was 16 guarded read-modify-writes of 8 bytes per lane, each
      // waited for on its own
#pragma unroll
      for (int pb = 0; pb < 2; ++pb)
#pragma unroll
        for (int ft = 0; ft < FT; ++ft)
          *reinterpret_cast<bf16x4*>(myX + swz(16 * pb + r, 2 * ft + (g >> 1), 2 * DK) + 8 * (g & 1)) =
              pack4(dx[ft][pb]);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = lane + 64 * e, row = c >> 4, ch = c & 15;
        const int n = n0 + row;
        bf16x8 v = *reinterpret_cast<const bf16x8*>(myX + swz(row, ch, 2 * DK));
        __bf16* pd = reinterpret_cast<__bf16*>(a.dX) +
                     ((int64_t)b * a.N + (n < n_hi ? n : n_hi - 1)) * DK + ch * 8;
        if (a.accumulate_dx) {
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] = (__bf16)((float)v[k] + (float)od[e][k]);
        }
        if (n < n_hi) *reinterpret_cast<bf16x8*>(pd) = v;
      }
    } else if (a.dX != nullptr) {
#pragma unroll
      for (int pb = 0; pb < 2; ++pb)
        if (live[pb]) {
          const int64_t ro = ((int64_t)b * a.N + n0 + 16 * pb + r) * DK;
#pragma unroll
          for (int ft = 0; ft < FT; ++ft) {
            f32x4 v = dx[ft][pb];
            if (ABF) {
              bf16x4* pd = reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(a.dX) + ro +
                                                     16 * ft + 4 * g);
              if (a.accumulate_dx) {
                const bf16x4 o = *pd;
                v[0] += (float)o[0]; v[1] += (float)o[1]; v[2] += (float)o[2]; v[3] += (float)o[3];
              }
              *pd = pack4(v);
            } else {
              float4* pd = reinterpret_cast<float4*>(reinterpret_cast<float*>(a.dX) + ro +
                                                     16 * ft + 4 * g);
              float4 o4 = float4{v[0], v[1], v[2], v[3]};
              if (a.accumulate_dx) {
                const float4 o = *pd;
                o4.x += o.x; o4.y += o.y; o4.z += o.z; o4.w += o.w;
              }
              *pd = o4;
            }
          }
        }
    }
  }

  // ---- dG: per-wave slabs (plain LDS stores over the dead images), then one global atomic
  //      per element per workgroup ----
  __syncthreads();
  float* slab = reinterpret_cast<float*>(smem) + wave * RP * DK;      // 4 x RP x 128 fp32
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int ft = 0; ft < FT; ++ft)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        slab[(16 * rb + 4 * g + e) * DK + 16 * ft + r] = dG[rb][ft][e];
  __syncthreads();
  const float* s0 = reinterpret_cast<const float*>(smem);
  // (PMA: 4 of the 32 padded rows are real.)  One slab per workgroup, plain stores: 256 workgroups
  // adding into one [R][128] block cost 11 of this kernel's 37 us at B = 128 - the adds to one
  // address serialise at ~25 ns each - and made the result depend on their order
  float* out = a.slabs + ((int64_t)b * a.S + sp) * a.R * DK;
  for (int i = tid; i < a.R * DK; i += 256)
    out[i] = s0[i] + s0[RP * DK + i] + s0[2 * RP * DK + i] + s0[3 * RP * DK + i];
}

// layer 1 (dk <= 4): thread = (query row r, point partition); accumulates DG only.  The
// set's points are staged in LDS chunk by chunk (coalesced) and re-read from there.
__global__ __launch_bounds__(256) void k_mab0_bwd_small(
    const float* __restrict__ X, const float* __restrict__ Gf, const float* __restrict__ dTf,
    const float* __restrict__ LSE, const float* __restrict__ Delta, int N, int R, int Rp, int dk,
    float* __restrict__ DG, const int32_t* __restrict__ lengths, float* __restrict__ slabs) {
  constexpr int CH = PCA_POINT_CHUNK;
  __shared__ float sD[256][4];
  __shared__ __attribute__((aligned(16))) float sX[CH * 4];
  const int b = blockIdx.x, tid = threadIdx.x;
  // gridDim.y workgroups share the rows of a set: this one owns rows [row0, row0 + Rb)
  const int Rb = R / gridDim.y, row0 = blockIdx.y * Rb;
  const int parts = 256 / Rb;
  const int rl = tid % Rb, part = tid / Rb;
  const int r = row0 + rl;
  constexpr float LN2 = 0.6931471805599453f;
  float gk[4], dt[4], acc[4];
  f32x2 acc2[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    gk[c] = c < dk ? Gf[r * dk + c] : 0.f;
    dt[c] = c < dk ? dTf[((int64_t)b * R + r) * dk + c] : 0.f;
  }
  const float lse = LSE[(int64_t)b * R + r], del = Delta[(int64_t)b * Rp + r];
  int len = N;
  if (lengths != nullptr) len = lengths[b] < N ? lengths[b] : N;
  float xr[16];
  if (len > 0) fetch_points(X + (int64_t)b * N * dk, len < CH ? len : CH, dk, xr);
  for (int n0 = 0; n0 < len; n0 += CH) {
    const int cn = (len - n0 < CH) ? len - n0 : CH;
    // [pair of points][component][2] (as k_mab0_attn_small): two points per packed-fp32 instruction;
    // unused components and the odd point of the chunk are zero (x = 0 contributes dS.x = 0); the
    // next chunk's loads are in flight while this one is worked on
    const int npairs = (cn + 1) >> 1;
    commit_points(xr, cn, dk, sX);
    if (n0 + CH < len)
      fetch_points(X + ((int64_t)b * N + n0 + CH) * dk, (len - n0 - CH < CH) ? len - n0 - CH : CH, dk, xr);
    if (part < parts) {
#pragma unroll 4
      for (int q = part; q < npairs; q += parts) {
        const float4 lo = *reinterpret_cast<const float4*>(&sX[q * 8]);
        const float4 hi = *reinterpret_cast<const float4*>(&sX[q * 8 + 4]);
        const f32x2 x0 = {lo.x, lo.y}, x1 = {lo.z, lo.w}, x2 = {hi.x, hi.y}, x3 = {hi.z, hi.w};
        const f32x2 sc = gk[0] * x0 + gk[1] * x1 + gk[2] * x2 + gk[3] * x3;
        const f32x2 da = dt[0] * x0 + dt[1] * x1 + dt[2] * x2 + dt[3] * x3;
        const f32x2 pe = {__builtin_amdgcn_exp2f(sc[0] - lse), __builtin_amdgcn_exp2f(sc[1] - lse)};
        const f32x2 ds = (LN2 * pe) * (da - del);
        acc2[0] += ds * x0; acc2[1] += ds * x1; acc2[2] += ds * x2; acc2[3] += ds * x3;
      }
    }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) acc[c] = acc2[c][0] + acc2[c][1];
#pragma unroll
  for (int c = 0; c < 4; ++c) sD[tid][c] = acc[c];
  __syncthreads();
  if (tid < Rb) {
    for (int c = 0; c < dk; ++c) {
      float v = 0.f;
      for (int p = 0; p < parts; ++p) v += sD[p * Rb + tid][c];
      // slabs != nullptr: per-set partial [b][R][dk] (summed in a fixed order by the caller)
      if (slabs != nullptr) slabs[((int64_t)b * R + row0 + tid) * dk + c] = v;
      else atomicAdd(&DG[(row0 + tid) * dk + c], v);
    }
  }
}

// ---------------------------------------------------------------------------------
// shared-query parameters.  dQs = sum_b dO[b] ([m][d]); DG = sum over sets of dS X in
// "ln2 units": dG_raw = sl2e * DG.
//   dWk[f][c]  += sum_q Qp[q][f] dG_raw[j m + q][c]            (j = head of f)
//   dQp[q][f]   = dQs[q][f] + sum_c dG_raw[j m + q][c] Wk[f][c]
//   dWq += dQp^T I ; dbq += colsum(dQp) ; dI += dQp Wq
// ---------------------------------------------------------------------------------
// dot of two strided sequences with NF independent load pairs in flight (the post kernels are a
// few dependent L2 round trips long and nothing else: a 128-term dot is 2 trips at NF = 64, 8 at 16)
template <int NF = 64>
__device__ __forceinline__ float dot_strided(const float* __restrict__ a, int64_t sa,
                                             const float* __restrict__ b, int64_t sb, int n) {
  float acc = 0.f;
  int i = 0;
  for (; i + NF <= n; i += NF) {
    float x[NF], y[NF];
#pragma unroll
    for (int u = 0; u < NF; ++u) { x[u] = a[(i + u) * sa]; y[u] = b[(i + u) * sb]; }
#pragma unroll
    for (int u = 0; u < NF; ++u) acc = fmaf(x[u], y[u], acc);
  }
  for (; i + 16 <= n; i += 16) {
    float x[16], y[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) { x[u] = a[(i + u) * sa]; y[u] = b[(i + u) * sb]; }
#pragma unroll
    for (int u = 0; u < 16; ++u) acc = fmaf(x[u], y[u], acc);
  }
  for (; i < n; ++i) acc = fmaf(a[i * sa], b[i * sb], acc);
  return acc;
}

// stage 1 (grid-parallel): dWk and dQp = dQs + (dG_raw Wk_h^T); blockIdx.y = MAB
__device__ __forceinline__ void post1_body(const Mab0PostJob& a, int blk);
__global__ __launch_bounds__(256) void k_mab0_post1(const Mab0PostJobs jobs) {
  post1_body(jobs.j[blockIdx.y], blockIdx.x);
}
__device__ __forceinline__ void post1_body(const Mab0PostJob& a, int blk) {
  const int m = a.m, d = a.d, dk = a.dk;
  const int dh = d / a.h;
  const int o = blk * 256 + threadIdx.x;
  if (o < d * dk) {
    if (a.DG == nullptr) return;            // keys were projected: dWk comes from the GEMM path
    const int f = o / dk, c = o - f * dk, j = f / dh;
    a.dWk[o] += a.sl2e * dot_strided(a.Qp + f, d, a.DG + (int64_t)j * m * dk + c, dk, m);
  } else if (o < d * dk + m * d) {
    const int oo = o - d * dk;
    const int q = oo / d, f = oo - q * d, j = f / dh;
    float qs;
    if (a.dQs != nullptr) {
      qs = a.dQs[oo];
    } else {                                  // sum over the sets, 16 loads in flight
      qs = 0.f;
      const int64_t sb = (int64_t)m * d;
      int bb = 0;
      for (; bb + 64 <= a.B; bb += 64) {
        float v[64];
#pragma unroll
        for (int u = 0; u < 64; ++u) v[u] = a.dO[(bb + u) * sb + oo];
#pragma unroll
        for (int u = 0; u < 64; ++u) qs += v[u];
      }
      for (; bb + 16 <= a.B; bb += 16) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = a.dO[(bb + u) * sb + oo];
#pragma unroll
        for (int u = 0; u < 16; ++u) qs += v[u];
      }
      for (; bb < a.B; ++bb) qs += a.dO[bb * sb + oo];
    }
    a.dQp[oo] = a.DG == nullptr ? qs
                                : qs + a.sl2e * dot_strided(a.DG + (int64_t)(j * m + q) * dk, 1,
                                                            a.Wk + (int64_t)f * dk, 1, dk);
  }
}
// stage 1 + riders: rows [0, J.n) of blockIdx.y are the post-1 jobs, then (when present) the
// classifier weight gradient (one workgroup per class) and the layer-1 fc_v gradient
__device__ __forceinline__ void post1_body(const Mab0PostJob& a, int blk);
__global__ __launch_bounds__(256) void k_terminal1(const Mab0PostJobs jobs, const ClsWgradArgs c,
                                                   int has_cls, const SmallWgradArgs w,
                                                   int has_sw, const SlabSumJobs late) {
  const int y = blockIdx.y;
  if (y < jobs.n) {
    post1_body(jobs.j[y], blockIdx.x);
  } else if (has_cls && y == jobs.n) {
    if ((int)blockIdx.x < c.C)
      cls_wgrad_body(c.dlogits, c.P, c.lossv, c.corrv, c.B, c.d, c.C, c.dWc, c.dbc, c.loss_out,
                     c.stats, blockIdx.x);
  } else if (has_sw && y == jobs.n + has_cls) {
    if ((int64_t)blockIdx.x * w.rows_per_wg < w.M)
      wgrad_small_body<float>(w.G, w.X, w.M, w.dq, w.rows_per_wg, w.x_head_stride, w.dW, w.db,
                              blockIdx.x, w.slab);
  } else {
    // rider rows: the weight-gradient slabs of this step, added in a fixed order
    __shared__ float4 red[4 * 64];
    slab_sum_body(late.j[y - jobs.n - has_cls - has_sw], blockIdx.x, threadIdx.x, red);
  }
}
// stage 2: dWq += dQp^T I ; dbq += colsum(dQp) ; dI += dQp Wq
__global__ __launch_bounds__(256) void k_mab0_post2(const Mab0PostJobs jobs, const SlabSumJobs late) {
  if ((int)blockIdx.y >= jobs.n) {       // rider rows (partials written by k_terminal1 itself)
    __shared__ float4 red[4 * 64];
    slab_sum_body(late.j[blockIdx.y - jobs.n], blockIdx.x, threadIdx.x, red);
    return;
  }
  const Mab0PostJob a = jobs.j[blockIdx.y];
  const int m = a.m, d = a.d, dq = a.dq;
  const int o = blockIdx.x * 256 + threadIdx.x;
  const int n1 = d * dq, n2 = n1 + d, n3 = n2 + (a.dI != nullptr ? m * dq : 0);
  if (o < n1) {
    const int f = o / dq, c = o - f * dq;
    a.dWq[o] += dot_strided(a.dQp + f, d, a.I + c, dq, m);
  } else if (o < n2) {
    const int f = o - n1;
    float acc = 0.f;
    for (int q = 0; q < m; ++q) acc += a.dQp[q * d + f];
    a.dbq[f] += acc;
  } else if (o < n3) {
    const int oo = o - n2;
    const int q = oo / dq, c = oo - q * dq;
    a.dI[oo] += dot_strided(a.dQp + (int64_t)q * d, 1, a.Wq + c, dq, d);
  }
}

}  // namespace
namespace {
// Everything a set needs between its attention forward and its attention backward.  The stages
// hand over through global memory written by this very workgroup (H = pooled features, Z, T,
// LSE, dP): a workgroup barrier makes those stores visible to the next stage.
__global__ __launch_bounds__(256) void k_pma_head(const PmaHeadArgs a) {
  const int b = blockIdx.x;
  mab0_epi_body<1>(a.Tp, a.Mp, a.Lp, a.S, a.T, a.LSE, a.Qp, a.WvT, a.bv, a.WoT, a.bo, a.m, a.d,
                   a.dk, a.h, a.H, a.Osave, a.Zsave, b);
  __syncthreads();
  cls_fwd_bwd_body(a.H, a.Wc, a.bc, a.labels, a.B, a.d, a.C, a.grad_scale, a.logits, a.dlogits,
                   a.dP, a.lossv, a.corrv, b);
  __syncthreads();
  mab0_epi_bwd_body<1>(a.dP, a.Zsave, a.T, a.LSE, a.Wo, a.Wv, a.m, a.d, a.dk, a.h, a.Rp, a.dZ,
                       a.dO, a.Th, a.dTf, a.dTb, a.dTt, a.Delta, a.LSEp, a.B, a.zero_ptr,
                       a.zero_n, b);
}

// The same chain for the shape the engine actually runs it on (one seed, d = dk = 128, four heads,
// C <= 64): every weight element a thread will need - 64 + 64 + 32 + 32 + 64 + 64 floats over the
// six GEMV stages - is requested before the first stage, and the stages hand over through LDS
// instead of through the global arrays they also write.  The generic bodies above spend the
// launch in ~25 dependent L2 round trips (16 loads in flight each, one batch after the other,
// plus the re-reads of H / Z / T / dP between stages): 25 us for ~2 us of arithmetic.  Same
// products in the same order; Delta is reduced by shuffles instead of LDS float atomics.
__global__ __launch_bounds__(256) void k_pma_head1(const PmaHeadArgs a) {
  constexpr int D = 128, DK = 128, DH = 32, R = 4;        // (four heads)
  __shared__ float sT[R * DK], sO[D], sP[D], sZ[D], sL[64], sdZ[D], sdO[D], sdP[D], part[D];
  __shared__ float sLSE[R], sDl[2 * R], red[2];
  __shared__ int ramax;
  const int b = blockIdx.x, tid = threadIdx.x, f = tid & 127, half = tid >> 7;
  const int S = a.S, C = a.C;

  // ---- stage 0 loads first (they are needed first), then all the weights ----
  float mp[2][8], lp[2][8], tp[2][8];
  const int SS = S < 8 ? S : 8;
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int i = tid + 256 * e, r = i >> 7, c = i & 127;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const bool on = q < SS;
      const int64_t o = ((int64_t)b * S + (on ? q : 0)) * R + r;
      mp[e][q] = on ? a.Mp[o] : -INFINITY;
      lp[e][q] = on ? a.Lp[o] : 0.f;
      tp[e][q] = on ? a.Tp[o * DK + c] : 0.f;
    }
  }
  const float qb = a.Qp[f] + a.bv[f], bo_f = a.bo[f];
  float w1[64], w2[64], w5[64], w6[2][DH], w4[32];
  float4 w3[8];
#pragma unroll
  for (int c = 0; c < 64; ++c) w1[c] = a.WvT[(int64_t)(half * 64 + c) * D + f];
#pragma unroll
  for (int c = 0; c < 64; ++c) w2[c] = a.WoT[(int64_t)(half * 64 + c) * D + f];
  const int c3 = tid >> 2, part3 = tid & 3;
#pragma unroll
  for (int u = 0; u < 8; ++u)
    w3[u] = *reinterpret_cast<const float4*>(a.Wc + (int64_t)(c3 < C ? c3 : 0) * D + part3 * 32 + 4 * u);
  const float bc3 = a.bc[c3 < C ? c3 : 0];
  const int c0 = half * (C / 2), c1 = half ? C : C / 2;
#pragma unroll
  for (int u = 0; u < 32; ++u) w4[u] = a.Wc[(int64_t)(c0 + u < c1 ? c0 + u : c0) * D + f];
#pragma unroll
  for (int k = 0; k < 64; ++k) w5[k] = a.Wo[(int64_t)(half * 64 + k) * D + f];
#pragma unroll
  for (int jj = 0; jj < 2; ++jj)
#pragma unroll
    for (int k = 0; k < DH; ++k)
      w6[jj][k] = a.Wv[(int64_t)((2 * half + jj) * DH + k) * DK + f];
  const int64_t y = a.labels[b];
  if (a.zero_ptr != nullptr)
    for (int i = b * 256 + tid; i < a.zero_n; i += gridDim.x * 256) a.zero_ptr[i] = 0.f;

  // ---- forward epilogue: merge the S point-range partials (mab0_epi_body) ----
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int i = tid + 256 * e, r = i >> 7, c = i & 127;
    float M = -INFINITY;
#pragma unroll
    for (int q = 0; q < 8; ++q) M = fmaxf(M, mp[e][q]);
    float L = 0.f, t = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if (q >= SS || mp[e][q] == -INFINITY) continue;
      const float fs = exp2f(mp[e][q] - M);
      L += fs * lp[e][q];
      t += fs * tp[e][q];
    }
    const float v = t / L;
    sT[i] = v;
    a.T[(int64_t)b * R * DK + i] = v;
    if (c == 0) {
      const float lse = M + log2f(L);
      a.LSE[(int64_t)b * R + r] = lse;
      sLSE[r] = lse;
    }
  }
  __syncthreads();
  // O = Qp + T_h Wv_h^T + bv ; each half of the workgroup takes half of the contraction
  {
    const int j = f / DH;
    float a1 = half == 0 ? qb : 0.f;
#pragma unroll
    for (int c = 0; c < 64; ++c) a1 = fmaf(sT[j * DK + half * 64 + c], w1[c], a1);
    if (half == 1) part[f] = a1;
    __syncthreads();
    if (half == 0) sO[f] = a1 + part[f];
    __syncthreads();
  }
  {
    float z1 = half == 0 ? bo_f : 0.f;
#pragma unroll
    for (int c = 0; c < 64; ++c) z1 = fmaf(sO[half * 64 + c], w2[c], z1);
    if (half == 1) part[f] = z1;
    __syncthreads();
    if (half == 0) {
      z1 += part[f];
      const float o1 = sO[f], hv = o1 + fmaxf(z1, 0.f);
      const int64_t o = (int64_t)b * D + f;
      a.H[o] = hv;
      a.Osave[o] = o1;
      a.Zsave[o] = z1;
      sP[f] = hv;
      sZ[f] = z1;
    }
    __syncthreads();
  }
  // ---- classifier + cross-entropy, forward and backward (cls_fwd_bwd_body) ----
  {
    float acc = 0.f;
    if (c3 < C) {
      const float* x = sP + part3 * 32;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        acc += x[4 * u] * w3[u].x + x[4 * u + 1] * w3[u].y + x[4 * u + 2] * w3[u].z +
               x[4 * u + 3] * w3[u].w;
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    if (c3 < C && part3 == 0) {
      acc += bc3;
      sL[c3] = acc;
      a.logits[(int64_t)b * C + c3] = acc;
    }
  }
  __syncthreads();
  if (tid < 64) {
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
    float sm = 0.f;
    for (int j = tid; j < C; j += 64) sm += expf(sL[j] - m);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 64);
    if (tid == 0) { red[0] = m; red[1] = sm; ramax = am; }
  }
  __syncthreads();
  {
    const float m = red[0], sm = red[1];
    const float gs = a.grad_scale / (float)a.B;
    if (tid == 0) {
      a.lossv[b] = m + logf(sm) - sL[y];
      a.corrv[b] = ramax == (int)y ? 1.f : 0.f;
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
      const float g = (expf(sL[c] - m) / sm - (c == y ? 1.f : 0.f)) * gs;
      sL[c] = g;
      a.dlogits[(int64_t)b * C + c] = g;
    }
    __syncthreads();
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < 32; ++u)
      if (c0 + u < c1) acc = fmaf(sL[c0 + u], w4[u], acc);
    if (half == 1) part[f] = acc;
    __syncthreads();
    if (half == 0) {
      const float dp = acc + part[f];
      a.dP[(int64_t)b * D + f] = dp;
      sdP[f] = dp;
    }
    __syncthreads();
  }
  // ---- backward epilogue (mab0_epi_bwd_body): dZ, dO = dP + dZ Wo, dT_h = dO_h Wv_h, Delta ----
  if (tid < D) {
    const float v = sZ[tid] > 0.f ? sdP[tid] : 0.f;
    sdZ[tid] = v;
    a.dZ[(int64_t)b * D + tid] = v;
  }
  __syncthreads();
  {
    float a5 = half == 0 ? sdP[f] : 0.f;
#pragma unroll
    for (int k = 0; k < 64; ++k) a5 = fmaf(sdZ[half * 64 + k], w5[k], a5);
    if (half == 1) part[f] = a5;
    __syncthreads();
    if (half == 0) {
      a5 += part[f];
      sdO[f] = a5;
      a.dO[(int64_t)b * D + f] = a5;
    }
    __syncthreads();
  }
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) {
    const int j = 2 * half + jj;                  // head = score row r (one seed)
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < DH; ++k) acc = fmaf(sdO[j * DH + k], w6[jj][k], acc);
    const float tv = sT[j * DK + f];
    a.Th[((int64_t)j * a.B + b) * DK + f] = tv;
    a.dTb[((int64_t)b * a.Rp + j) * DK + f] = (__bf16)acc;
    int pos = 0;
#pragma unroll
    for (int p = 0; p < 32; ++p)
      if (perm32(p) == j) pos = p;
    a.dTt[((int64_t)b * DK + f) * a.Rp + pos] = (__bf16)acc;
    // Delta[r] = sum over the 128 columns: the wave's 64 by shuffles, the two waves through LDS
    float dl = acc * tv;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dl += __shfl_xor(dl, o, 64);
    if ((tid & 63) == 0) sDl[2 * j + ((tid >> 6) & 1)] = dl;
  }
  __syncthreads();
  for (int r = tid; r < a.Rp; r += 256) {
    a.Delta[(int64_t)b * a.Rp + r] = r < R ? sDl[2 * r] + sDl[2 * r + 1] : 0.f;
    a.LSEp[(int64_t)b * a.Rp + r] = r < R ? sLSE[r] : 1.0e30f;
  }
  for (int o = tid; o < (a.Rp - R) * DK; o += 256) {       // padding rows / columns of the images
    const int r = R + o / DK, cc = o % DK;
    a.dTb[((int64_t)b * a.Rp + r) * DK + cc] = (__bf16)0.f;
    const int rb32 = r & ~31, ro = r & 31;
    int pos = 0;
#pragma unroll
    for (int p = 0; p < 32; ++p)
      if (perm32(p) == ro) pos = p;
    a.dTt[((int64_t)b * DK + cc) * a.Rp + rb32 + pos] = (__bf16)0.f;
  }
}
}  // namespace

int pma_head_args(const pca_mab_shape& s, const pca_mab_params& p, void* saved, void* ws_bwd,
                  float* P, const float* Wc, const float* bc, const int64_t* labels, int C,
                  float grad_scale, float* logits, float* dlogits, float* dP, float* dWc,
                  float* dbc, float* loss_out, float* stats, float* cls_ws, BwdDefer* defer,
                  PmaHeadArgs* out) {
  PCA_REQUIRE(s.nq == 1 && s.dk > 4 && defer != nullptr, "pma_head: needs the fused PMA (k = 1)");
  Mab0Saved v;
  mab0_carve_saved(s, &v, saved);
  Mab0BwdWs w;
  mab0_carve_bwd_ws(s, &w, ws_bwd);
  const int d = s.d, m = s.nq, h = s.h, dk = s.dk, R = h * m, Rp = (int)cdiv(R, 32) * 32;
  PmaHeadArgs a{};
  a.Tp = v.Tp; a.Mp = v.Mp; a.Lp = v.Lp; a.S = mab0_splits(s); a.T = v.T; a.LSE = v.LSE;
  a.Qp = v.Qp; a.WvT = v.WvT; a.bv = p.bv; a.WoT = v.WoT; a.bo = p.bo;
  a.m = m; a.d = d; a.dk = dk; a.h = h; a.H = P; a.Osave = v.O; a.Zsave = v.Z;
  a.Wc = Wc; a.bc = bc; a.labels = labels; a.B = s.B; a.C = C; a.grad_scale = grad_scale;
  a.logits = logits; a.dlogits = dlogits; a.dP = dP; a.lossv = cls_ws; a.corrv = cls_ws + s.B;
  a.Wo = p.wo; a.Wv = p.wv; a.Rp = Rp; a.dZ = w.dZ; a.dO = w.dO; a.Th = w.Th; a.dTf = nullptr;
  a.dTb = w.dTb; a.dTt = w.dTt; a.Delta = w.Delta; a.LSEp = w.LSEp;
  a.zero_ptr = w.DG; a.zero_n = Rp * dk;
  *out = a;
  defer->cls = ClsWgradArgs{dlogits, P, a.lossv, a.corrv, s.B, d, C, dWc, dbc, loss_out, stats};
  defer->has_cls = 1;
  return PCA_OK;
}

int pma_head_launch(const pca_mab_shape& s, const pca_mab_params& p, void* saved, void* ws_bwd,
                    float* P, const float* Wc, const float* bc, const int64_t* labels, int C,
                    float grad_scale, float* logits, float* dlogits, float* dP, float* dWc,
                    float* dbc, float* loss_out, float* stats, float* cls_ws, BwdDefer* defer,
                    hipStream_t st) {
  PmaHeadArgs a{};
  PCA_TRY(pma_head_args(s, p, saved, ws_bwd, P, Wc, bc, labels, C, grad_scale, logits, dlogits, dP, dWc, dbc,
                        loss_out, stats, cls_ws, defer, &a));
  const int d = s.d, m = s.nq, h = s.h, dk = s.dk, R = h * m, Rp = a.Rp;
  size_t lds = ((size_t)R * dk + (size_t)m * d) * sizeof(float);
  const size_t l2 = (size_t)(d + C) * sizeof(float);
  const size_t l3 = (2 * (size_t)m * d + (size_t)Rp + (size_t)d) * sizeof(float);
  lds = lds > l2 ? lds : l2;
  lds = lds > l3 ? lds : l3;
  constexpr bool v1 = false;       // (the generic k_pma_head serves the shapes k_pma_head1 does not)
  if (!v1 && d == 128 && dk == 128 && h == 4 && m == 1 && C <= 64 && a.S >= 1 && a.S <= 8)
    hipLaunchKernelGGL(k_pma_head1, dim3(s.B), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL(k_pma_head, dim3(s.B), dim3(256), lds, st, a);
  return check_launch("k_pma_head");
}

int terminal_launch(const BwdDefer& D, hipStream_t st, const SlabSumJobs* late_in) {
  SlabSumJobs late{};
  if (late_in != nullptr) late = *late_in;
  if (!D.has_cls && !D.has_sw && late.n == 0) return mab0_post_launch(D.posts, st);
  const Mab0PostJobs& J = D.posts;
  int n1 = 0;
  for (int i = 0; i < J.n; ++i) {
    const int e1 = J.j[i].d * J.j[i].dk + J.j[i].m * J.j[i].d;
    n1 = e1 > n1 ? e1 : n1;
  }
  int gx = (int)cdiv(n1, 256);
  if (D.has_cls && D.cls.C > gx) gx = D.cls.C;
  if (D.has_sw) {
    const int gs = (int)cdiv(D.sw.M, D.sw.rows_per_wg);
    gx = gs > gx ? gs : gx;
  }
  for (int i = 0; i < late.n; ++i) {
    PCA_REQUIRE(slab_sum_job_ok(late.j[i]), "terminal: rider alignment");
    const int need = (int)cdiv(late.j[i].n, 256);
    gx = need > gx ? need : gx;
  }
  hipStream_t ts = terminal_stream(st);
  hipLaunchKernelGGL(k_terminal1,
                     dim3(gx, J.n + (D.has_cls ? 1 : 0) + (D.has_sw ? 1 : 0) + late.n),
                     dim3(256), 0, ts, J, D.cls, D.has_cls ? 1 : 0, D.sw, D.has_sw ? 1 : 0, late);
  PCA_TRY(check_launch("k_terminal1"));
  // the layer-1 fc_v partials k_terminal1 wrote (slab mode) are summed by rider rows of post 2
  SlabSumJobs late2{};
  if (D.has_sw && D.sw.slab != nullptr) {
    const int nwg = (int)cdiv(D.sw.M, D.sw.rows_per_wg), n1 = 128 * D.sw.dq, stride = n1 + 128;
    late2.j[late2.n++] = SlabSumJob{D.sw.slab, D.sw.dW, nwg, n1, 1, stride};
    if (D.sw.db != nullptr) late2.j[late2.n++] = SlabSumJob{D.sw.slab + n1, D.sw.db, nwg, 128, 1, stride};
  }
  if (J.n == 0) return slab_sum_jobs(late2, ts);
  int n2 = 0;
  for (int i = 0; i < J.n; ++i) {
    const Mab0PostJob& a = J.j[i];
    const int e2 = a.d * a.dq + a.d + (a.dI ? a.m * a.dq : 0);
    n2 = e2 > n2 ? e2 : n2;
  }
  for (int i = 0; i < late2.n; ++i) n2 = late2.j[i].n > n2 ? late2.j[i].n : n2;
  hipLaunchKernelGGL(k_mab0_post2, dim3((unsigned)cdiv(n2, 256), J.n + late2.n), dim3(256), 0, ts, J,
                     late2);
  return check_launch("k_mab0_post2");
}

int mab0_post_launch(const Mab0PostJobs& J, hipStream_t st) {
  if (J.n == 0) return PCA_OK;
  int n1 = 0, n2 = 0;
  for (int i = 0; i < J.n; ++i) {
    const Mab0PostJob& a = J.j[i];
    const int e1 = a.d * a.dk + a.m * a.d, e2 = a.d * a.dq + a.d + (a.dI ? a.m * a.dq : 0);
    n1 = e1 > n1 ? e1 : n1;
    n2 = e2 > n2 ? e2 : n2;
  }
  hipStream_t ts = terminal_stream(st);     // off the critical path
  hipLaunchKernelGGL(k_mab0_post1, dim3((unsigned)cdiv(n1, 256), J.n), dim3(256), 0, ts, J);
  PCA_TRY(check_launch("k_mab0_post1"));
  hipLaunchKernelGGL(k_mab0_post2, dim3((unsigned)cdiv(n2, 256), J.n), dim3(256), 0, ts, J,
                     SlabSumJobs{});
  return check_launch("k_mab0_post2");
}

int mab0_bwd_small_launch(const float* X, const float* Gf, const float* dTf, const float* LSE,
                          const float* Delta, int B, int N, int R, int Rp, int dk, float* DG,
                          const int32_t* lengths, hipStream_t st, float* slabs) {
  PCA_REQUIRE(R == 64 || R == 128 || R == 256, "mab0_bwd_small: %d score rows", R);
  hipLaunchKernelGGL(k_mab0_bwd_small, dim3(B, small_row_split(B, R)), dim3(256), 0, st, X, Gf, dTf,
                     LSE, Delta, N, R, Rp, dk, DG, lengths, slabs);
  return check_launch("k_mab0_bwd_small");
}

// point ranges per set of k_mab0_bwd (96+ KiB of LDS: one workgroup per CU)
int mab0_bwd_splits(const pca_mab_shape& s) {
  int S = mab0_splits(s);
  while (S > 1 && s.B * S > 256) S /= 2;
  return S;
}
size_t mab0_carve_bwd_ws(const pca_mab_shape& s, Mab0BwdWs* out, void* base) {
  Carver c(base);
  Mab0BwdWs w;
  const int R = s.h * s.nq, Rp = (int)cdiv(R, 32) * 32;
  const size_t Bm = (size_t)s.B * s.nq;
  w.dZ = c.take<float>(Bm * s.d);
  w.dO = c.take<float>(Bm * s.d);
  w.Th = c.take<float>((size_t)s.B * R * s.dk);
  w.dTf = c.take<float>((size_t)s.B * R * s.dk);
  w.Delta = c.take<float>((size_t)s.B * Rp);
  w.LSEp = c.take<float>((size_t)s.B * Rp);
  w.DG = c.take<float>((size_t)Rp * s.dk);
  w.dQs = c.take<float>((size_t)s.nq * s.d);
  w.dQp = c.take<float>((size_t)s.nq * s.d);
  w.dTb = c.take<__bf16>((size_t)s.B * Rp * s.dk);
  w.dTt = c.take<__bf16>((size_t)s.B * Rp * s.dk);
  w.GtP = c.take<__bf16>((size_t)Rp * s.dk);
  w.slabs = c.take<float>(s.dk <= 4 ? (size_t)s.B * R * s.dk
                                     : (size_t)s.B * mab0_bwd_splits(s) * R * s.dk);
  if (out) *out = w;
  return c.off;
}

size_t mab0_bf16_bwd_ws_bytes(const pca_mab_shape& s) {
  if (s.d == 256) return mab0_d256_bwd_ws_bytes(s);
  return mab0_carve_bwd_ws(s, nullptr, nullptr);
}

// dQ -> dI [m, dq] (ACCUMULATED, may be null), dK -> dX [B, N, dk] (written or accumulated)
int mab0_bf16_bwd(const pca_mab_shape& s, const float* I, const void* X,
                  const pca_mab_params& p, const void* saved, const float* dH, float* dI,
                  void* dX, int dk_accumulate, const pca_mab_grads& gr, void* ws,
                  hipStream_t st) {
  return mab0_bf16_bwd_ex(s, I, X, p, saved, dH, dI, dX, dk_accumulate, gr, ws, 0, st);
}
int mab0_bf16_bwd_ex(const pca_mab_shape& s, const float* I, const void* X,
                     const pca_mab_params& p, const void* saved, const float* dH, float* dI,
                     void* dX, int dk_accumulate, const pca_mab_grads& gr, void* ws, int flags,
                     hipStream_t st, BwdDefer* defer) {
  if (s.d == 256)
    return mab0_d256_bwd(s, I, X, p, saved, dH, dI, dX, dk_accumulate, gr, ws, st, defer);
  Mab0Saved v;
  mab0_carve_saved(s, &v, const_cast<void*>(saved));
  Mab0BwdWs w;
  mab0_carve_bwd_ws(s, &w, ws);
  const bool head_done = (flags & PCA_F_SKIP_HEAD) != 0;
  const int d = s.d, m = s.nq, h = s.h, dk = s.dk, R = h * m, Rp = (int)cdiv(R, 32) * 32;
  const int64_t Bm = (int64_t)s.B * m;
  const bool small = dk <= 4;
  const float sl2e = 1.4426950408889634f / sqrtf((float)d);
  if (small && dX != nullptr) {
    set_error("mab0_bf16_bwd: dK for dk <= 4 is not built (the set is the model input)");
    return PCA_EUNSUPPORTED;
  }

  const size_t el = (2 * (size_t)m * d + (size_t)Rp + (size_t)d) * sizeof(float);   // + split scratch
  if (head_done) {
    // dZ, dO, Th, dT images, Delta, LSEp and dQs come from k_mid_bwd
  } else if (m > 2)
    hipLaunchKernelGGL((k_mab0_epi_bwd<8>), dim3(s.B), dim3(256), el, st, dH, v.Z, v.T, v.LSE, p.wo,
                       p.wv, m, d, dk, h, Rp, w.dZ, w.dO, w.Th, small ? w.dTf : nullptr,
                       small ? nullptr : w.dTb, small ? nullptr : w.dTt, w.Delta, w.LSEp, s.B,
                       w.DG, Rp * dk);
  else
    hipLaunchKernelGGL((k_mab0_epi_bwd<1>), dim3(s.B), dim3(256), el, st, dH, v.Z, v.T, v.LSE, p.wo,
                       p.wv, m, d, dk, h, Rp, w.dZ, w.dO, w.Th, small ? w.dTf : nullptr,
                       small ? nullptr : w.dTb, small ? nullptr : w.dTt, w.Delta, w.LSEp, s.B,
                       w.DG, Rp * dk);
  PCA_TRY(check_launch("k_mab0_epi_bwd"));
  // (with head_done the caller's k_mid_bwd has cleared DG)

  if (small) {
    // slab mode: per-set partials [B][R][dk] instead of atomics, summed in a fixed order
    float* sl = (wgrad_slabs_on() && (R * dk) % 4 == 0) ? w.slabs : nullptr;
    hipLaunchKernelGGL(k_mab0_bwd_small, dim3(s.B, (R % 64 == 0 && R <= 512) ? 2 : 1), dim3(256), 0, st,
                       reinterpret_cast<const float*>(X), v.Gf, w.dTf, v.LSE,
                       w.Delta, s.nk, R, Rp, dk, w.DG, s.k_lengths, sl);
    PCA_TRY(check_launch("k_mab0_bwd_small"));
    if (sl != nullptr) {
      // (only the first R * dk floats of the [Rp][dk] block are read by the post stage)
      const SlabSumJob sj{sl, w.DG, s.B, R * dk, 0, 0};
      if (defer != nullptr) {
        PCA_REQUIRE(defer->sums.n < 40, "mab0_bf16_bwd: slab-sum table full");
        defer->sums.j[defer->sums.n++] = sj;
      } else {
        SlabSumJobs one{};
        one.j[one.n++] = sj;
        PCA_TRY(slab_sum_jobs(one, st));
      }
    }
  } else {
    const int S = mab0_bwd_splits(s);
    Mab0BwdArgs a{X, v.Gb, v.GtP, w.dTb, w.dTt, w.LSEp, w.Delta, dX, w.DG, s.B, s.nk,
                  dk_accumulate ? 1 : 0, S, s.k_lengths, R, w.slabs};
    size_t lds = 2 * (size_t)Rp * 256 + 2 * (size_t)128 * Rp * 2 + 2 * 4 * 32 * 256 +
                 2 * Rp * sizeof(float);
    if (lds < (size_t)4 * Rp * 128 * 4) lds = (size_t)4 * Rp * 128 * 4;     // merge slabs
    static std::once_flag once;
    std::call_once(once, [] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_mab0_bwd<64, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_mab0_bwd<64, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_mab0_bwd<32, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_mab0_bwd<32, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    const double pts = (double)s.B * s.nk;
    const bool abf = s.k_dtype == PCA_BF16;
    // algorithmic bytes: X in, dX out (read as well when it accumulates onto mab1's dQ)
    const double eb0 = abf ? 2.0 : 4.0;
    ProfScope ps(PCA_K_MAB0_BWD, st, 4.0 * pts * (2.0 * dk * d + 2.0 * m * d),
                 pts * eb0 * dk * (1.0 + (dX != nullptr ? (dk_accumulate ? 2.0 : 1.0) : 0.0)));
    const dim3 grid(s.B, S);
    if (Rp == 32 && abf) hipLaunchKernelGGL((k_mab0_bwd<32, true>), grid, dim3(256), lds, st, a);
    else if (Rp == 32) hipLaunchKernelGGL((k_mab0_bwd<32, false>), grid, dim3(256), lds, st, a);
    else if (abf) hipLaunchKernelGGL((k_mab0_bwd<64, true>), grid, dim3(256), lds, st, a);
    else hipLaunchKernelGGL((k_mab0_bwd<64, false>), grid, dim3(256), lds, st, a);
    ps.end();
    PCA_TRY(check_launch("k_mab0_bwd"));
    // dG = the workgroups' slabs added in a fixed order: with the other sums of the step when
    // the caller defers them, else right away
    const SlabSumJob sj{w.slabs, w.DG, s.B * S, R * dk, 0};
    if (defer != nullptr) {
      PCA_REQUIRE(defer->sums.n < 40, "mab0_bf16_bwd: slab-sum table full");
      defer->sums.j[defer->sums.n++] = sj;
    } else {
      SlabSumJobs one{};
      one.j[one.n++] = sj;
      PCA_TRY(slab_sum_jobs(one, st));
    }
  }

  // ---- parameter gradients of the epilogue: [B*m]-row reductions, ONE MFMA launch ----
  if (!(flags & PCA_F_SKIP_WGRAD)) {
    WgradJobs jobs{};
    jobs.j[0] = WgradJob{w.dZ, v.O, gr.wo, gr.bo, Bm, 0, 128};
    jobs.n = 1;
    if (!small) {
      const int dh = d / h;
      for (int j = 0; j < h; ++j)          // dWv rows of head j  <-  dO^T . T_j
        jobs.j[jobs.n++] = WgradJob{w.dO, w.Th + (int64_t)j * Bm * dk, gr.wv,
                                    j == 0 ? gr.bv : nullptr, Bm, j * dh, (j + 1) * dh};
    }
    hipStream_t ts = terminal_stream(st);
    PCA_TRY(wgrad128_defer(defer, jobs, false, 64, ts));
    if (small)
      PCA_TRY(wgrad_small_f32_launch(w.dO, w.Th, Bm, dk, (int64_t)Bm * dk, gr.wv, gr.bv, ts));
  }
  // (the sum of dO over the sets is taken inside k_mab0_post1)
  // k_mid_bwd / k_mab0_epi_bwd left dO per set, the post kernel sums it
  Mab0PostJob pj{nullptr, w.DG, v.Qp, p.wk, I, p.wq, gr.wk, w.dQp, gr.wq,
                 gr.bq, dI, m, d, dk, s.dq, h, sl2e, w.dO, s.B};
  if (defer != nullptr) {
    PCA_REQUIRE(defer->posts.n < 3, "mab0_bf16_bwd: post-job table full");
    defer->posts.j[defer->posts.n++] = pj;
    return PCA_OK;
  }
  Mab0PostJobs one{};
  one.j[one.n++] = pj;
  return mab0_post_launch(one, st);
}

}  // namespace pca
