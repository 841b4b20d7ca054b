// Single-launch forward of the many-queries block at d = 128 / 4 heads / m = 16 keys (ISAB's
// mab1(X, H), set_transformer-master/modules.py:53 with :19-33 inside) - the WAVE = HEAD, WEIGHTS
// IN REGISTERS layout of k_isab1_fwd256 (d256_fused.hip) at the cfg2 / cfg3 shape:
//     Qp = fc_q(x) ; per head A = softmax(Qp_h Kp_h^T / sqrt d) ; O = Qp + A Vp ; Y = O + relu(fc_o(O))
// A workgroup is 4 waves; wave j owns head j and keeps its [32 x 128] slices of Wq and Wo as MFMA A
// operands in 32 VGPRs.  The whole kernel needs ~100 registers and 66 KiB of LDS, so two
// workgroups (eight waves) share a CU where k_mab1_fwd<128, ...> - 100 KiB of weight images in LDS,
// every wave computing all 128 features of its 32 points - runs one unit per wave: at cfg2 (65 536
// points = 2048 tiles) a workgroup here walks 4 tiles with the next one streaming in by LDS-DMA.
// Saved tensors (Qp, O, ReLU mask words) keep the layouts k_mab1_bwd<128, ...> reads.
// Measured at cfg2 (B = 128, N = 512): 20.2 us against 22.8 for the d -> d block.  The floor is not
// latency but the SAVED tensors: a training forward reads X and writes Y, O, Qp - four [65 536 x 128]
// bf16 tensors = 67 MB, 15 us at the 4.5 TB/s these streams reach - so a deeper X ring (tried: one
// to three tiles ahead, same time) buys nothing; only saving less would.
// Roofline unit: 2 (dq d + d^2 + 2 m d) FLOP per point, 2 (dq + d) bytes per point.
#include "d256_bf16.hpp"

#include <math.h>

#include <mutex>

namespace pca {

namespace {

constexpr int P = 32, NBK = P / 16;
constexpr int XB = 4, PD = XB - 1;          // X ring, tiles ahead

struct Fused128Args {
  const void* X;          // [B*N][D] bf16, or fp32 [B*N][dq] for dq <= 4
  const __bf16* Wq;       // [D][D] bf16, natural (nullptr for dq <= 4)
  const float* WqF;       // dq <= 4: fp32 [D][dq]
  const float* bq;
  const __bf16* KpP;      // [B][MI][D] (K-permuted features inside each head)
  const __bf16* Vt;       // [B][D][MI] (keys in natural order for MI = 16)
  const __bf16* WoP;      // [D][D] bf16, K-PERMUTED inside each block of 32 (the image the engine holds)
  const float* bo;
  __bf16* Y;              // [B*N][D]
  __bf16 *QpS, *OS;       // saved for the backward (nullable)
  uint32_t* mask;         // ReLU mask bits in the layout of mab1_mask_index<D> (nullable)
  int B, N, dq, tiles_per_set, units_per_wg;
  float scale_log2e;
};

typedef __attribute__((address_space(3))) void lptr_t;

#define PCA_WAIT_VM_CASE(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
__device__ __forceinline__ void wait_vm128(int n) {
  switch (n) {
    PCA_WAIT_VM_CASE(1) PCA_WAIT_VM_CASE(2) PCA_WAIT_VM_CASE(3) PCA_WAIT_VM_CASE(4)
    PCA_WAIT_VM_CASE(5) PCA_WAIT_VM_CASE(6) PCA_WAIT_VM_CASE(7) PCA_WAIT_VM_CASE(8)
    PCA_WAIT_VM_CASE(9) PCA_WAIT_VM_CASE(10) PCA_WAIT_VM_CASE(11) PCA_WAIT_VM_CASE(12)
    PCA_WAIT_VM_CASE(13) PCA_WAIT_VM_CASE(14) PCA_WAIT_VM_CASE(15) PCA_WAIT_VM_CASE(16)
    PCA_WAIT_VM_CASE(17) PCA_WAIT_VM_CASE(18) PCA_WAIT_VM_CASE(19) PCA_WAIT_VM_CASE(20)
    PCA_WAIT_VM_CASE(21) PCA_WAIT_VM_CASE(22) PCA_WAIT_VM_CASE(23) PCA_WAIT_VM_CASE(24)
    PCA_WAIT_VM_CASE(25) PCA_WAIT_VM_CASE(26) PCA_WAIT_VM_CASE(27) PCA_WAIT_VM_CASE(28)
    default:
      if (n > 28) asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}

template <int D, int MI, bool SMALL>
__global__ __launch_bounds__(64 * (D / 32), 2) void k_isab1_fwd_t(const Fused128Args a) {
  static_assert(D == 128 && MI == 16, "built for the cfg2 / cfg3 shape");
  constexpr int NW = D / 32, NT = 64 * NW, KS = D / 32, CPR = D / 8;      // CPR: 16-byte chunks per row
  constexpr int ROWB = D * 2, TILEB = P * ROWB, NB_OFF = 16 * ROWB;       // 8 KiB tiles
  constexpr int DMA_PER_WAVE = TILEB / 1024 / NW;
  constexpr int MW = D / 128;                                             // mask words per 16-point block
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // X: a ring of XB tiles filled by LDS-DMA XB - 1 tiles ahead - a tile is ~1 us of work here, less
  // than a memory round trip, so one tile ahead leaves the loop waiting for its input every time;
  // O / Qp: one tile each, stored right after barrier B1 of their own iteration; Y / mask: by
  // iteration parity, stored after barrier B0 of the next iteration (two barriers per tile)
  char* sXb = smem;                         // X tiles          [XB][TILEB]
  char* sOb = smem + XB * TILEB;            // O tile
  char* sQb = sOb + TILEB;                  // Qp tile (training: saved for the backward)
  char* sYb = sQb + TILEB;                  // Y tiles          [2][TILEB]
  uint32_t* sMaskb = reinterpret_cast<uint32_t*>(sYb + 2 * TILEB);        // [2][NBK][MW][64]
  float* sBias = reinterpret_cast<float*>(sMaskb + 2 * NBK * MW * 64);    // bq [D], bo [D]
  const int tid = threadIdx.x, lane = tid & 63;
  const int j = __builtin_amdgcn_readfirstlane(tid >> 6);                 // head of this wave (scalar)
  const int r = lane & 15, g = lane >> 4;
  auto toff = [](int row, int c16) { return swz(row, c16, ROWB); };

  // ---- this head's weight slices: A operands [row = feature 32 j + 16 t + r][k-slots 32 s + 8 g ..] ----
  bf16x8 wqa[SMALL ? 1 : KS][2], woa[KS][2];
#pragma unroll
  for (int s = 0; s < KS; ++s)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int64_t o = (int64_t)(32 * j + 16 * t + r) * D + 32 * s + 8 * g;
      woa[s][t] = *reinterpret_cast<const bf16x8*>(a.WoP + o);
      if (!SMALL) wqa[s][t] = *reinterpret_cast<const bf16x8*>(a.Wq + o);
    }
  float wqs[SMALL ? 2 : 1][4][4];
  if (tid < D) {
    sBias[tid] = a.bq[tid];
    sBias[D + tid] = a.bo[tid];
  }
  auto bias4 = [&](int which, int t) {
    const float4 q4 = *reinterpret_cast<const float4*>(sBias + which * D + 32 * j + 16 * t + 4 * g);
    return f32x4{q4.x, q4.y, q4.z, q4.w};
  };
  // workgroup barrier that orders LDS traffic only (no vmcnt(0): see k_isab1_fwd256)
  auto lds_barrier = [] {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    if (SMALL) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int c = 0; c < 4; ++c)
          wqs[SMALL ? t : 0][e][c] = c < a.dq ? a.WqF[(32 * j + 16 * t + 4 * g + e) * a.dq + c] : 0.f;
    }
  }

  const int total_units = a.B * a.tiles_per_set;
  const int u0 = blockIdx.x * a.units_per_wg;
  const int u1 = (u0 + a.units_per_wg < total_units) ? u0 + a.units_per_wg : total_units;
  // LDS-DMA of a unit's X tile (linear in LDS, swizzled at the source; issued from inline asm so that
  // hipcc does not put vmcnt(0) in front of every later ds_read: see k_isab1_fwd256)
  auto dma_x = [&](int unit, char* dst) {
    const int b = unit / a.tiles_per_set, n0 = (unit - b * a.tiles_per_set) * P;
#pragma unroll
    for (int i = 0; i < DMA_PER_WAVE; ++i) {
      const int p = (DMA_PER_WAVE * j + i) * 64 + lane;
      const int row = p / CPR, slot = p % CPR;
      const int ch = (slot & ~15) | ((slot ^ row) & 15);
      const int n = n0 + row < a.N ? n0 + row : a.N - 1;
      const __bf16* src = reinterpret_cast<const __bf16*>(a.X) + ((int64_t)b * a.N + n) * D + ch * 8;
      const unsigned ldst = __builtin_amdgcn_readfirstlane(
          (unsigned)(uintptr_t)(lptr_t*)(dst + (DMA_PER_WAVE * j + i) * 1024));
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                   "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(src), "s"(ldst) : "memory");
    }
  };
  if (!SMALL) {
#pragma unroll 1
    for (int k = 0; k < PD && u0 + k < u1; ++k) dma_x(u0 + k, sXb + k * TILEB);
  }

  // per-lane byte offsets into a tile, computed once:
  //   oB[s]   : natural B-operand fragment (16 bytes), row r, chunk 4 s + g          (GEMM1 over X)
  //   oD[t]   : accumulator-layout 8 bytes, row r, features 32 j + 16 t + 4 g
  //   oP[s][h]: K-PERMUTED B-operand halves (8 bytes each) of k-block s: features 32 s + 16 h + 4 g
  //             (GEMM2 over O against the K-permuted Wo image: k-slot 8 g + e <-> feature 4 g + e,
  //             e < 4, and 16 + 4 g + e - 4 otherwise)
  //   oC      : coalesced 16-byte piece of row tid / CPR, chunk tid % CPR
  int oB[KS], oD[2], oP[KS][2];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    oB[s] = toff(r, 4 * s + g);
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) oP[s][hh] = toff(r, 4 * s + 2 * hh + (g >> 1)) + 8 * (g & 1);
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) oD[t] = toff(r, 4 * j + 2 * t + (g >> 1)) + 8 * (g & 1);
  const int oC = toff(tid / CPR, tid % CPR);

  // coalesced stores of finished tiles from their LDS images
  auto store_y = [&](int unit, int par) {          // Y and the ReLU mask words of tile `unit`
    const int b = unit / a.tiles_per_set, tile = unit - b * a.tiles_per_set;
    const int n0 = tile * P, nlive = a.N - n0;
    const int64_t rowbase = (int64_t)b * a.N + n0;
#pragma unroll
    for (int i = 0; i < P * CPR / NT; ++i) {
      const int c = tid + NT * i, row = c / CPR, ch = c % CPR;
      if (row < nlive)
        *reinterpret_cast<uint4*>(a.Y + (rowbase + row) * D + ch * 8) =
            *reinterpret_cast<const uint4*>(sYb + par * TILEB + oC + NB_OFF * i);
    }
    if (a.mask != nullptr && tid < NBK * MW * 64) {
      const int nb = tid / (MW * 64), w = (tid / 64) % MW;
      const int tiles128 = (a.tiles_per_set * P + 127) / 128;
      const int64_t blk = (int64_t)b * tiles128 * 8 + tile * NBK + nb;
      a.mask[(blk * MW + w) * 64 + lane] = sMaskb[par * NBK * MW * 64 + (nb * MW + w) * 64 + lane];
    }
  };
  auto store_oq = [&](int unit) {                  // training: the saved O and Qp of tile `unit`
    const int b = unit / a.tiles_per_set, tile = unit - b * a.tiles_per_set;
    const int n0 = tile * P, nlive = a.N - n0;
    const int64_t rowbase = (int64_t)b * a.N + n0;
#pragma unroll
    for (int i = 0; i < P * CPR / NT; ++i) {
      const int c = tid + NT * i, row = c / CPR, ch = c % CPR;
      if (row < nlive) {
        if (a.OS != nullptr)
          *reinterpret_cast<uint4*>(a.OS + (rowbase + row) * D + ch * 8) =
              *reinterpret_cast<const uint4*>(sOb + oC + NB_OFF * i);
        if (a.QpS != nullptr)
          *reinterpret_cast<uint4*>(a.QpS + (rowbase + row) * D + ch * 8) =
              *reinterpret_cast<const uint4*>(sQb + oC + NB_OFF * i);
      }
    }
  };
  // VM instructions per thread of the two store passes of a FULL tile (a ragged tile - the last
  // of a set when 32 does not divide N - issues fewer: the waits that would count it drain instead)
  const bool ragged = a.N % P != 0;
  auto full_tile = [&](int unit) {
    return !ragged || unit % a.tiles_per_set != a.tiles_per_set - 1;
  };
  const int n_y = (P * CPR / NT) + ((a.mask != nullptr && j < NBK * MW) ? 1 : 0);
  const int n_oq = (P * CPR / NT) * ((a.OS != nullptr ? 1 : 0) + (a.QpS != nullptr ? 1 : 0));

  // the set's K / V operands of this wave and (layer 1) the first tile's points: fetched before the
  // first barrier instead of at their first use
  auto load_xv = [&](int unit, float (&dst)[NBK][4]) {
    const int b = unit / a.tiles_per_set, tile = unit - b * a.tiles_per_set;
    const int nlive = a.N - tile * P;
    const int64_t rowbase = (int64_t)b * a.N + tile * P;
#pragma unroll
    for (int nb = 0; nb < NBK; ++nb) {
      const int n = 16 * nb + r;
      const int64_t rr = rowbase + (n < nlive ? n : nlive - 1);
#pragma unroll
      for (int c = 0; c < 4; ++c)
        dst[nb][c] = c < a.dq ? reinterpret_cast<const float*>(a.X)[rr * a.dq + (c < a.dq ? c : 0)] : 0.f;
    }
  };
  float xnext[NBK][4];
  if (SMALL && u0 < u1) load_xv(u0, xnext);
  int cur_b = -1;
  bf16x8 kpa;              // the head's 16 keys: A operand [key r][k-slots 8 g .. of the head's 32 features]
  bf16x4 vta[2];           // V^T: A operand [feature 16 t + r][keys 4 g ..]
  for (int unit = u0; unit < u1; ++unit) {
    const int k = unit - u0, par = k & 1;
    char* sXc = sXb + (k % XB) * TILEB;
    char* sO = sOb;
    char* sQ = sQb;
    char* sY = sYb + par * TILEB;
    uint32_t* sMask = sMaskb + par * NBK * MW * 64;
    const int b = unit / a.tiles_per_set;
    float xv[SMALL ? NBK : 1][4];
    if (SMALL) {               // layer 1: this tile's points were fetched one iteration ago
#pragma unroll
      for (int nb = 0; nb < NBK; ++nb)
#pragma unroll
        for (int c = 0; c < 4; ++c) xv[SMALL ? nb : 0][c] = xnext[nb][c];
      if (unit + 1 < u1) load_xv(unit + 1, xnext);
    }
    // Tile k + PD starts to stream in (the last readers of its buffer passed barrier B1 of the
    // previous iteration).  Tile k's DMA, issued PD iterations ago, must have landed; younger in
    // the queue are the DMAs of tiles k + 1 .. k + PD and, of each of the last PD iterations i, the
    // Y / mask stores of tile i - 1 and the O / Qp stores of tile i.
    if (!SMALL) {
      if (unit + PD < u1) dma_x(unit + PD, sXb + ((k + PD) % XB) * TILEB);
      const int ahead = (u1 - 1 - unit) < PD ? (u1 - 1 - unit) : PD;
      int younger = ahead * DMA_PER_WAVE;
#pragma unroll
      for (int i = 1; i <= PD; ++i) {
        if (k - i >= 0) younger += full_tile(unit - i) ? n_oq : -1000;
        if (k - i - 1 >= 0) younger += full_tile(unit - i - 1) ? n_y : -1000;
      }
      wait_vm128(younger);
    }
    lds_barrier();                       // B0: X tile (all waves' pieces); previous tile's Y / mask
    if (unit > u0) store_y(unit - 1, par ^ 1);
    if (b != cur_b) {
      kpa = *reinterpret_cast<const bf16x8*>(a.KpP + ((int64_t)b * MI + r) * D + 32 * j + 8 * g);
#pragma unroll
      for (int t = 0; t < 2; ++t)
        vta[t] = *reinterpret_cast<const bf16x4*>(a.Vt + ((int64_t)b * D + 32 * j + 16 * t + r) * MI +
                                                  4 * g);
      cur_b = b;
    }
    // ---- GEMM1: Qp_h^T[f][pt] = Wq_h . X^T + bq ----
    f32x4 acc[2][NBK];
    if (SMALL) {
      const f32x4 bqv[2] = {bias4(0, 0), bias4(0, 1)};
#pragma unroll
      for (int nb = 0; nb < NBK; ++nb)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc[t][nb][e] = bqv[t][e] + wqs[SMALL ? t : 0][e][0] * xv[SMALL ? nb : 0][0] +
                            wqs[SMALL ? t : 0][e][1] * xv[SMALL ? nb : 0][1] +
                            wqs[SMALL ? t : 0][e][2] * xv[SMALL ? nb : 0][2] +
                            wqs[SMALL ? t : 0][e][3] * xv[SMALL ? nb : 0][3];
    } else {
      const f32x4 bq0 = bias4(0, 0), bq1 = bias4(0, 1);
#pragma unroll
      for (int nb = 0; nb < NBK; ++nb) {
        acc[0][nb] = bq0;
        acc[1][nb] = bq1;
      }
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int nb = 0; nb < NBK; ++nb) {
          const bf16x8 bx = *reinterpret_cast<const bf16x8*>(sXc + oB[s] + NB_OFF * nb);
          acc[0][nb] = mfma32(wqa[SMALL ? 0 : s][0], bx, acc[0][nb]);
          acc[1][nb] = mfma32(wqa[SMALL ? 0 : s][1], bx, acc[1][nb]);
        }
    }
    if (a.QpS != nullptr) {              // own 32-column slice of the Qp tile
#pragma unroll
      for (int nb = 0; nb < NBK; ++nb)
#pragma unroll
        for (int t = 0; t < 2; ++t)
          *reinterpret_cast<bf16x4*>(sQ + oD[t] + NB_OFF * nb) = pack4(acc[t][nb]);
    }
    // ---- attention over the 16 inducing keys, all inside the wave ----
#pragma unroll
    for (int nb = 0; nb < NBK; ++nb) {
      const bf16x8 qb = pack8(acc[0][nb], acc[1][nb]);
      const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
      f32x4 s0 = mfma32(kpa, qb, z4);            // [key 4 g + e][point r]
      float mx = fmaxf(fmaxf(s0[0], s0[1]), fmaxf(s0[2], s0[3]));
      mx = wave16_max(mx);
      float sum = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s0[e] = __builtin_amdgcn_exp2f((s0[e] - mx) * a.scale_log2e);
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
    // ---- own slice of the O tile ----
#pragma unroll
    for (int nb = 0; nb < NBK; ++nb)
#pragma unroll
      for (int t = 0; t < 2; ++t)
        *reinterpret_cast<bf16x4*>(sO + oD[t] + NB_OFF * nb) = pack4(acc[t][nb]);
    lds_barrier();                       // B1: O (and Qp) tiles complete; X tile consumed
    if (n_oq != 0) store_oq(unit);
    // ---- GEMM2: Z_h^T = Wo_h . O^T + bo ; Y_h = O_h + relu(Z_h) ----
    {
      const f32x4 bo0 = bias4(1, 0), bo1 = bias4(1, 1);
#pragma unroll
      for (int nb = 0; nb < NBK; ++nb) {
        acc[0][nb] = bo0;
        acc[1][nb] = bo1;
      }
    }
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int nb = 0; nb < NBK; ++nb) {
        const bf16x4 lo = *reinterpret_cast<const bf16x4*>(sO + oP[s][0] + NB_OFF * nb);
        const bf16x4 hi = *reinterpret_cast<const bf16x4*>(sO + oP[s][1] + NB_OFF * nb);
        bf16x8 ob;
#pragma unroll
        for (int e = 0; e < 4; ++e) { ob[e] = lo[e]; ob[4 + e] = hi[e]; }
        acc[0][nb] = mfma32(woa[s][0], ob, acc[0][nb]);
        acc[1][nb] = mfma32(woa[s][1], ob, acc[1][nb]);
      }
#pragma unroll
    for (int nb = 0; nb < NBK; ++nb) {
      uint32_t bits = 0u;                // byte (j & 3) of mask word j >> 2: feature tiles 2 j, 2 j + 1
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const bf16x4 o4 = *reinterpret_cast<const bf16x4*>(sO + oD[t] + NB_OFF * nb);
        f32x4 y;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float zz = acc[t][nb][e];
          y[e] = (float)o4[e] + fmaxf(zz, 0.f);
          if (zz > 0.f) bits |= 1u << (4 * t + e);
        }
        *reinterpret_cast<bf16x4*>(sY + oD[t] + NB_OFF * nb) = pack4(y);
      }
      if (a.mask != nullptr)
        reinterpret_cast<uint8_t*>(sMask)[((nb * MW + (j >> 2)) * 64 + lane) * 4 + (j & 3)] =
            (uint8_t)bits;
    }
  }
  if (u0 < u1) {
    lds_barrier();
    store_y(u1 - 1, (u1 - 1 - u0) & 1);
  }
}

}  // namespace

// X: bf16 [B*N][128] (dq = 128) or fp32 [B*N][dq] (dq <= 4); WqB natural bf16 image, WoP K-permuted
// bf16 image (prep_weight modes 0 / 1); outputs bf16.  QpS / OS / mask nullable (inference).
int isab1_fwd128_fused(const void* X, int dq, const __bf16* WqB, const float* WqF, const float* bq,
                       const __bf16* KpP, const __bf16* Vt, const __bf16* WoP, const float* bo,
                       __bf16* Y, __bf16* QpS, __bf16* OS, uint32_t* mask, int B, int N,
                       hipStream_t st) {
  constexpr int D = 128, MI = 16;
  Fused128Args a{};
  a.X = X; a.Wq = WqB; a.WqF = WqF; a.bq = bq; a.KpP = KpP; a.Vt = Vt; a.WoP = WoP; a.bo = bo;
  a.Y = Y; a.QpS = QpS; a.OS = OS; a.mask = mask;
  a.B = B; a.N = N; a.dq = dq;
  a.tiles_per_set = (int)cdiv(N, P);
  a.scale_log2e = 1.4426950408889634f / sqrtf((float)D);
  const int total = B * a.tiles_per_set;
  int grid = total < 512 ? total : 512;                  // two workgroups per CU
  a.units_per_wg = (int)cdiv(total, grid);
  grid = (int)cdiv(total, a.units_per_wg);
  const size_t lds = (XB + 4) * (size_t)(P * D * 2) + 2 * NBK * (D / 128) * 64 * sizeof(uint32_t) +
                     2 * D * sizeof(float);
  static std::once_flag once;
  std::call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_isab1_fwd_t<D, MI, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_isab1_fwd_t<D, MI, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  const double pts = (double)B * N;
  ProfScope ps(PCA_K_MAB1_FWD, st, 2.0 * pts * ((double)dq * D + (double)D * D + 2.0 * MI * D),
               pts * ((dq <= 4 ? 4.0 : 2.0) * dq + 2.0 * D));
  if (dq <= 4)
    hipLaunchKernelGGL((k_isab1_fwd_t<D, MI, true>), dim3(grid), dim3(64 * (D / 32)), lds, st, a);
  else
    hipLaunchKernelGGL((k_isab1_fwd_t<D, MI, false>), dim3(grid), dim3(64 * (D / 32)), lds, st, a);
  ps.end();
  return check_launch("k_isab1_fwd128");
}

}  // namespace pca
