// Training kernels of the d = 256 / 8-head / m = 32 Set Transformer (BASELINE configs[3], the
// north-star shape): the pieces of the MAB adjoint (set_transformer-master/
// modules.py:19-33 backwards) that the d = 128 kernels hold in ONE launch do not fit a CU at
// d = 256 (two 128 KiB weight images), so the backward of the many-queries block runs as
//
//   k_rowgemm<BWD_O>     dZ = dY.[Z>0] ; dO = dY + dZ Wo           (Wo^T image resident in LDS)
//   k_attn1_bwd          per head: P recomputed, dA, dS, dQp = dO + dS Kp ; dKp, dVp of the set
//                        (WAVE = HEAD: a wave owns the 32 features of one head of its 32-point
//                        tiles - no weight image at all, 9 KiB of LDS per wave)
//   k_rowgemm<BWD_Q>     dX (+)= dQp Wq                            (Wq^T image resident in LDS)
//   k_wgrad256           dW[256 x 256] = G^T A over the B*N rows, deterministic two-stage sum
//
// and the few-queries block (ISAB mab0 at dk = 256) in the reference's own formulation
// (modules.py:21: the N keys ARE projected) because the four 128 KiB operand images of the
// reassociated backward exceed the register file + LDS of a CU:
//
//   k_rowgemm<PROJ>      Kp / Vp = X Wk^T + bk  (bf16, [B*N, 256])
//   k_fq_attn_fwd        flash attention of the m shared queries of one head over the set's
//                        keys (wave = head, head dim 32): online softmax, O partials per range
//   k_fq_attn_bwd        dKp, dVp (bf16) and the set's dQp; both score orientations are
//                        recomputed on the MFMA (one extra 16x16x32 each) instead of transposed
//   k_rowgemm<BWD_Q> x2  dX (+)= dKp Wk + dVp Wv
//
// All activations cross these kernels in bf16 ([rows][256], row-major); accumulation, softmax
// statistics, biases and residuals are fp32.  Layout conventions: mfma_common.hpp.
#include "d256_bf16.hpp"
#include "slab_sum_body.hpp"

#include <math.h>

#include <mutex>

namespace pca {

namespace {

constexpr int TP = M1_TP;      // 128 points per 4-wave tile (as the forward: the ReLU mask index)

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// transposed fragment (k = the 32 points of a wave tile: slot (g, j) <-> point perm32(8g + j))
// from a small row-major bf16 image with `rb` bytes per row, for the 16 columns from col0
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

__device__ __forceinline__ bf16x8 cat8(bf16x4 lo, bf16x4 hi) {
  bf16x8 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) { r[e] = lo[e]; r[4 + e] = hi[e]; }
  return r;
}
__device__ __forceinline__ f32x4 tof(bf16x4 v) {
  return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
__device__ __forceinline__ bf16x4 zero4b() {
  return bf16x4{(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
}

// =====================================================================================
// k_rowgemm: Out^T[f][pt] = sum_k W[f][k] In^T[k][pt]  with the [D][D] bf16 weight image resident
// in LDS (K-permuted rows: the B operand comes from accumulator-layout registers, cat8) and one
// 32-point tile per wave.  Same unit / wave -> point mapping as k_mab1_fwd (the ReLU mask index).
//
// Global traffic is full 128-byte lines: a wave moves its [32 rows][256] bf16 tile in four
// 64-column chunks through a private 4 KiB LDS buffer - 16-byte coalesced loads / stores on the
// memory side (8 lanes = one 128-byte row segment), 8-byte accumulator-layout accesses on the LDS
// side.  (Accumulator-layout accesses straight to global memory touch 32 bytes of 16 different
// rows per instruction; with 8 waves x 16 KiB of rows in flight the 32 KiB L1 re-fetched every
// line up to four times: measured 209 -> see DESIGN.md.)
// =====================================================================================
enum { RG_PROJ = 0, RG_BWD_O = 1, RG_BWD_Q = 2, RG_FWD_O = 3 };

struct RowGemmArgs {
  const __bf16* In;       // [B*N][D]: PROJ X ; BWD_O dY ; BWD_Q dQp (or dKp / dVp) ; FWD_O O
  const __bf16* W;        // [D][D] image (see the launcher for which)
  const float* bias;      // PROJ, FWD_O
  const float* inv_scale; // F8: 1 / (per-tensor power-of-two scale of the fp8 weight image)
  const uint32_t* mask;   // BWD_O: ReLU mask bits of the forward
  uint32_t* mask_out;     // FWD_O (nullable)
  __bf16* Out;            // PROJ Y ; BWD_O dO ; BWD_Q dX ; FWD_O Y
  __bf16* Out2;           // BWD_O: dZ
  int B, N, tiles_per_set, accumulate;
};

// byte offset of 16-byte piece c16 (0..7) of row `row` in a [32][64] bf16 chunk (128-byte rows);
// the XOR spreads the 16 rows of an accumulator-layout access over all banks
__device__ __forceinline__ int cko(int row, int c16) {
  return row * 128 + ((c16 ^ ((row >> 1) & 7)) << 4);
}

// NB = 16-point blocks per wave: 2 (8 waves, 256 VGPRs) or 1 (16 waves of 128 VGPRs: four
// wavefronts per SIMD to hide the memory latency, twice the LDS weight reads per MFMA)
// F8 (PROJ, FWD_O): W is an fp8 e4m3 image of s * W ([D][D] bytes), the activations are converted
// to fp8 in registers; the accumulators are rescaled by 1 / s before the bias
template <int D, int MODE, int NB, bool F8 = false>
__global__ __launch_bounds__(1024 / NB, 4 / NB) void k_rowgemm(const RowGemmArgs a) {
  constexpr int NW = 16 / NB, NT = 64 * NW, SUBS = 2;
  constexpr int DT = D / 16, KS = D / 32, ROWB = F8 ? D : D * 2, NCH = D / 64;
  static_assert(!F8 || MODE == RG_PROJ || MODE == RG_FWD_O, "fp8 operands: forward projections");
  constexpr int CB = 16 * NB * 128;       // bytes of a wave's chunk buffer ([16 NB rows][64])
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sW = smem;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // wave wv of the workgroup = 16-point blocks [wv * NB, wv * NB + NB) of a 256-point unit; in the
  // forward's terms (mask index): tile `sub` of the unit, wave (0..3), block nb0 + nb of that wave
  const int blk0 = wv * NB, sub = blk0 >> 3, wave = (blk0 & 7) >> 1, nb0 = blk0 & 1;
  const int r = lane & 15, g = lane >> 4;
  char* myC = smem + D * ROWB + wv * CB;          // (behind the weight image)
  {
    constexpr int CPR = ROWB / 16;                 // 16-byte chunks per image row
    constexpr int NC = D * CPR / NT;
    uint4 wv[NC];
    const char* gW = reinterpret_cast<const char*>(a.W);
#pragma unroll
    for (int e = 0; e < NC; ++e) {
      const int c = tid + NT * e, row = c / CPR, c16 = c % CPR;
      wv[e] = *reinterpret_cast<const uint4*>(gW + (int64_t)row * ROWB + c16 * 16);
    }
#pragma unroll
    for (int e = 0; e < NC; ++e) {
      const int c = tid + NT * e, row = c / CPR, c16 = c % CPR;
      if (F8) {
        *reinterpret_cast<uint2*>(sW + f8off<D>(row, 2 * c16)) = uint2{wv[e].x, wv[e].y};
        *reinterpret_cast<uint2*>(sW + f8off<D>(row, 2 * c16 + 1)) = uint2{wv[e].z, wv[e].w};
      } else {
        *reinterpret_cast<uint4*>(sW + swz(row, c16, ROWB)) = wv[e];
      }
    }
  }
  const float inv_s = F8 ? a.inv_scale[0] : 1.f;
  (void)inv_s;
  __syncthreads();
  const int units_per_set = (a.tiles_per_set + SUBS - 1) / SUBS;
  const int total_units = a.B * units_per_set;
  // coalesced side of the chunk moves: piece i of a lane = row (lane + 64 i) / 8, 16-byte column
  // (lane + 64 i) % 8; accumulator side: row 16 nb + r, 8 bytes at columns 16 t' + 4 g
  const int crow0 = lane >> 3, cc16 = lane & 7;
  for (int unit = blockIdx.x; unit < total_units; unit += gridDim.x) {
    const int b = unit / units_per_set, tile = (unit - b * units_per_set) * SUBS + sub;
    if (tile >= a.tiles_per_set) continue;                 // (no barrier inside the loop)
    const int n_base = tile * TP + wave * 32 + nb0 * 16;
    const int64_t rowbase = (int64_t)b * a.N + n_base;
    const int nlive = a.N - n_base;                        // rows of this wave tile that exist
    // FWD_O keeps O (the bf16 B operand) alive for the residual: its output features are
    // computed in two halves of 128 so that the accumulators need 64 registers, not 128
    constexpr int HT = 1, DTH = DT / HT;
    f32x4 acc[DTH][NB];
    bf16x8 bop[KS][NB];
    uint32_t bits[NB][D / 128];
    if (MODE == RG_BWD_O) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int w = 0; w < D / 128; ++w)
          bits[nb][w] = a.mask[mab1_mask_index<D>(b, a.tiles_per_set, tile, wave, nb0 + nb, w, lane)];
    }
    constexpr int NI = 2 * NB;             // 16-byte pieces per lane and chunk
    uint4 st[2][NI];
    auto fetch = [&](int c, uint4 (&dst)[NI]) {
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        // rows past the end of the set read its last row instead (never stored: a point only
        // feeds its own output column) - a load under a divergent condition gets its own basic
        // block and s_waitcnt vmcnt(0), which serialised the NI loads of a chunk
        // (BWD_O keeps the guarded form: with the loads free to move, the scheduler hoists them
        //  over the 128 accumulators it initialises from dY and spills 144 bytes per lane)
        const int n = n_base + crow0 + 8 * i;
        if (MODE == RG_BWD_O)
          dst[i] = n < a.N ? *reinterpret_cast<const uint4*>(a.In + ((int64_t)b * a.N + n) * D +
                                                             64 * c + 8 * cc16)
                           : uint4{0u, 0u, 0u, 0u};
        else
          dst[i] = *reinterpret_cast<const uint4*>(
              a.In + ((int64_t)b * a.N + (n < a.N ? n : a.N - 1)) * D + 64 * c + 8 * cc16);
      }
    };
    fetch(0, st[0]);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      if (c + 1 < NCH) fetch(c + 1, st[(c + 1) & 1]);
#pragma unroll
      for (int i = 0; i < NI; ++i)
        *reinterpret_cast<uint4*>(myC + cko(crow0 + 8 * i, cc16)) = st[c & 1][i];
      bf16x4 in4[4][NB];
#pragma unroll
      for (int tq = 0; tq < 4; ++tq)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          in4[tq][nb] = *reinterpret_cast<const bf16x4*>(myC + cko(16 * nb + r, 2 * tq + (g >> 1)) +
                                                        8 * (g & 1));
#pragma unroll
      for (int tq = 0; tq < 4; ++tq) {
        const int t = 4 * c + tq;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          if constexpr (MODE == RG_BWD_O) {
            acc[t][nb] = tof(in4[tq][nb]);                  // dO starts as dY (residual path)
            bf16x4 z4;
#pragma unroll
            for (int e = 0; e < 4; ++e)
              z4[e] = ((bits[nb][t / 8] >> ((t & 7) * 4 + e)) & 1u) ? in4[tq][nb][e] : (__bf16)0.f;
            in4[tq][nb] = z4;
            // dZ goes back through the chunk buffer to leave in full lines
            *reinterpret_cast<bf16x4*>(myC + cko(16 * nb + r, 2 * tq + (g >> 1)) + 8 * (g & 1)) = z4;
          }
        }
      }
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        bop[2 * c][nb] = cat8(in4[0][nb], in4[1][nb]);
        bop[2 * c + 1][nb] = cat8(in4[2][nb], in4[3][nb]);
      }
      if (MODE == RG_BWD_O) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const int row = crow0 + 8 * i;
          const uint4 v = *reinterpret_cast<const uint4*>(myC + cko(row, cc16));
          if (row < nlive)
            *reinterpret_cast<uint4*>(a.Out2 + (rowbase + row) * D + 64 * c + 8 * cc16) = v;
        }
      }
    }
    // FWD_O: Y = O + relu(Z) - O is the bf16 B operand itself (feature 16 t + 4 g + e = element
    // 4 (t & 1) + e of k-block t / 2); ReLU mask bits in the layout of k_mab1_fwd
    uint32_t mb[NB][D / 128];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int w = 0; w < D / 128; ++w) mb[nb][w] = 0u;
#pragma unroll
    for (int hf = 0; hf < HT; ++hf) {
      if (MODE != RG_BWD_O) {
#pragma unroll
        for (int tt = 0; tt < DTH; ++tt) {
          float4 b4 = float4{0.f, 0.f, 0.f, 0.f};
          if (!F8 && (MODE == RG_PROJ || MODE == RG_FWD_O))
            b4 = *reinterpret_cast<const float4*>(a.bias + 16 * (hf * DTH + tt) + 4 * g);
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) acc[tt][nb] = f32x4{b4.x, b4.y, b4.z, b4.w};
        }
      }
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        if (F8) {
          f8x8 b8[NB];
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) b8[nb] = bf_to_f8(bop[s][nb]);
#pragma unroll
          for (int tt = 0; tt < DTH; ++tt) {
            const f8x8 wa8 = *reinterpret_cast<const f8x8*>(
                sW + f8off<D>(16 * (hf * DTH + tt) + r, 4 * s + g));
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[tt][nb] = mfma32_f8(wa8, b8[nb], acc[tt][nb]);
          }
        } else {
#pragma unroll
          for (int tt = 0; tt < DTH; ++tt) {
            const bf16x8 wa = *reinterpret_cast<const bf16x8*>(
                sW + swz(16 * (hf * DTH + tt) + r, 4 * s + g, ROWB));
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[tt][nb] = mfma32(wa, bop[s][nb], acc[tt][nb]);
          }
        }
      }
      if (F8) {
#pragma unroll
        for (int tt = 0; tt < DTH; ++tt) {
          const float4 b4 = *reinterpret_cast<const float4*>(a.bias + 16 * (hf * DTH + tt) + 4 * g);
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            acc[tt][nb][0] = acc[tt][nb][0] * inv_s + b4.x;
            acc[tt][nb][1] = acc[tt][nb][1] * inv_s + b4.y;
            acc[tt][nb][2] = acc[tt][nb][2] * inv_s + b4.z;
            acc[tt][nb][3] = acc[tt][nb][3] * inv_s + b4.w;
          }
        }
      }
#pragma unroll
      for (int cc = 0; cc < NCH / HT; ++cc) {
        const int c = hf * (NCH / HT) + cc;
#pragma unroll
        for (int tq = 0; tq < 4; ++tq) {
          const int t = 4 * c + tq, tt = t - hf * DTH;
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            f32x4 v = acc[tt][nb];
            if (MODE == RG_FWD_O) {
              const bf16x8 ob = bop[t / 2][nb];
              const f32x4 of = tof((t & 1) ? __builtin_shufflevector(ob, ob, 4, 5, 6, 7)
                                           : __builtin_shufflevector(ob, ob, 0, 1, 2, 3));
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float zz = v[e];
                if (zz > 0.f) mb[nb][t / 8] |= 1u << ((t & 7) * 4 + e);
                v[e] = of[e] + fmaxf(zz, 0.f);
              }
            }
            if (MODE == RG_BWD_Q && a.accumulate) {
              const int n = 16 * nb + r;
              if (n < nlive) {
                const f32x4 o = tof(*reinterpret_cast<const bf16x4*>(a.Out + (rowbase + n) * D +
                                                                     16 * t + 4 * g));
                v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3];
              }
            }
            *reinterpret_cast<bf16x4*>(myC + cko(16 * nb + r, 2 * tq + (g >> 1)) + 8 * (g & 1)) =
                pack4(v);
          }
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const int row = crow0 + 8 * i;
          const uint4 v = *reinterpret_cast<const uint4*>(myC + cko(row, cc16));
          if (row < nlive)
            *reinterpret_cast<uint4*>(a.Out + (rowbase + row) * D + 64 * c + 8 * cc16) = v;
        }
      }
    }
    if (MODE == RG_FWD_O && a.mask_out != nullptr) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int w = 0; w < D / 128; ++w)
          a.mask_out[mab1_mask_index<D>(b, a.tiles_per_set, tile, wave, nb0 + nb, w, lane)] =
              mb[nb][w];
    }
  }
}

template <int MODE, int NBW, bool F8 = false>
int launch_rowgemm_nb(const RowGemmArgs& a, hipStream_t st) {
  constexpr int D = 256;
  static std::once_flag once;
  std::call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_rowgemm<D, MODE, NBW, F8>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  const int total = a.B * ((a.tiles_per_set + 1) / 2);
  const int grid = total < 256 ? total : 256;
  hipLaunchKernelGGL((k_rowgemm<D, MODE, NBW, F8>), dim3(grid), dim3(1024 / NBW),
                     (size_t)D * D * (F8 ? 1 : 2) + 8 * 4096, st, a);
  return check_launch("k_rowgemm");
}
template <int MODE>
int launch_rowgemm(const RowGemmArgs& a, hipStream_t st) {
  // measured at configs[3] (B = 128, N = 4096): 8 waves x 2 blocks is the faster shape for PROJ /
  // BWD_O / BWD_Q (126 / 147 / 172 us against 133 / 173 / 176), 16 waves x 1 block for FWD_O,
  // whose 8-wave build spills (181 against 195 us)
  if (MODE == RG_FWD_O) return launch_rowgemm_nb<MODE, 1>(a, st);
  return launch_rowgemm_nb<MODE, 2>(a, st);
}

// =====================================================================================
// k_attn1_bwd: attention adjoint of the many-queries block, WAVE = HEAD
// =====================================================================================
struct Attn1BwdArgs {
  const __bf16* dO;         // [B*N][D]
  const __bf16* QpS;        // [B*N][D] projected queries saved by the forward
  const __bf16 *KpP, *VpP;  // [B][MI][D] (K-permuted features inside each head)
  const __bf16* Kt;         // [B][D][MI] (keys in perm32 order)
  __bf16* dQp;              // [B*N][D]
  float *dKpPart, *dVpPart; // [B][nparts][MI][D]
  int B, N, nparts, pts_per_part;
  float scale, scale_log2e;
};

// (k_attn1_bwd, the per-wave global-traffic form of round 1, was removed in round 4: k_attn1_bwd2 below has
//  been the measured winner since round 2 - DESIGN.md 4.5.)

// k_attn1_bwd2: the attention adjoint of the many-queries block, WAVE = HEAD, with full-line global
// traffic.  (With every wave fetching its head's 64-byte slice of each row straight from memory - 16 rows
// x 32 bytes per load instruction, 8-byte stores - it took 243 us at configs[3] for 0.8 GB, 3.3 TB/s.)
// Here the workgroup moves whole [32 points][256] tiles: Qp and dO arrive by LDS-DMA (1 KiB per wave
// instruction, double buffered, swizzled like the single-launch forward's tiles), each wave reads /
// writes its head's slice of the tiles in LDS, and the dQp tile leaves in 16-byte pieces of full rows.
typedef __attribute__((address_space(3))) void lds_void_t;
template <int D>
__global__ __launch_bounds__(64 * (D / 32), 2) void k_attn1_bwd2(const Attn1BwdArgs a) {
  constexpr int MI = 32, ROWB = D * 2, TILEB = 32 * ROWB;
  constexpr int PQ = 72, IMG = 32 * PQ;
  static_assert(D == 256, "8 waves, 2 DMA pieces per wave and tensor");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sQb = smem;                       // [2][TILEB]
  char* sOb = smem + 2 * TILEB;           // [2][TILEB]
  char* sOut = smem + 4 * TILEB;          // dQp tile
  const int tid = threadIdx.x, lane = tid & 63;
  const int j = __builtin_amdgcn_readfirstlane(tid >> 6);      // head of this wave
  const int r = lane & 15, g = lane >> 4;
  const int b = blockIdx.x / a.nparts, part = blockIdx.x - b * a.nparts;
  char* myDS = smem + 5 * TILEB + j * 4 * IMG;
  char* myP = myDS + IMG;
  char* myQ = myP + IMG;
  char* myO = myQ + IMG;

  bf16x8 kpa[2], vpa[2], kta[2];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt) {
    const int64_t o = ((int64_t)b * MI + 16 * kt + r) * D + 32 * j + 8 * g;
    kpa[kt] = *reinterpret_cast<const bf16x8*>(a.KpP + o);
    vpa[kt] = *reinterpret_cast<const bf16x8*>(a.VpP + o);
    kta[kt] = *reinterpret_cast<const bf16x8*>(a.Kt + ((int64_t)b * D + 32 * j + 16 * kt + r) * MI +
                                               8 * g);
  }
  f32x4 dkp[2][2], dvp[2][2];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      dkp[kt][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
      dvp[kt][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  const int n_lo = part * a.pts_per_part;
  const int n_hi = n_lo + a.pts_per_part < a.N ? n_lo + a.pts_per_part : a.N;
  const int T = n_lo < n_hi ? (n_hi - n_lo + 31) / 32 : 0;
  // accumulator-layout 8 bytes of this lane in a tile (row = point r, features 32 j + 16 t + 4 g)
  // and its coalesced 16-byte piece (row tid / 32, chunk tid % 32), as in k_isab1_fwd256
  int oD[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) oD[t] = swz(r, 4 * j + 2 * t + (g >> 1), ROWB) + 8 * (g & 1);
  const int oC = swz(tid >> 5, tid & 31, ROWB);
  auto lds_barrier = [] {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  // LDS-DMA of tile k of both tensors (issued from inline asm: see k_isab1_fwd256)
  auto dma = [&](int k) {
    const int n0 = n_lo + 32 * k, par = k & 1;
#pragma unroll
    for (int w = 0; w < 2; ++w) {
      const __bf16* base = w == 0 ? a.QpS : a.dO;
      char* dst = (w == 0 ? sQb : sOb) + par * TILEB;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int p = (2 * j + i) * 64 + lane;
        const int row = p >> 5, slot = p & 31;
        const int ch = (slot & ~15) | ((slot ^ row) & 15);
        const int n = n0 + row < a.N ? n0 + row : a.N - 1;
        const __bf16* src = base + ((int64_t)b * a.N + n) * D + ch * 8;
        const unsigned ldst = __builtin_amdgcn_readfirstlane(
            (unsigned)(uintptr_t)(lds_void_t*)(dst + (2 * j + i) * 1024));
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(ldst) : "memory");
      }
    }
  };
  if (T > 0) dma(0);
  for (int k = 0; k < T; ++k) {
    const int par = k & 1, n0 = n_lo + 32 * k, nlive = n_hi - n0;
    const char* sQ = sQb + par * TILEB;
    const char* sO = sOb + par * TILEB;
    // tile k + 1 starts to stream in; everything older than it and the (two, for a full tile)
    // dQp stores of tile k - 1 - i.e. this tile's DMA - must have landed
    if (k + 1 < T) {
      dma(k + 1);
      if (k == 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // (tile k - 1 was full: not the last)
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    lds_barrier();                       // B0: both tiles complete; the previous dQp tile is stored
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const bool live = 16 * nb + r < nlive;
      const bf16x4 qlo = *reinterpret_cast<const bf16x4*>(sQ + oD[0] + 8192 * nb);
      const bf16x4 qhi = *reinterpret_cast<const bf16x4*>(sQ + oD[1] + 8192 * nb);
      const bf16x4 o0 = *reinterpret_cast<const bf16x4*>(sO + oD[0] + 8192 * nb);
      const bf16x4 o1 = *reinterpret_cast<const bf16x4*>(sO + oD[1] + 8192 * nb);
      const bf16x8 qb = cat8(qlo, qhi), dob = cat8(o0, o1);
      f32x4 dq0 = tof(o0), dq1 = tof(o1);                 // dQp starts as dO (residual Q_)
      f32x4 p0 = {0.f, 0.f, 0.f, 0.f}, p1 = p0, da0 = p0, da1 = p0;
      p0 = mfma32(kpa[0], qb, p0);
      p1 = mfma32(kpa[1], qb, p1);
      da0 = mfma32(vpa[0], dob, da0);
      da1 = mfma32(vpa[1], dob, da1);
      float mx = fmaxf(fmaxf(fmaxf(p0[0], p0[1]), fmaxf(p0[2], p0[3])),
                       fmaxf(fmaxf(p1[0], p1[1]), fmaxf(p1[2], p1[3])));
      mx = wave16_max(mx);
      float sum = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        p0[e] = __builtin_amdgcn_exp2f((p0[e] - mx) * a.scale_log2e);
        p1[e] = __builtin_amdgcn_exp2f((p1[e] - mx) * a.scale_log2e);
        sum += p0[e] + p1[e];
      }
      sum = wave16_sum(sum);
      const float inv = __builtin_amdgcn_rcpf(sum);
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
      // wave-private [point][.] images for the sums over points (padding points: P = dS = 0, so
      // the duplicated rows the DMA fetched for them do not count)
      const int pt = 16 * nb + r;
      const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<bf16x4*>(myDS + pt * PQ + 8 * g) = pack4(live ? ds0 : zero4);
      *reinterpret_cast<bf16x4*>(myDS + pt * PQ + 32 + 8 * g) = pack4(live ? ds1 : zero4);
      *reinterpret_cast<bf16x4*>(myP + pt * PQ + 8 * g) = pack4(live ? p0 : zero4);
      *reinterpret_cast<bf16x4*>(myP + pt * PQ + 32 + 8 * g) = pack4(live ? p1 : zero4);
      *reinterpret_cast<bf16x4*>(myQ + pt * PQ + 8 * g) = qlo;
      *reinterpret_cast<bf16x4*>(myQ + pt * PQ + 32 + 8 * g) = qhi;
      *reinterpret_cast<bf16x4*>(myO + pt * PQ + 8 * g) = o0;
      *reinterpret_cast<bf16x4*>(myO + pt * PQ + 32 + 8 * g) = o1;
      const bf16x8 dsb = pack8(ds0, ds1);
      dq0 = mfma32(kta[0], dsb, dq0);
      dq1 = mfma32(kta[1], dsb, dq1);
      *reinterpret_cast<bf16x4*>(sOut + oD[0] + 8192 * nb) = pack4(dq0);
      *reinterpret_cast<bf16x4*>(sOut + oD[1] + 8192 * nb) = pack4(dq1);
    }
    bf16x8 qf[2], of[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      qf[tt] = tr_frag_small(myQ, PQ, 16 * tt, lane);
      of[tt] = tr_frag_small(myO, PQ, 16 * tt, lane);
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      const bf16x8 ads = tr_frag_small(myDS, PQ, 16 * kt, lane);
      const bf16x8 ap = tr_frag_small(myP, PQ, 16 * kt, lane);
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        dkp[kt][tt] = mfma32(ads, qf[tt], dkp[kt][tt]);
        dvp[kt][tt] = mfma32(ap, of[tt], dvp[kt][tt]);
      }
    }
    lds_barrier();                       // B1: the dQp tile is complete
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = tid + 512 * i, row = c >> 5, ch = c & 31;
      if (row < nlive)
        *reinterpret_cast<uint4*>(a.dQp + ((int64_t)b * a.N + n0 + row) * D + ch * 8) =
            *reinterpret_cast<const uint4*>(sOut + oC + 8192 * i);
    }
  }
  const int64_t pbase = ((int64_t)b * a.nparts + part) * MI * D;
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int64_t o = pbase + (int64_t)(16 * kt + 4 * g + e) * D + 32 * j + 16 * tt + r;
        a.dKpPart[o] = dkp[kt][tt][e];
        a.dVpPart[o] = dvp[kt][tt][e];
      }
}

// k_attn1_bwd3: k_attn1_bwd2 with the fc_o adjoint in front of it (what k_rowgemm<BWD_O> did in
// a launch of its own): dZ = dY . [Z > 0] ; dO = dY + dZ Wo, the head's 32 columns of dO computed
// by the wave that consumes them - dO never goes to memory (one [B*N, 256] tensor less written
// and read per block).  The wave keeps its [32 x 256] slice of Wo^T as MFMA A operands (64
// registers); the dZ tile is assembled in LDS from the waves' own slices (mask bytes in the
// forward's layout) and leaves for the weight-gradient pass in full rows.
struct Attn1Bwd3Args {
  Attn1BwdArgs base;
  const __bf16* dY;         // [B*N][D]
  const uint32_t* mask;     // ReLU mask words (mab1_mask_index<256>)
  const __bf16* WoT;        // [256][256] bf16: row = column c of Wo, col = feature f (Wo[f][c])
  __bf16* dZ;               // [B*N][D]
  int tiles128;             // 128-point tiles per set (mask pitch)
  // SMALLQ (layer 1, dq <= 4): Qp is recomputed from the points - nothing was saved
  const float* Xs;          // [B*N][dq] fp32
  const float* WqF;         // [256][dq] fp32
  const float* bq;
  int dq;
};
// SMALLQ: layer 1 (two or three input columns).  The projected queries are not read back (the
// forward does not save them: one [B*N, 256] tensor less written and one less read) but recomputed
// from the tile's points with the forward's own expression, so the bf16 values are the same.
// STORE_DZ = false: dZ does not leave the kernel - the fc_o weight-gradient job then reads dY and the
// mask itself (Wgrad256Job::mask): one [B*N, 256] tensor less written per block.
template <int D, bool SMALLQ, bool STORE_DZ = true>
__global__ __launch_bounds__(64 * (D / 32), 2) void k_attn1_bwd3(const Attn1Bwd3Args aa) {
  const Attn1BwdArgs& a = aa.base;
  constexpr int MI = 32, ROWB = D * 2, TILEB = 32 * ROWB, KS = D / 32;
  constexpr int PQ = 72, IMG = 32 * PQ;
  static_assert(D == 256, "8 waves, 2 DMA pieces per wave and tensor");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sY = smem;                        // dY tile (single: refilled once phase A has read it)
  char* sQb = smem + TILEB;               // [2][TILEB] Qp tiles
  char* sZ = smem + 3 * TILEB;            // dZ tile
  char* sOut = smem + 4 * TILEB;          // dQp tile
  uint32_t* sMaskb = reinterpret_cast<uint32_t*>(smem + 5 * TILEB);      // [2][256 words]
  // SMALLQ: the tile's points [32][dq] fp32 arrive by LDS-DMA too (in the unused Qp tile buffers):
  // ordinary global loads inside the loop make hipcc drain the DMA queue at their first use
  float* sPts = reinterpret_cast<float*>(sQb);                           // [2][128 floats]
  const int tid = threadIdx.x, lane = tid & 63;
  const int j = __builtin_amdgcn_readfirstlane(tid >> 6);      // head of this wave
  const int r = lane & 15, g = lane >> 4;
  const int b = blockIdx.x / a.nparts, part = blockIdx.x - b * a.nparts;
  char* myDS = smem + 5 * TILEB + 2048 + j * 4 * IMG;
  char* myP = myDS + IMG;
  char* myQ = myP + IMG;
  char* myO = myQ + IMG;

  bf16x8 woa[KS][2];
#pragma unroll
  for (int s = 0; s < KS; ++s)
#pragma unroll
    for (int t = 0; t < 2; ++t)
      woa[s][t] = *reinterpret_cast<const bf16x8*>(aa.WoT + (int64_t)(32 * j + 16 * t + r) * D +
                                                   32 * s + 8 * g);
  bf16x8 kpa[2], vpa[2], kta[2];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt) {
    const int64_t o = ((int64_t)b * MI + 16 * kt + r) * D + 32 * j + 8 * g;
    kpa[kt] = *reinterpret_cast<const bf16x8*>(a.KpP + o);
    vpa[kt] = *reinterpret_cast<const bf16x8*>(a.VpP + o);
    kta[kt] = *reinterpret_cast<const bf16x8*>(a.Kt + ((int64_t)b * D + 32 * j + 16 * kt + r) * MI +
                                               8 * g);
  }
  float wqs[SMALLQ ? 2 : 1][4][4];
  f32x4 bqv[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  if (SMALLQ) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const float4 b4 = *reinterpret_cast<const float4*>(aa.bq + 32 * j + 16 * t + 4 * g);
      bqv[t] = f32x4{b4.x, b4.y, b4.z, b4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int c = 0; c < 4; ++c)
          wqs[SMALLQ ? t : 0][e][c] =
              c < aa.dq ? aa.WqF[(32 * j + 16 * t + 4 * g + e) * aa.dq + c] : 0.f;
    }
  }
  f32x4 dkp[2][2], dvp[2][2];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      dkp[kt][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
      dvp[kt][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  const int n_lo = part * a.pts_per_part;
  const int n_hi = n_lo + a.pts_per_part < a.N ? n_lo + a.pts_per_part : a.N;
  const int T = n_lo < n_hi ? (n_hi - n_lo + 31) / 32 : 0;
  int oB[4], oD[2];
#pragma unroll
  for (int k = 0; k < 4; ++k) oB[k] = swz(r, 4 * k + g, ROWB);
#pragma unroll
  for (int t = 0; t < 2; ++t) oD[t] = swz(r, 4 * j + 2 * t + (g >> 1), ROWB) + 8 * (g & 1);
  const int oC = swz(tid >> 5, tid & 31, ROWB);
  auto lds_barrier = [] {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  auto dma_piece = [&](const void* src, char* dst) {
    const unsigned ldst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_void_t*)dst);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(ldst) : "memory");
  };
  // tile k of a [B*N][256] tensor into `dst` (2 pieces of 1 KiB per wave, swizzled at the source)
  auto dma_tile = [&](const __bf16* base, int k, char* dst) {
    const int n0 = n_lo + 32 * k;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int p = (2 * j + i) * 64 + lane;
      const int row = p >> 5, slot = p & 31;
      const int ch = (slot & ~15) | ((slot ^ row) & 15);
      const int n = n0 + row < a.N ? n0 + row : a.N - 1;
      dma_piece(base + ((int64_t)b * a.N + n) * D + ch * 8, dst + (2 * j + i) * 1024);
    }
  };
  // Qp tile k and (wave 0) the tile's 256 mask words: 1 KiB contiguous in the forward's layout
  auto dma_q = [&](int k) {
    if (!SMALLQ) dma_tile(a.QpS, k, sQb + (k & 1) * TILEB);
    if (SMALLQ && (j == 1 || j == 2)) {
      // floats [64 (j - 1), 64 j) of the tile's 32 * dq (<= 128) values; past the end of the tensor:
      // its last element (such points are masked out below)
      const int64_t first = ((int64_t)b * a.N + n_lo + 32 * k) * aa.dq;
      const int64_t last = (int64_t)a.B * a.N * aa.dq - 1;
      int64_t i = first + 64 * (j - 1) + lane;
      i = i < last ? i : last;
      const unsigned ldst = __builtin_amdgcn_readfirstlane(
          (unsigned)(uintptr_t)(lds_void_t*)(sPts + (k & 1) * 128 + 64 * (j - 1)));
      unsigned keep;
      const float* src = aa.Xs + i;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                   "global_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(src), "s"(ldst) : "memory");
    }
    if (j == 0) {
      const int tile32 = (n_lo >> 5) + k;
      const uint32_t* m = aa.mask + (((int64_t)b * aa.tiles128 * 8 + 2 * tile32) * 2) * 64;
      dma_piece(m + 4 * lane, reinterpret_cast<char*>(sMaskb + (k & 1) * 256));
    }
  };
  if (T > 0) {
    dma_q(0);
    dma_tile(aa.dY, 0, sY);
  }
  // DMA instructions of dma_q in this wave
  const int nq = (SMALLQ ? ((j == 1 || j == 2) ? 1 : 0) : 2) + (j == 0 ? 1 : 0);
  for (int k = 0; k < T; ++k) {
    const int par = k & 1, n0 = n_lo + 32 * k, nlive = n_hi - n0;
    const char* sQ = sQb + par * TILEB;
    const uint32_t* sMask = sMaskb + par * 256;
    float xv[2][4];
    // Qp / mask of tile k + 1 start now (their buffers were last read in iteration k - 1); what must
    // have landed is this tile's dY (issued after barrier B1 of iteration k - 1) and everything older:
    // younger are only the 4 stores of tile k - 1 (always a full tile) and the DMA just issued
    {
      int younger = k > 0 ? (STORE_DZ ? 4 : 2) : 0;
      if (k + 1 < T) {
        dma_q(k + 1);
        younger += nq;
      }
      switch (younger) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
      }
    }
    lds_barrier();                       // B0: dY, Qp, mask of tile k; tile k - 1 fully stored from LDS
    if (SMALLQ) {                        // this lane's points (row r of block nb)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int c = 0; c < 4; ++c)
          xv[nb][c] = c < aa.dq ? sPts[par * 128 + (16 * nb + r) * aa.dq + (c < aa.dq ? c : 0)] : 0.f;
    }
    // ---- phase A: own slice of dZ = dY . [Z > 0]; the dY slice stays as the residual ----
    bf16x4 res[2][2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const uint32_t bits = (sMask[(nb * 2 + (j >> 2)) * 64 + lane] >> (8 * (j & 3))) & 0xffu;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const bf16x4 y4 = *reinterpret_cast<const bf16x4*>(sY + oD[t] + 8192 * nb);
        res[t][nb] = y4;
        bf16x4 z4;
#pragma unroll
        for (int e = 0; e < 4; ++e) z4[e] = ((bits >> (4 * t + e)) & 1u) ? y4[e] : (__bf16)0.f;
        *reinterpret_cast<bf16x4*>(sZ + oD[t] + 8192 * nb) = z4;
      }
    }
    lds_barrier();                       // B1: dZ tile complete; dY tile consumed
    if (k + 1 < T) dma_tile(aa.dY, k + 1, sY);
    if (STORE_DZ) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int c = tid + 512 * i, row = c >> 5, ch = c & 31;
        if (row < nlive)
          *reinterpret_cast<uint4*>(aa.dZ + ((int64_t)b * a.N + n0 + row) * D + ch * 8) =
              *reinterpret_cast<const uint4*>(sZ + oC + 8192 * i);
      }
    }
    // ---- phase B: dO_h = dY_h + (dZ Wo)_h, then the attention adjoint of this head ----
    f32x4 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) acc[t][nb] = tof(res[t][nb]);
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const bf16x8 zb =
            *reinterpret_cast<const bf16x8*>(sZ + oB[s & 3] + 256 * (s >> 2) + 8192 * nb);
        acc[0][nb] = mfma32(woa[s][0], zb, acc[0][nb]);
        acc[1][nb] = mfma32(woa[s][1], zb, acc[1][nb]);
      }
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const bool live = 16 * nb + r < nlive;
      bf16x4 qlo, qhi;
      if (SMALLQ) {                        // the forward's expression (k_isab1_fwd256<true>)
        f32x4 q[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            q[t][e] = bqv[t][e] + wqs[SMALLQ ? t : 0][e][0] * xv[nb][0] +
                      wqs[SMALLQ ? t : 0][e][1] * xv[nb][1] + wqs[SMALLQ ? t : 0][e][2] * xv[nb][2] +
                      wqs[SMALLQ ? t : 0][e][3] * xv[nb][3];
        qlo = pack4(q[0]);
        qhi = pack4(q[1]);
      } else {
        qlo = *reinterpret_cast<const bf16x4*>(sQ + oD[0] + 8192 * nb);
        qhi = *reinterpret_cast<const bf16x4*>(sQ + oD[1] + 8192 * nb);
      }
      const bf16x4 o0 = pack4(acc[0][nb]), o1 = pack4(acc[1][nb]);     // dO, bf16 as before
      const bf16x8 qb = cat8(qlo, qhi), dob = cat8(o0, o1);
      f32x4 dq0 = tof(o0), dq1 = tof(o1);                 // dQp starts as dO (residual Q_)
      f32x4 p0 = {0.f, 0.f, 0.f, 0.f}, p1 = p0, da0 = p0, da1 = p0;
      p0 = mfma32(kpa[0], qb, p0);
      p1 = mfma32(kpa[1], qb, p1);
      da0 = mfma32(vpa[0], dob, da0);
      da1 = mfma32(vpa[1], dob, da1);
      float mx = fmaxf(fmaxf(fmaxf(p0[0], p0[1]), fmaxf(p0[2], p0[3])),
                       fmaxf(fmaxf(p1[0], p1[1]), fmaxf(p1[2], p1[3])));
      mx = wave16_max(mx);
      float sum = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        p0[e] = __builtin_amdgcn_exp2f((p0[e] - mx) * a.scale_log2e);
        p1[e] = __builtin_amdgcn_exp2f((p1[e] - mx) * a.scale_log2e);
        sum += p0[e] + p1[e];
      }
      sum = wave16_sum(sum);
      const float inv = __builtin_amdgcn_rcpf(sum);
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
      const int pt = 16 * nb + r;
      const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<bf16x4*>(myDS + pt * PQ + 8 * g) = pack4(live ? ds0 : zero4);
      *reinterpret_cast<bf16x4*>(myDS + pt * PQ + 32 + 8 * g) = pack4(live ? ds1 : zero4);
      *reinterpret_cast<bf16x4*>(myP + pt * PQ + 8 * g) = pack4(live ? p0 : zero4);
      *reinterpret_cast<bf16x4*>(myP + pt * PQ + 32 + 8 * g) = pack4(live ? p1 : zero4);
      *reinterpret_cast<bf16x4*>(myQ + pt * PQ + 8 * g) = qlo;
      *reinterpret_cast<bf16x4*>(myQ + pt * PQ + 32 + 8 * g) = qhi;
      *reinterpret_cast<bf16x4*>(myO + pt * PQ + 8 * g) = o0;
      *reinterpret_cast<bf16x4*>(myO + pt * PQ + 32 + 8 * g) = o1;
      const bf16x8 dsb = pack8(ds0, ds1);
      dq0 = mfma32(kta[0], dsb, dq0);
      dq1 = mfma32(kta[1], dsb, dq1);
      *reinterpret_cast<bf16x4*>(sOut + oD[0] + 8192 * nb) = pack4(dq0);
      *reinterpret_cast<bf16x4*>(sOut + oD[1] + 8192 * nb) = pack4(dq1);
    }
    bf16x8 qf[2], of[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      qf[tt] = tr_frag_small(myQ, PQ, 16 * tt, lane);
      of[tt] = tr_frag_small(myO, PQ, 16 * tt, lane);
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      const bf16x8 ads = tr_frag_small(myDS, PQ, 16 * kt, lane);
      const bf16x8 ap = tr_frag_small(myP, PQ, 16 * kt, lane);
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        dkp[kt][tt] = mfma32(ads, qf[tt], dkp[kt][tt]);
        dvp[kt][tt] = mfma32(ap, of[tt], dvp[kt][tt]);
      }
    }
    lds_barrier();                       // B2: the dQp tile is complete
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = tid + 512 * i, row = c >> 5, ch = c & 31;
      if (row < nlive)
        *reinterpret_cast<uint4*>(a.dQp + ((int64_t)b * a.N + n0 + row) * D + ch * 8) =
            *reinterpret_cast<const uint4*>(sOut + oC + 8192 * i);
    }
  }
  const int64_t pbase = ((int64_t)b * a.nparts + part) * MI * D;
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int64_t o = pbase + (int64_t)(16 * kt + 4 * g + e) * D + 32 * j + 16 * tt + r;
        a.dKpPart[o] = dkp[kt][tt][e];
        a.dVpPart[o] = dvp[kt][tt][e];
      }
}

// dk[b][i] = sum_p kp[b][p][i] (same for v): the per-range partials of k_attn1_bwd
__global__ void k_sum_parts256(const float* __restrict__ kp, const float* __restrict__ vp,
                               float* __restrict__ dk, float* __restrict__ dv, int B, int nparts,
                               int n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)B * n) return;
  const int64_t b = i / n, o = i - b * n;
  float x = 0.f, y = 0.f;
  for (int p = 0; p < nparts; ++p) {
    x += kp[(b * nparts + p) * n + o];
    y += vp[(b * nparts + p) * n + o];
  }
  dk[i] = x;
  dv[i] = y;
}

// =====================================================================================
// k_wgrad256: dW[D x D] = G[M x D]^T A[M x D] (+ db = column sums of G), bf16 operands
// =====================================================================================
// [32 rows][256] bf16 tile as two [32][128] halves, each in the transposed-read layout (b) of
// cdna_hip_programming.md T10 (conflict-free ds_read_tr16_b64)
__device__ __forceinline__ int tr_off256(int row, int ch) {
  return (ch >> 4) * (32 * 256) + 256 * row +
         16 * ((ch & 15) ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}
__device__ __forceinline__ bf16x8 tr_frag256(const char* img, int t, int lane) {
  const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
  const int a0 = tr_off256(4 * g + q, 2 * t + (p >> 1)) + 8 * (p & 1);
  const int a1 = tr_off256(16 + 4 * g + q, 2 * t + (p >> 1)) + 8 * (p & 1);
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + a0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + a1));
  const bf16x4 l4 = __builtin_bit_cast(bf16x4, lo), h4 = __builtin_bit_cast(bf16x4, hi);
  bf16x8 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) { r[e] = l4[e]; r[4 + e] = h4[e]; }
  return r;
}

// One workgroup = one row range of one job; 8 waves, wave w owns the [64 x 128] output block
// (G features 64 (w >> 1) .., A features 128 (w & 1) ..): 4 + 8 transposed fragments feed 32 MFMAs
// per 32-row tile.  The fp32 block leaves as a slab ([nwg][D][D]); k_wgrad256_sum adds the slabs
// of a job into dW in a fixed order (no atomics: the result is reproducible run to run).
__device__ __forceinline__ bf16x8 ld8(const __bf16* p) {
  return *reinterpret_cast<const bf16x8*>(p);
}
__device__ __forceinline__ bf16x8 ld8(const float* p) {
  const float4 lo = reinterpret_cast<const float4*>(p)[0], hi = reinterpret_cast<const float4*>(p)[1];
  bf16x8 v;
  v[0] = (__bf16)lo.x; v[1] = (__bf16)lo.y; v[2] = (__bf16)lo.z; v[3] = (__bf16)lo.w;
  v[4] = (__bf16)hi.x; v[5] = (__bf16)hi.y; v[6] = (__bf16)hi.z; v[7] = (__bf16)hi.w;
  return v;
}

// T: element type of G and A in memory (bf16, or fp32 rounded to bf16 while staged: the [B*m]-row
// reductions of the per-set epilogues).  Two tiles are in flight per thread (registers) while a
// third is consumed from LDS: one 32-row tile (32 KiB) ahead per CU was latency-bound at 2.6 TB/s.
template <typename T>
__global__ __launch_bounds__(512, 2) void k_wgrad256(const Wgrad256Jobs jobs, int rows_per_wg,
                                                    float* __restrict__ slabs,
                                                    float* __restrict__ bslabs) {
  constexpr int D = 256, NT = 512, TB = 32 * D * 2;      // bytes of one 32-row tile
  __shared__ __attribute__((aligned(16))) char lds[4 * TB];     // 2 buffers x (G, A)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  // 1-D grid of nwg * jobs.n workgroups.  When the jobs share an operand (dKp^T X and dVp^T X), the
  // jobs of one row block sit eight linear ids apart, i.e. on the same XCD back to back, and read its
  // tiles from that XCD's L2 instead of twice from memory (measured 250 -> 198 us for the pair; for
  // jobs with nothing in common the same order costs 4 %, so they keep job-major ids).
  const int nwg = gridDim.x / jobs.n;
  int bx, by;
  if (nwg % 8 == 0 && rows_per_wg < 0) {         // (rows_per_wg < 0: the host's "jobs share an operand")
    const int g8 = blockIdx.x >> 3, l8 = blockIdx.x & 7;
    by = g8 % jobs.n;
    bx = (g8 / jobs.n) * 8 + l8;
  } else {
    by = blockIdx.x / nwg;
    bx = blockIdx.x - by * nwg;
  }
  const Wgrad256Job job = jobs.j[by];
  const T* __restrict__ G = reinterpret_cast<const T*>(job.G);
  const T* __restrict__ A = reinterpret_cast<const T*>(job.A);
  // 64-row blocks are dealt round-robin: the workgroups that run together read neighbouring
  // addresses (one contiguous range per workgroup puts all of them 512 KiB apart, on the same few
  // HBM channels at the same moment)
  (void)rows_per_wg;
  const int64_t r0 = (int64_t)bx * 64, r1 = job.M, stride = (int64_t)nwg * 64;
  const int gt0 = 4 * (wave >> 1), at0 = 8 * (wave & 1);
  f32x4 acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  bf16x8 vg[2][2], va[2][2];               // [ring slot][piece]
  auto fetch = [&](int64_t base, bf16x8 (&g2)[2], bf16x8 (&a2)[2]) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int c = tid + e * NT, row = c >> 5, ch = c & 31;
      // (unconditional loads, rows past the end zeroed afterwards: see k_wgrad_small256)
      const bool ok = base + row < r1;
      const int64_t rc = ok ? base + row : r1 - 1;
      g2[e] = ld8(G + rc * D + ch * 8);
      a2[e] = ld8(A + rc * D + ch * 8);
      if (!ok) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { g2[e][k] = (__bf16)0.f; a2[e][k] = (__bf16)0.f; }
      }
    }
  };
  auto consume = [&](int buf, bf16x8 (&g2)[2], bf16x8 (&a2)[2], int64_t refill) {
    char* sG = lds + buf * 2 * TB;
    char* sA = sG + TB;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int c = tid + e * NT, row = c >> 5, ch = c & 31;
      *reinterpret_cast<bf16x8*>(sG + tr_off256(row, ch)) = g2[e];
      *reinterpret_cast<bf16x8*>(sA + tr_off256(row, ch)) = a2[e];
      if (job.db != nullptr) {
#pragma unroll
        for (int k = 0; k < 8; ++k) bs[k] += (float)g2[e][k];
      }
    }
    __syncthreads();
    if (refill < r1) fetch(refill, g2, a2);
    bf16x8 ga[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) ga[i] = tr_frag256(sG, gt0 + i, lane);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const bf16x8 ab = tr_frag256(sA, at0 + t, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][t] = mfma32(ga[i], ab, acc[i][t]);
    }
  };
  if (r0 < r1) fetch(r0, vg[0], va[0]);
  if (r0 + 32 < r1) fetch(r0 + 32, vg[1], va[1]);
  for (int64_t base = r0; base < r1; base += stride) {
    consume(0, vg[0], va[0], base + stride);
    if (base + 32 < r1) consume(1, vg[1], va[1], base + stride + 32);
  }
  // slab layout [job][output row][workgroup][256]: the partial sums of one output row lie next to
  // each other, so the summing pass streams 1 KiB x nwg contiguous bytes per row (with one
  // [256][256] block per workgroup it read 1 KiB out of every 256 KiB: 1.2 TB/s)
  float* slab = slabs + (int64_t)by * D * nwg * D + (int64_t)bx * D;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int grow = 16 * (gt0 + i) + 4 * g + e;
#pragma unroll
      for (int t = 0; t < 8; ++t)
        slab[(int64_t)grow * nwg * D + 16 * (at0 + t) + r] = acc[i][t][e];
    }
  if (job.db != nullptr) {
    // threads with equal (tid & 31) hold partial sums of the same 8 columns
    __syncthreads();
    float* red = reinterpret_cast<float*>(lds);             // [16][256]
#pragma unroll
    for (int k = 0; k < 8; ++k) red[(tid >> 5) * D + (tid & 31) * 8 + k] = bs[k];
    __syncthreads();
    if (tid < D) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) t += red[q * D + tid];
      bslabs[((int64_t)by * nwg + bx) * D + tid] = t;
    }
  }
}

// Round 3: the bf16 jobs with the operand tiles streamed by LDS-DMA into a ring of four (G, A) tile
// pairs (128 KiB), three pairs ahead.  k_wgrad256 above keeps two 32-row tiles in registers per
// thread - 64 KiB in flight per CU, and 3.0 TB/s is what that bought by Little's law at the
// latency this streaming pattern sees; its MFMA work would sustain 16 TB/s.  Here 96 KiB are in
// flight, nothing is staged through registers, and the barrier per tile only publishes pieces
// that have already landed.  A 1 KiB piece of the transposed-read layout is 4 rows x 256 bytes
// of one [32][128] half: LDS is written linearly, so lane l fetches chunk (l & 15) ^ s(row) of row
// 4 p + (l >> 4) - the swizzle is an involution, the permutation moves to the source side.
__global__ __launch_bounds__(512, 2) void k_wgrad256_dma(const Wgrad256Jobs jobs, int rows_per_wg,
                                                        float* __restrict__ slabs,
                                                        float* __restrict__ bslabs) {
  constexpr int D = 256, TB = 32 * D * 2, NB = 4, PD = NB - 1;
  extern __shared__ __attribute__((aligned(16))) char lds[];    // [NB][G tile | A tile]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int nwg = gridDim.x / jobs.n;
  int bx, by;
  if (nwg % 8 == 0 && rows_per_wg < 0) {
    const int g8 = blockIdx.x >> 3, l8 = blockIdx.x & 7;
    by = g8 % jobs.n;
    bx = (g8 / jobs.n) * 8 + l8;
  } else {
    by = blockIdx.x / nwg;
    bx = blockIdx.x - by * nwg;
  }
  const Wgrad256Job job = jobs.j[by];
  const char* G = reinterpret_cast<const char*>(job.G);
  const char* A = reinterpret_cast<const char*>(job.A);
  const int64_t r0 = (int64_t)bx * 64, r1 = job.M, stride = (int64_t)nwg * 64;
  // tiles of this workgroup: 64-row blocks dealt round-robin, two 32-row tiles per block
  const int64_t nblk = r0 < r1 ? (r1 - r0 + stride - 1) / stride : 0;
  const int ntile = (int)(2 * nblk);
  auto tile_row = [&](int t) { return r0 + (int64_t)(t >> 1) * stride + 32 * (t & 1); };
  const int gt0 = 4 * (wave >> 1), at0 = 8 * (wave & 1);
  f32x4 acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // this wave's four pieces of a tile pair: pieces 2 wave, 2 wave + 1 of G and of A (piece p:
  // half p >> 3, rows 4 (p & 7) ..); per lane the row inside the tile and the source byte offset
  int prow[2], poff[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int p = 2 * wave + e, half = p >> 3, row = 4 * (p & 7) + (lane >> 4);
    const int sw = ((row & 3) << 2) | ((row >> 2) & 3);
    prow[e] = row;
    poff[e] = half * 256 + (((lane & 15) ^ sw) << 4);
  }
  // job.mask: the 256 ReLU-mask words of the tile's 32 rows (1 KiB, contiguous) ride along as a fifth
  // piece of wave 0 into the mask ring behind the tile ring
  const bool masked = job.mask != nullptr;
  const bool mask_wave = masked && wave == 0;
  char* sMaskRing = lds + NB * 2 * TB;
  auto dma = [&](int t) {
    const int64_t base = tile_row(t);
    char* dst = lds + (t % NB) * 2 * TB;
#pragma unroll
    for (int op = 0; op < 2; ++op)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        int64_t row = base + prow[e];
        row = row < r1 ? row : r1 - 1;                 // rows past the end: a valid line (zeroed below)
        const char* src = (op == 0 ? G : A) + row * (D * 2) + poff[e];
        const unsigned ldst = __builtin_amdgcn_readfirstlane(
            (unsigned)(uintptr_t)(lds_void_t*)(dst + op * TB + (2 * wave + e) * 1024));
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(ldst) : "memory");
      }
    if (mask_wave) {
      const char* src = reinterpret_cast<const char*>(job.mask) + base * 32 + lane * 16;
      const unsigned ldst = __builtin_amdgcn_readfirstlane(
          (unsigned)(uintptr_t)(lds_void_t*)(sMaskRing + (t % NB) * 1024));
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                   "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(src), "s"(ldst) : "memory");
    }
  };
#pragma unroll 1
  for (int t = 0; t < PD && t < ntile; ++t) dma(t);
#pragma unroll 1
  for (int t = 0; t < ntile; ++t) {
    // tile t has landed when at most the pieces of the (up to PD - 1) tiles behind it are pending
    {
      const int ahead = (ntile - 1 - t) < (PD - 1) ? (ntile - 1 - t) : (PD - 1);
      if (mask_wave) {                                 // (five pieces per tile in this wave)
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                      // every wave's pieces; tile t - 1 consumed
    asm volatile("" ::: "memory");
    if (t + PD < ntile) dma(t + PD);                   // into the buffer tile t - 1 just left
    char* sG = lds + (t % NB) * 2 * TB;
    char* sA = sG + TB;
    const int64_t base = tile_row(t);
    if (base + 32 > r1) {                              // (uniform; the last tile of the job only)
      const int c0 = tid, row0 = c0 >> 5;              // rows past the end contribute nothing
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int c = tid + e * 512, row = c >> 5, ch = c & 31;
        if (base + row >= r1) *reinterpret_cast<uint4*>(sG + tr_off256(row, ch)) = uint4{0u, 0u, 0u, 0u};
      }
      (void)row0;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    if (masked) {
      // G . [mask] in place: thread = (row, 8 features) as below; the features 8 ch .. 8 ch + 7 of a row
      // are two nibbles of the forward's layout (word = 16-row block x head half x lane (r, g), byte =
      // head, bit 4 t + e  <->  feature 32 j + 16 t + 4 g + e): lanes g0 = 2 (ch & 1) and g0 + 1
      const uint32_t* sM = reinterpret_cast<const uint32_t*>(sMaskRing + (t % NB) * 1024);
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int c = tid + e * 512, row = c >> 5, ch = c & 31;
        const int wi = ((row >> 4) * 2 + (ch >> 4)) * 64 + (row & 15) + 32 * (ch & 1);
        const int sh = 8 * ((ch >> 2) & 3) + 4 * ((ch >> 1) & 1);
        const uint32_t n0 = sM[wi] >> sh, n1 = sM[wi + 16] >> sh;
        bf16x8 gv = *reinterpret_cast<const bf16x8*>(sG + tr_off256(row, ch));
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (!((n0 >> k) & 1u)) gv[k] = (__bf16)0.f;
          if (!((n1 >> k) & 1u)) gv[4 + k] = (__bf16)0.f;
        }
        *reinterpret_cast<bf16x8*>(sG + tr_off256(row, ch)) = gv;
        if (job.db != nullptr) {
#pragma unroll
          for (int k = 0; k < 8; ++k) bs[k] += (float)gv[k];
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    } else if (job.db != nullptr) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int c = tid + e * 512, row = c >> 5, ch = c & 31;
        const bf16x8 gv = *reinterpret_cast<const bf16x8*>(sG + tr_off256(row, ch));
#pragma unroll
        for (int k = 0; k < 8; ++k) bs[k] += (float)gv[k];
      }
    }
    bf16x8 ga[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) ga[i] = tr_frag256(sG, gt0 + i, lane);
#pragma unroll
    for (int tt = 0; tt < 8; ++tt) {
      const bf16x8 ab = tr_frag256(sA, at0 + tt, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][tt] = mfma32(ga[i], ab, acc[i][tt]);
    }
  }
  float* slab = slabs + (int64_t)by * D * nwg * D + (int64_t)bx * D;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int grow = 16 * (gt0 + i) + 4 * g + e;
#pragma unroll
      for (int t = 0; t < 8; ++t)
        slab[(int64_t)grow * nwg * D + 16 * (at0 + t) + r] = acc[i][t][e];
    }
  if (job.db != nullptr) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(lds);             // [16][256]
#pragma unroll
    for (int k = 0; k < 8; ++k) red[(tid >> 5) * D + (tid & 31) * 8 + k] = bs[k];
    __syncthreads();
    if (tid < D) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) t += red[q * D + tid];
      bslabs[((int64_t)by * nwg + bx) * D + tid] = t;
    }
  }
}

__global__ __launch_bounds__(256) void k_wgrad256_sum(const Wgrad256Jobs jobs, int nwg,
                                                     int rows_per_wg,
                                                     const float* __restrict__ slabs,
                                                     const float* __restrict__ bslabs) {
  constexpr int D = 256;
  const Wgrad256Job job = jobs.j[blockIdx.y];
  (void)rows_per_wg;
  const int64_t blocks = (job.M + 63) / 64;
  const int used = (int)(blocks < nwg ? blocks : nwg);                  // slabs with rows
  // blockIdx.x < 256: output row of dW; == 256: the bias sums (same walk over [nwg][256] partials -
  // as a plain loop of one load per slab in one workgroup it was 256 dependent round trips, 90 us)
  const bool bias_row = blockIdx.x == D;
  if (bias_row && job.db == nullptr) return;
  {
    // 256 outputs per workgroup: lane group sg adds slabs sg, sg + 4, ... of four consecutive
    // outputs (16-byte loads, eight in flight), then the four groups are added in a fixed order
    __shared__ float4 red[4][64];
    const int sg = threadIdx.x >> 6, c = threadIdx.x & 63;
    const float* s = bias_row ? bslabs + (int64_t)blockIdx.y * nwg * D + 4 * c
                              : slabs + ((int64_t)blockIdx.y * D + blockIdx.x) * nwg * D + 4 * c;
    float* out = bias_row ? job.db + 4 * c : job.dW + blockIdx.x * 256 + 4 * c;
    float4 t = {0.f, 0.f, 0.f, 0.f};
    int w = sg;
    for (; w + 28 < used; w += 32) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        v[u] = *reinterpret_cast<const float4*>(s + (int64_t)(w + 4 * u) * D);
#pragma unroll
      for (int u = 0; u < 8; ++u) { t.x += v[u].x; t.y += v[u].y; t.z += v[u].z; t.w += v[u].w; }
    }
    for (; w < used; w += 4) {
      const float4 v = *reinterpret_cast<const float4*>(s + (int64_t)w * D);
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    red[sg][c] = t;
    __syncthreads();
    if (sg == 0) {
      float4 o = *reinterpret_cast<float4*>(out);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = red[q][c];
        o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
      }
      *reinterpret_cast<float4*>(out) = o;
    }
  }
}

// out[i] (+)= sum_s slabs[s][i], i < n (n % 4 == 0, 16-byte aligned): the fixed-order replacement of
// fp32 atomics wherever workgroups reduce into one small tensor (same walk as k_wgrad256_sum: four
// lane groups take every fourth slab, eight 16-byte loads in flight, then a fixed-order merge)
__global__ __launch_bounds__(256) void k_slab_sum(const float* __restrict__ slabs, int S, int n,
                                                 float* __restrict__ out, int accumulate) {
  __shared__ float4 red[4][64];
  const int sg = threadIdx.x >> 6, c = threadIdx.x & 63;
  const int i = blockIdx.x * 256 + 4 * c;
  float4 t = {0.f, 0.f, 0.f, 0.f};
  if (i < n) {
    const float* s = slabs + i;
    int w = sg;
    for (; w + 28 < S; w += 32) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(s + (int64_t)(w + 4 * u) * n);
#pragma unroll
      for (int u = 0; u < 8; ++u) { t.x += v[u].x; t.y += v[u].y; t.z += v[u].z; t.w += v[u].w; }
    }
    for (; w < S; w += 4) {
      const float4 v = *reinterpret_cast<const float4*>(s + (int64_t)w * n);
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
  }
  red[sg][c] = t;
  __syncthreads();
  if (sg == 0 && i < n) {
    float4 o = accumulate ? *reinterpret_cast<float4*>(out + i) : float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = red[q][c];
      o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
    }
    *reinterpret_cast<float4*>(out + i) = o;
  }
}

// =====================================================================================
// few-queries attention (ISAB mab0 at dk = 256) over projected keys: wave = head, head dim 32
// =====================================================================================
struct FqArgs {
  const __bf16 *Kp, *Vp;     // [B*N][D]
  const float* Qp;           // [m][D] shared projected query (fp32)
  // forward
  float *Op, *Mp, *Lp;       // [B][S][m][D] unnormalised sum_n 2^(s-M) Vp ; [B][S][H][MQ] M, L
  // backward
  const float* dOa;          // [B][m][D] gradient w.r.t. A Vp (= dO)
  const float* LSE;          // [B][H][MQ] log2-domain
  const float* Delta;        // [B][H][MQ]
  __bf16 *dKp, *dVp;         // [B*N][D]
  float* dQpPart;            // [B][S][m][D]
  int B, N, m, S;
  float scale_log2e, scale;  // log2(e)/sqrt(d) ; 1/sqrt(d)
  const int32_t* lengths;
};

// shared query of head j as MFMA operands: element (q, f) = Qp[q][32 j + f] * mul
// as B operand [k = f][col = q] / as A operand [row = q][k = f]: the same registers
__device__ __forceinline__ bf16x8 q_frag(const float* Qp, int D, int m, int q, int j, int g,
                                         float mul) {
  bf16x8 v;
  if (q < m) {
    const float4 lo = *reinterpret_cast<const float4*>(Qp + (int64_t)q * D + 32 * j + 8 * g);
    const float4 hi = *reinterpret_cast<const float4*>(Qp + (int64_t)q * D + 32 * j + 8 * g + 4);
    v[0] = (__bf16)(lo.x * mul); v[1] = (__bf16)(lo.y * mul); v[2] = (__bf16)(lo.z * mul);
    v[3] = (__bf16)(lo.w * mul); v[4] = (__bf16)(hi.x * mul); v[5] = (__bf16)(hi.y * mul);
    v[6] = (__bf16)(hi.z * mul); v[7] = (__bf16)(hi.w * mul);
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (__bf16)0.f;
  }
  return v;
}

// forward: S^T[pt][q] = Kp_h[pt][:] . (sl2e Qp_h[q][:]) on the MFMA with the points on the
// accumulator rows (a query's statistics: in-lane + 2 cross-lane steps), online softmax,
// O^T[f][q] += Vp_h^T[f][pt] P^T[pt][q] with the probability tile as B operand straight from the
// accumulators and Vp^T through a wave-private LDS tile + ds_read_tr16_b64.
template <int D, int QT>          // QT = query tiles of 16 (m <= 16 QT)
__global__ __launch_bounds__(64 * (D / 32)) void k_fq_attn_fwd(const FqArgs a) {
  constexpr int PV = 72, H = D / 32, MQ = 16 * QT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, j = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int b = blockIdx.x, sp = blockIdx.y;
  char* myV = smem + j * 32 * PV;
  const int per = (int)(((int64_t)(a.N + 31) / 32 + a.S - 1) / a.S) * 32;
  int len = a.N;
  if (a.lengths != nullptr) len = a.lengths[b] < a.N ? a.lengths[b] : a.N;
  const int n_lo = sp * per, n_hi = (n_lo + per < len) ? n_lo + per : len;
  bf16x8 qf[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) qf[qt] = q_frag(a.Qp, D, a.m, 16 * qt + r, j, g, a.scale_log2e);
  float mrow[QT], lrow[QT];
  f32x4 ot[2][QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    mrow[qt] = -INFINITY;
    lrow[qt] = 0.f;
    ot[0][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
    ot[1][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (int n0 = n_lo; n0 < n_hi; n0 += 32) {
    bf16x8 kr[2];
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) {
      const int n = n0 + 16 * pb + r;
      bf16x8 vr;
      {   // unconditional loads (last row for points past the range), zeroed afterwards: a load
          // under a divergent `if` is waited for with vmcnt(0) before the next one is issued
        const int64_t o = ((int64_t)b * a.N + (n < n_hi ? n : n_hi - 1)) * D + 32 * j + 8 * g;
        kr[pb] = *reinterpret_cast<const bf16x8*>(a.Kp + o);
        vr = *reinterpret_cast<const bf16x8*>(a.Vp + o);
        if (n >= n_hi) {
#pragma unroll
          for (int e = 0; e < 8; ++e) { kr[pb][e] = (__bf16)0.f; vr[e] = (__bf16)0.f; }
        }
      }
      {   // (two 8-byte stores: the padded 72-byte pitch is not 16-byte aligned)
        bf16x4 lo4, hi4;
#pragma unroll
        for (int e = 0; e < 4; ++e) { lo4[e] = vr[e]; hi4[e] = vr[4 + e]; }
        *reinterpret_cast<bf16x4*>(myV + (16 * pb + r) * PV + 16 * g) = lo4;
        *reinterpret_cast<bf16x4*>(myV + (16 * pb + r) * PV + 16 * g + 8) = hi4;
      }
    }
    bf16x8 vt[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) vt[tt] = tr_frag_small(myV, PV, 16 * tt, lane);
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      f32x4 s[2];
      float mt = -INFINITY;
#pragma unroll
      for (int pb = 0; pb < 2; ++pb) {
        s[pb] = mfma32(kr[pb], qf[qt], f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (n0 + 16 * pb + 4 * g + e >= n_hi) s[pb][e] = -INFINITY;
          mt = fmaxf(mt, s[pb][e]);
        }
      }
      mt = wave16_max(mt);
      const float mnew = fmaxf(mrow[qt], mt);           // finite: the tile has >= 1 live point
      const float alpha = __builtin_amdgcn_exp2f(mrow[qt] - mnew);
      float ls = 0.f;
#pragma unroll
      for (int pb = 0; pb < 2; ++pb)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s[pb][e] = __builtin_amdgcn_exp2f(s[pb][e] - mnew);
          ls += s[pb][e];
        }
      ls = wave16_sum(ls);
      lrow[qt] = lrow[qt] * alpha + ls;
      mrow[qt] = mnew;
      const bf16x8 pb8 = pack8(s[0], s[1]);
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
        for (int e = 0; e < 4; ++e) ot[tt][qt][e] *= alpha;     // column q = this lane's query
        ot[tt][qt] = mfma32(vt[tt], pb8, ot[tt][qt]);
      }
    }
  }
  // partial (O^T, M, L) of this range: rows f = 16 tt + 4 g + e, column q = 16 qt + r
  const int64_t pb0 = ((int64_t)b * a.S + sp);
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int q = 16 * qt + r;
    if (q < a.m) {
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
        *reinterpret_cast<float4*>(a.Op + (pb0 * a.m + q) * D + 32 * j + 16 * tt + 4 * g) =
            float4{ot[tt][qt][0], ot[tt][qt][1], ot[tt][qt][2], ot[tt][qt][3]};
      if (g == 0) {
        a.Mp[(pb0 * H + j) * MQ + q] = mrow[qt];
        a.Lp[(pb0 * H + j) * MQ + q] = lrow[qt];
      }
    }
  }
}

// k_fq_attn_fwd2 (m = 32, d = 256): the forward with full-line reads - the workgroup streams whole
// [32 keys][256] tiles of Kp and Vp in by LDS-DMA (double buffered) and every wave (= head) takes its
// 64-byte slices from LDS, where k_fq_attn_fwd reads 16 rows x 64 bytes per load instruction.
__global__ __launch_bounds__(512, 2) void k_fq_attn_fwd2(const FqArgs a) {
  constexpr int D = 256, QT = 2, PV = 72, H = D / 32, MQ = 16 * QT;
  constexpr int ROWB = D * 2, TILEB = 32 * ROWB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sKb = smem;                        // [2][TILEB]
  char* sVb = smem + 2 * TILEB;            // [2][TILEB]
  const int tid = threadIdx.x, lane = tid & 63;
  const int j = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int b = blockIdx.x, sp = blockIdx.y;
  char* myV = smem + 4 * TILEB + j * 32 * PV;
  const int per = (int)(((int64_t)(a.N + 31) / 32 + a.S - 1) / a.S) * 32;
  int len = a.N;
  if (a.lengths != nullptr) len = a.lengths[b] < a.N ? a.lengths[b] : a.N;
  const int n_lo = sp * per, n_hi = (n_lo + per < len) ? n_lo + per : len;
  const int T = n_lo < n_hi ? (n_hi - n_lo + 31) / 32 : 0;
  bf16x8 qf[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) qf[qt] = q_frag(a.Qp, D, a.m, 16 * qt + r, j, g, a.scale_log2e);
  float mrow[QT], lrow[QT];
  f32x4 ot[2][QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    mrow[qt] = -INFINITY;
    lrow[qt] = 0.f;
    ot[0][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
    ot[1][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int oK = swz(r, 4 * j + g, ROWB);
  auto lds_barrier = [] {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  auto dma = [&](int k) {
    const int n0 = n_lo + 32 * k, par = k & 1;
#pragma unroll
    for (int w = 0; w < 2; ++w) {
      const __bf16* base = w == 0 ? a.Kp : a.Vp;
      char* dst = (w == 0 ? sKb : sVb) + par * TILEB;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int p = (2 * j + i) * 64 + lane;
        const int row = p >> 5, slot = p & 31;
        const int ch = (slot & ~15) | ((slot ^ row) & 15);
        const int n = n0 + row < a.N ? n0 + row : a.N - 1;
        const __bf16* src = base + ((int64_t)b * a.N + n) * D + ch * 8;
        const unsigned ldst = __builtin_amdgcn_readfirstlane(
            (unsigned)(uintptr_t)(lds_void_t*)(dst + (2 * j + i) * 1024));
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(ldst) : "memory");
      }
    }
  };
  if (T > 0) dma(0);
  for (int k = 0; k < T; ++k) {
    const int par = k & 1, n0 = n_lo + 32 * k;
    const char* sK = sKb + par * TILEB;
    const char* sV = sVb + par * TILEB;
    if (k + 1 < T) {
      dma(k + 1);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // everything but the 4 pieces just issued
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    lds_barrier();                       // tiles of iteration k complete (and k - 1 consumed by all)
    bf16x8 kr[2];
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) {
      const int n = n0 + 16 * pb + r;
      kr[pb] = *reinterpret_cast<const bf16x8*>(sK + oK + 8192 * pb);
      bf16x8 vr = *reinterpret_cast<const bf16x8*>(sV + oK + 8192 * pb);
      if (n >= n_hi) {                     // rows past the range: the DMA fetched other rows
#pragma unroll
        for (int e = 0; e < 8; ++e) { kr[pb][e] = (__bf16)0.f; vr[e] = (__bf16)0.f; }
      }
      bf16x4 lo4, hi4;
#pragma unroll
      for (int e = 0; e < 4; ++e) { lo4[e] = vr[e]; hi4[e] = vr[4 + e]; }
      *reinterpret_cast<bf16x4*>(myV + (16 * pb + r) * PV + 16 * g) = lo4;
      *reinterpret_cast<bf16x4*>(myV + (16 * pb + r) * PV + 16 * g + 8) = hi4;
    }
    bf16x8 vt[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) vt[tt] = tr_frag_small(myV, PV, 16 * tt, lane);
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      f32x4 s[2];
      float mt = -INFINITY;
#pragma unroll
      for (int pb = 0; pb < 2; ++pb) {
        s[pb] = mfma32(kr[pb], qf[qt], f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (n0 + 16 * pb + 4 * g + e >= n_hi) s[pb][e] = -INFINITY;
          mt = fmaxf(mt, s[pb][e]);
        }
      }
      mt = wave16_max(mt);
      const float mnew = fmaxf(mrow[qt], mt);           // finite: the tile has >= 1 live point
      const float alpha = __builtin_amdgcn_exp2f(mrow[qt] - mnew);
      float ls = 0.f;
#pragma unroll
      for (int pb = 0; pb < 2; ++pb)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s[pb][e] = __builtin_amdgcn_exp2f(s[pb][e] - mnew);
          ls += s[pb][e];
        }
      ls = wave16_sum(ls);
      lrow[qt] = lrow[qt] * alpha + ls;
      mrow[qt] = mnew;
      const bf16x8 pb8 = pack8(s[0], s[1]);
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
        for (int e = 0; e < 4; ++e) ot[tt][qt][e] *= alpha;
        ot[tt][qt] = mfma32(vt[tt], pb8, ot[tt][qt]);
      }
    }
  }
  const int64_t pb0 = ((int64_t)b * a.S + sp);
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int q = 16 * qt + r;
    if (q < a.m) {
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
        *reinterpret_cast<float4*>(a.Op + (pb0 * a.m + q) * D + 32 * j + 16 * tt + 4 * g) =
            float4{ot[tt][qt][0], ot[tt][qt][1], ot[tt][qt][2], ot[tt][qt][3]};
      if (g == 0) {
        a.Mp[(pb0 * H + j) * MQ + q] = mrow[qt];
        a.Lp[(pb0 * H + j) * MQ + q] = lrow[qt];
      }
    }
  }
}

// k_fq_proj_fwd (m = 32, d = 256): fc_k / fc_v of the N keys AND the few-queries attention over
// them in one pass over X.  Per 32-point tile the workgroup computes the Kp / Vp tiles as
// k_rowstream<1,2> does (wave j: output features 32 j .. 32 j + 31 of both, the two [32 x 256]
// weight slices in 128 registers, X tiles by LDS-DMA three deep), every wave writes its slices to
// the Kp / Vp tiles in LDS - and head j's attention needs exactly the slices wave j just wrote, so
// it reads them back without a barrier and runs k_fq_attn_fwd2's body on them; after one barrier
// the two tiles leave for the backward in full rows.  Against PROJ2 + k_fq_attn_fwd2 the Kp / Vp
// tensors are written but never read back: two [B*N, 256] passes less.
struct FqProjArgs {
  FqArgs f;
  const __bf16* X;               // [B*N][256]
  const __bf16 *WkB, *WvB;       // natural bf16 images [256][256] (F8: fp8 e4m3 bytes of s * W)
  const float *bk, *bv;
  __bf16 *KpO, *VpO;             // [B*N][256] outputs (saved for the backward)
  const float* inv_scale;        // F8: 1 / s_k, 1 / s_v
};
// F8 (PCA_MODE_FP8, round 3): fc_k / fc_v on v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 operands, block
// scales 2^0: twice the bf16 rate, scripts/probe/mfma_f8_probe.hip).  The X tile is converted to
// fp8 ONCE per tile by the whole workgroup (16 elements per thread, into an 8 KiB fp8 tile with
// 16-byte chunks XOR-swizzled by row) instead of once per fragment and wave as k_rowstream<F8>
// does (28 VALU instructions per fragment, eight waves converting the same values); the weight
// slices are 32 + 32 registers instead of 64 + 64 and a tile's 64 MFMAs of 16 cycles become 16 of
// 32.  The attention on the bf16 Kp / Vp slices is unchanged.
typedef int fq_v8i __attribute__((ext_vector_type(8)));
template <bool F8>
__global__ __launch_bounds__(512, 2) void k_fq_proj_fwd(const FqProjArgs aa) {
  const FqArgs& a = aa.f;
  constexpr int D = 256, QT = 2, PV = 72, H = D / 32, MQ = 16 * QT, KS = D / 32;
  constexpr int ROWB = D * 2, TILEB = 32 * ROWB, NBUF = 3, PD = NBUF - 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sXb = smem;                        // [NBUF][TILEB]
  char* sK = smem + NBUF * TILEB;          // Kp tile
  char* sV = sK + TILEB;                   // Vp tile
  char* sX8 = sV + TILEB + H * 32 * PV;    // F8: the current X tile as fp8 [32][256 B]
  const int tid = threadIdx.x, lane = tid & 63;
  const int j = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int b = blockIdx.x, sp = blockIdx.y;
  char* myV = sV + TILEB + j * 32 * PV;
  const int per = (int)(((int64_t)(a.N + 31) / 32 + a.S - 1) / a.S) * 32;
  int len = a.N;
  if (a.lengths != nullptr) len = a.lengths[b] < a.N ? a.lengths[b] : a.N;
  // the projections cover every row of the range (the backward reads padded rows too); the
  // attention only the keys: rows below min(range end, len)
  const int n_lo = sp * per;
  const int n_end = (n_lo + per < a.N) ? n_lo + per : a.N;
  const int n_hi = n_end < len ? n_end : len;
  const int T = n_lo < n_end ? (n_end - n_lo + 31) / 32 : 0;
  bf16x8 wk[F8 ? 1 : KS][2], wv[F8 ? 1 : KS][2];
  fq_v8i wk8[F8 ? 2 : 1][2], wv8[F8 ? 2 : 1][2];
  float inv_k = 1.f, inv_v = 1.f;
  if (F8) {
    inv_k = aa.inv_scale[0];
    inv_v = aa.inv_scale[1];
#pragma unroll
    for (int S = 0; S < 2; ++S)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int o = (32 * j + 16 * t + r) * D + 128 * S + 32 * g;
        const uint4* pk = reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(aa.WkB) + o);
        const uint4* pv = reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(aa.WvB) + o);
        const uint4 k0 = pk[0], k1 = pk[1], v0 = pv[0], v1 = pv[1];
        wk8[F8 ? S : 0][t] = fq_v8i{(int)k0.x, (int)k0.y, (int)k0.z, (int)k0.w,
                                    (int)k1.x, (int)k1.y, (int)k1.z, (int)k1.w};
        wv8[F8 ? S : 0][t] = fq_v8i{(int)v0.x, (int)v0.y, (int)v0.z, (int)v0.w,
                                    (int)v1.x, (int)v1.y, (int)v1.z, (int)v1.w};
      }
  } else {
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int64_t o = (int64_t)(32 * j + 16 * t + r) * D + 32 * s + 8 * g;
        wk[F8 ? 0 : s][t] = *reinterpret_cast<const bf16x8*>(aa.WkB + o);
        wv[F8 ? 0 : s][t] = *reinterpret_cast<const bf16x8*>(aa.WvB + o);
      }
  }
  // F8: this thread's share of the tile conversion (row tid / 16, fp8 chunk tid % 16 = bf16 chunks
  // 2 c, 2 c + 1) and this lane's two 16-byte chunks of an fp8 row (k = 32 g .. 32 g + 31; + 128 S)
  const int cvr = tid >> 4, cvc = tid & 15;
  const int cv_src0 = swz(cvr, 2 * cvc, ROWB), cv_src1 = swz(cvr, 2 * cvc + 1, ROWB);
  const int cv_dst = cvr * D + ((cvc ^ (cvr & 15)) << 4);
  int x8r[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) x8r[c] = r * D + (((2 * g + c) ^ r) << 4);
  f32x4 bkz[2], bvz[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const float4 k4 = *reinterpret_cast<const float4*>(aa.bk + 32 * j + 16 * t + 4 * g);
    const float4 v4 = *reinterpret_cast<const float4*>(aa.bv + 32 * j + 16 * t + 4 * g);
    bkz[t] = f32x4{k4.x, k4.y, k4.z, k4.w};
    bvz[t] = f32x4{v4.x, v4.y, v4.z, v4.w};
  }
  bf16x8 qf[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) qf[qt] = q_frag(a.Qp, D, a.m, 16 * qt + r, j, g, a.scale_log2e);
  float mrow[QT], lrow[QT];
  f32x4 ot[2][QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    mrow[qt] = -INFINITY;
    lrow[qt] = 0.f;
    ot[0][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
    ot[1][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  int oB[4], oD[2];
#pragma unroll
  for (int k = 0; k < 4; ++k) oB[k] = swz(r, 4 * k + g, ROWB);
#pragma unroll
  for (int t = 0; t < 2; ++t) oD[t] = swz(r, 4 * j + 2 * t + (g >> 1), ROWB) + 8 * (g & 1);
  const int oK = swz(r, 4 * j + g, ROWB);
  const int oC = swz(tid >> 5, tid & 31, ROWB);
  auto lds_barrier = [] {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  auto dma = [&](int k) {
    const int n0 = n_lo + 32 * k;
    char* dst = sXb + (k % NBUF) * TILEB;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int p = (2 * j + i) * 64 + lane;
      const int row = p >> 5, slot = p & 31;
      const int ch = (slot & ~15) | ((slot ^ row) & 15);
      const int n = n0 + row < a.N ? n0 + row : a.N - 1;
      const __bf16* src = aa.X + ((int64_t)b * a.N + n) * D + ch * 8;
      const unsigned ldst = __builtin_amdgcn_readfirstlane(
          (unsigned)(uintptr_t)(lds_void_t*)(dst + (2 * j + i) * 1024));
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                   "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(src), "s"(ldst) : "memory");
    }
  };
#pragma unroll 1
  for (int k = 0; k < PD && k < T; ++k) dma(k);
#pragma unroll 1
  for (int k = 0; k < T; ++k) {
    const int n0 = n_lo + 32 * k, nlive = a.N - n0;
    const char* sX = sXb + (k % NBUF) * TILEB;
    // tile k + PD streams in; tile k must have landed: younger are the DMAs of the tiles ahead (2
    // pieces each) and the 4 stores of each of the last PD tiles (all full: not the last of a range)
    if (k + PD < T) dma(k + PD);
    {
      const int ahead = (T - 1 - k) < PD ? (T - 1 - k) : PD;
      const int behind = k < PD ? k : PD;
      switch (2 * ahead + 4 * behind) {
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      }
    }
    lds_barrier();                       // B0: X tile k; the previous Kp / Vp tiles are stored
    if (F8) {
      const bf16x8 x0 = *reinterpret_cast<const bf16x8*>(sX + cv_src0);
      const bf16x8 x1 = *reinterpret_cast<const bf16x8*>(sX + cv_src1);
      uint4 q;
      q.x = cvt4_f8((float)x0[0], (float)x0[1], (float)x0[2], (float)x0[3]);
      q.y = cvt4_f8((float)x0[4], (float)x0[5], (float)x0[6], (float)x0[7]);
      q.z = cvt4_f8((float)x1[0], (float)x1[1], (float)x1[2], (float)x1[3]);
      q.w = cvt4_f8((float)x1[4], (float)x1[5], (float)x1[6], (float)x1[7]);
      *reinterpret_cast<uint4*>(sX8 + cv_dst) = q;
      lds_barrier();                     // B0b: the fp8 tile (the previous tile's readers passed B1)
    }
    // ---- Kp_h^T, Vp_h^T = W_h . X^T + b: own slices of the two tiles ----
    {
      f32x4 ak[2][2], av[2][2];
      if (F8) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          fq_v8i fb8[2];
#pragma unroll
          for (int S = 0; S < 2; ++S) {
            const uint4 lo = *reinterpret_cast<const uint4*>(sX8 + 16 * D * nb + (x8r[0] ^ (S << 7)));
            const uint4 hi = *reinterpret_cast<const uint4*>(sX8 + 16 * D * nb + (x8r[1] ^ (S << 7)));
            fb8[S] = fq_v8i{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w,
                            (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
          }
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            f32x4 k4 = {0.f, 0.f, 0.f, 0.f}, v4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int S = 0; S < 2; ++S) {
              k4 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wk8[F8 ? S : 0][t], fb8[S], k4, 0, 0,
                                                                   0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
              v4 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wv8[F8 ? S : 0][t], fb8[S], v4, 0, 0,
                                                                   0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              ak[t][nb][e] = __builtin_fmaf(k4[e], inv_k, bkz[t][e]);
              av[t][nb][e] = __builtin_fmaf(v4[e], inv_v, bvz[t][e]);
            }
          }
        }
      } else {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) { ak[t][nb] = bkz[t]; av[t][nb] = bvz[t]; }
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          const bf16x8 bx =
              *reinterpret_cast<const bf16x8*>(sX + oB[s & 3] + 256 * (s >> 2) + 8192 * nb);
          ak[0][nb] = mfma32(wk[F8 ? 0 : s][0], bx, ak[0][nb]);
          ak[1][nb] = mfma32(wk[F8 ? 0 : s][1], bx, ak[1][nb]);
          av[0][nb] = mfma32(wv[F8 ? 0 : s][0], bx, av[0][nb]);
          av[1][nb] = mfma32(wv[F8 ? 0 : s][1], bx, av[1][nb]);
        }
      }
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          *reinterpret_cast<bf16x4*>(sK + oD[t] + 8192 * nb) = pack4(ak[t][nb]);
          *reinterpret_cast<bf16x4*>(sV + oD[t] + 8192 * nb) = pack4(av[t][nb]);
        }
    }
    // ---- head j's attention on the slices this wave just wrote (k_fq_attn_fwd2's body) ----
    if (n0 < n_hi) {
      bf16x8 kr[2];
#pragma unroll
      for (int pb = 0; pb < 2; ++pb) {
        const int n = n0 + 16 * pb + r;
        kr[pb] = *reinterpret_cast<const bf16x8*>(sK + oK + 8192 * pb);
        bf16x8 vr = *reinterpret_cast<const bf16x8*>(sV + oK + 8192 * pb);
        if (n >= n_hi) {
#pragma unroll
          for (int e = 0; e < 8; ++e) { kr[pb][e] = (__bf16)0.f; vr[e] = (__bf16)0.f; }
        }
        bf16x4 lo4, hi4;
#pragma unroll
        for (int e = 0; e < 4; ++e) { lo4[e] = vr[e]; hi4[e] = vr[4 + e]; }
        *reinterpret_cast<bf16x4*>(myV + (16 * pb + r) * PV + 16 * g) = lo4;
        *reinterpret_cast<bf16x4*>(myV + (16 * pb + r) * PV + 16 * g + 8) = hi4;
      }
      bf16x8 vt[2];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) vt[tt] = tr_frag_small(myV, PV, 16 * tt, lane);
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        f32x4 s[2];
        float mt = -INFINITY;
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
          s[pb] = mfma32(kr[pb], qf[qt], f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (n0 + 16 * pb + 4 * g + e >= n_hi) s[pb][e] = -INFINITY;
            mt = fmaxf(mt, s[pb][e]);
          }
        }
        mt = wave16_max(mt);
        const float mnew = fmaxf(mrow[qt], mt);
        const float alpha = __builtin_amdgcn_exp2f(mrow[qt] - mnew);
        float ls = 0.f;
#pragma unroll
        for (int pb = 0; pb < 2; ++pb)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            s[pb][e] = __builtin_amdgcn_exp2f(s[pb][e] - mnew);
            ls += s[pb][e];
          }
        ls = wave16_sum(ls);
        lrow[qt] = lrow[qt] * alpha + ls;
        mrow[qt] = mnew;
        const bf16x8 pb8 = pack8(s[0], s[1]);
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
          for (int e = 0; e < 4; ++e) ot[tt][qt][e] *= alpha;
          ot[tt][qt] = mfma32(vt[tt], pb8, ot[tt][qt]);
        }
      }
    }
    lds_barrier();                       // B1: Kp / Vp tiles complete (X tile k consumed)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = tid + 512 * i, row = c >> 5, ch = c & 31;
      if (row < nlive) {
        const int64_t o = ((int64_t)b * a.N + n0 + row) * D + ch * 8;
        *reinterpret_cast<uint4*>(aa.KpO + o) = *reinterpret_cast<const uint4*>(sK + oC + 8192 * i);
        *reinterpret_cast<uint4*>(aa.VpO + o) = *reinterpret_cast<const uint4*>(sV + oC + 8192 * i);
      }
    }
  }
  const int64_t pb0 = ((int64_t)b * a.S + sp);
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int q = 16 * qt + r;
    if (q < a.m) {
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
        *reinterpret_cast<float4*>(a.Op + (pb0 * a.m + q) * D + 32 * j + 16 * tt + 4 * g) =
            float4{ot[tt][qt][0], ot[tt][qt][1], ot[tt][qt][2], ot[tt][qt][3]};
      if (g == 0) {
        a.Mp[(pb0 * H + j) * MQ + q] = mrow[qt];
        a.Lp[(pb0 * H + j) * MQ + q] = lrow[qt];
      }
    }
  }
}

// backward.  Orientation A (points on accumulator rows, as the forward): P^T, dS^T -> the set's
// dQp_h^T[f][q] += Kp_h^T[f][pt] dS^T[pt][q] (Kp^T through the LDS tile).  Orientation B
// (queries on accumulator rows: S = Qp_h Kp_h^T recomputed with one more MFMA - the per-lane
// Kp row registers serve as A operand of one and B operand of the other) -> dVp^T[f][pt] =
// dO_h^T[f][q] P[q][pt], dKp^T[f][pt] = Qp_h^T[f][q] dS[q][pt] with P / dS as B operands straight
// from the accumulators; both leave as 8-byte bf16 stores.
template <int D, int QT>
__global__ __launch_bounds__(64 * (D / 32)) void k_fq_attn_bwd(const FqArgs a) {
  constexpr int PV = 72, H = D / 32, MQ = 16 * QT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, j = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int b = blockIdx.x, sp = blockIdx.y;
  char* myK = smem + j * 32 * PV;
  const int per = (int)(((int64_t)(a.N + 31) / 32 + a.S - 1) / a.S) * 32;
  int len = a.N;
  if (a.lengths != nullptr) len = a.lengths[b] < a.N ? a.lengths[b] : a.N;
  const int n_lo = sp * per;
  const int n_hi = (n_lo + per < a.N) ? n_lo + per : a.N;          // rows written (zeros past len)
  // operands of head j.  qs = sl2e Qp (scores in the log2 domain), qn = Qp (for dKp), do = dO
  bf16x8 qs[QT], dof[QT];
  // A operands [row = f][k = q] of the products that sum over the queries:
  //   m > 16: 16x16x32, k-slot 8 g + i <-> query perm32(8 g + i) (pack8 order of the B operand)
  //   m <= 16: 16x16x16, k-slot 4 g + e <-> query 4 g + e
  bf16x8 qnT[2], doT[2];
  bf16x4 qn4[2], do4[2];
  float lse_c[QT], del_c[QT];             // per column q = 16 qt + r        (orientation A)
  float lse_r[QT][4], del_r[QT][4];       // per row q = 16 qt + 4 g + e      (orientation B)
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int q = 16 * qt + r;
    qs[qt] = q_frag(a.Qp, D, a.m, q, j, g, a.scale_log2e);
    dof[qt] = q_frag(a.dOa + (int64_t)b * a.m * D, D, a.m, q, j, g, 1.0f);
    lse_c[qt] = q < a.m ? a.LSE[((int64_t)b * H + j) * MQ + q] : 1.0e30f;
    del_c[qt] = q < a.m ? a.Delta[((int64_t)b * H + j) * MQ + q] : 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int qq = 16 * qt + 4 * g + e;
      lse_r[qt][e] = qq < a.m ? a.LSE[((int64_t)b * H + j) * MQ + qq] : 1.0e30f;
      del_r[qt][e] = qq < a.m ? a.Delta[((int64_t)b * H + j) * MQ + qq] : 0.f;
    }
  }
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    const int f = 32 * j + 16 * tt + r;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int kq = perm32(8 * g + i);
      const bool ok = QT == 2 && kq < a.m;
      qnT[tt][i] = (__bf16)(ok ? a.Qp[(int64_t)kq * D + f] : 0.f);
      doT[tt][i] = (__bf16)(ok ? a.dOa[((int64_t)b * a.m + kq) * D + f] : 0.f);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int kq = 4 * g + e;
      const bool ok = QT == 1 && kq < a.m;
      qn4[tt][e] = (__bf16)(ok ? a.Qp[(int64_t)kq * D + f] : 0.f);
      do4[tt][e] = (__bf16)(ok ? a.dOa[((int64_t)b * a.m + kq) * D + f] : 0.f);
    }
  }
  f32x4 dq[2][QT];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) dq[tt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int n0 = n_lo; n0 < n_hi; n0 += 32) {
    bf16x8 kr[2], vr[2];
    bool key_c[2];                        // orientation B: column = point 16 pb + r
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) {
      const int n = n0 + 16 * pb + r;
      key_c[pb] = n < len;
      {
        const int64_t o = ((int64_t)b * a.N + (n < n_hi ? n : n_hi - 1)) * D + 32 * j + 8 * g;
        kr[pb] = *reinterpret_cast<const bf16x8*>(a.Kp + o);
        vr[pb] = *reinterpret_cast<const bf16x8*>(a.Vp + o);
        if (n >= n_hi) {
#pragma unroll
          for (int e = 0; e < 8; ++e) { kr[pb][e] = (__bf16)0.f; vr[pb][e] = (__bf16)0.f; }
        }
      }
      {
        bf16x4 lo4, hi4;
#pragma unroll
        for (int e = 0; e < 4; ++e) { lo4[e] = kr[pb][e]; hi4[e] = kr[pb][4 + e]; }
        *reinterpret_cast<bf16x4*>(myK + (16 * pb + r) * PV + 16 * g) = lo4;
        *reinterpret_cast<bf16x4*>(myK + (16 * pb + r) * PV + 16 * g + 8) = hi4;
      }
    }
    bf16x8 kt[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) kt[tt] = tr_frag_small(myK, PV, 16 * tt, lane);
    // ---- orientation A: dQp ----
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      f32x4 ds[2];
#pragma unroll
      for (int pb = 0; pb < 2; ++pb) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const f32x4 sv = mfma32(kr[pb], qs[qt], z);
        const f32x4 da = mfma32(vr[pb], dof[qt], z);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool key = n0 + 16 * pb + 4 * g + e < len;
          const float p = key ? __builtin_amdgcn_exp2f(sv[e] - lse_c[qt]) : 0.f;
          ds[pb][e] = p * (da[e] - del_c[qt]) * a.scale;
        }
      }
      const bf16x8 dsb = pack8(ds[0], ds[1]);
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) dq[tt][qt] = mfma32(kt[tt], dsb, dq[tt][qt]);
    }
    // ---- orientation B: dKp, dVp ----
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) {
      f32x4 pq[QT], dsq[QT];
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const f32x4 sv = mfma32(qs[qt], kr[pb], z);          // rows q, column pt
        const f32x4 da = mfma32(dof[qt], vr[pb], z);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float p = key_c[pb] ? __builtin_amdgcn_exp2f(sv[e] - lse_r[qt][e]) : 0.f;
          pq[qt][e] = p;
          dsq[qt][e] = p * (da[e] - del_r[qt][e]) * a.scale;
        }
      }
      const int n = n0 + 16 * pb + r;
      if (n < n_hi) {
        const int64_t o = ((int64_t)b * a.N + n) * D + 32 * j;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          f32x4 dv = {0.f, 0.f, 0.f, 0.f}, dk = {0.f, 0.f, 0.f, 0.f};
          if (QT == 2) {
            dv = mfma32(doT[tt], pack8(pq[0], pq[QT - 1]), dv);
            dk = mfma32(qnT[tt], pack8(dsq[0], dsq[QT - 1]), dk);
          } else {
            dv = mfma16(do4[tt], pack4(pq[0]), dv);
            dk = mfma16(qn4[tt], pack4(dsq[0]), dk);
          }
          *reinterpret_cast<bf16x4*>(a.dVp + o + 16 * tt + 4 * g) = pack4(dv);
          *reinterpret_cast<bf16x4*>(a.dKp + o + 16 * tt + 4 * g) = pack4(dk);
        }
      }
    }
  }
  const int64_t pb0 = ((int64_t)b * a.S + sp);
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int q = 16 * qt + r;
    if (q < a.m) {
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
        *reinterpret_cast<float4*>(a.dQpPart + (pb0 * a.m + q) * D + 32 * j + 16 * tt + 4 * g) =
            float4{dq[tt][qt][0], dq[tt][qt][1], dq[tt][qt][2], dq[tt][qt][3]};
    }
  }
}

// k_fq_attn_bwd2 (m = 32, d = 256): the same arithmetic with full-line global traffic, as
// k_attn1_bwd2 is to k_attn1_bwd.  The workgroup streams whole [32 keys][256] tiles of Kp and Vp in
// by LDS-DMA (double buffered), each wave (= head) reads its 64-byte slices from LDS, writes its
// slices of the dKp / dVp tiles to LDS, and the two tiles leave in 16-byte pieces of full rows -
// the per-wave form reads 16 rows x 64 bytes per load instruction and writes 8-byte pieces.
__global__ __launch_bounds__(512, 2) void k_fq_attn_bwd2(const FqArgs a) {
  constexpr int D = 256, QT = 2, PV = 72, H = D / 32, MQ = 16 * QT;
  constexpr int ROWB = D * 2, TILEB = 32 * ROWB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sKb = smem;                        // [2][TILEB]
  char* sVb = smem + 2 * TILEB;            // [2][TILEB]
  char* sDK = smem + 4 * TILEB;            // dKp tile
  char* sDV = smem + 5 * TILEB;            // dVp tile
  const int tid = threadIdx.x, lane = tid & 63;
  const int j = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int b = blockIdx.x, sp = blockIdx.y;
  char* myK = smem + 6 * TILEB + j * 32 * PV;
  const int per = (int)(((int64_t)(a.N + 31) / 32 + a.S - 1) / a.S) * 32;
  int len = a.N;
  if (a.lengths != nullptr) len = a.lengths[b] < a.N ? a.lengths[b] : a.N;
  const int n_lo = sp * per;
  const int n_hi = (n_lo + per < a.N) ? n_lo + per : a.N;
  const int T = n_lo < n_hi ? (n_hi - n_lo + 31) / 32 : 0;
  bf16x8 qs[QT], dof[QT], qnT[2], doT[2];
  float lse_c[QT], del_c[QT], lse_r[QT][4], del_r[QT][4];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int q = 16 * qt + r;
    qs[qt] = q_frag(a.Qp, D, a.m, q, j, g, a.scale_log2e);
    dof[qt] = q_frag(a.dOa + (int64_t)b * a.m * D, D, a.m, q, j, g, 1.0f);
    lse_c[qt] = q < a.m ? a.LSE[((int64_t)b * H + j) * MQ + q] : 1.0e30f;
    del_c[qt] = q < a.m ? a.Delta[((int64_t)b * H + j) * MQ + q] : 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int qq = 16 * qt + 4 * g + e;
      lse_r[qt][e] = qq < a.m ? a.LSE[((int64_t)b * H + j) * MQ + qq] : 1.0e30f;
      del_r[qt][e] = qq < a.m ? a.Delta[((int64_t)b * H + j) * MQ + qq] : 0.f;
    }
  }
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    const int f = 32 * j + 16 * tt + r;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int kq = perm32(8 * g + i);
      const bool ok = kq < a.m;
      qnT[tt][i] = (__bf16)(ok ? a.Qp[(int64_t)kq * D + f] : 0.f);
      doT[tt][i] = (__bf16)(ok ? a.dOa[((int64_t)b * a.m + kq) * D + f] : 0.f);
    }
  }
  f32x4 dq[2][QT];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) dq[tt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // this lane's pieces of a tile: the head's 16 bytes of row r (natural chunk 4 j + g), its
  // accumulator-layout 8 bytes (features 32 j + 16 t + 4 g), the coalesced 16-byte piece
  int oK, oD[2];
  oK = swz(r, 4 * j + g, ROWB);
#pragma unroll
  for (int t = 0; t < 2; ++t) oD[t] = swz(r, 4 * j + 2 * t + (g >> 1), ROWB) + 8 * (g & 1);
  const int oC = swz(tid >> 5, tid & 31, ROWB);
  auto lds_barrier = [] {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  auto dma = [&](int k) {                 // tile k of Kp and Vp (issued from inline asm: k_isab1_fwd256)
    const int n0 = n_lo + 32 * k, par = k & 1;
#pragma unroll
    for (int w = 0; w < 2; ++w) {
      const __bf16* base = w == 0 ? a.Kp : a.Vp;
      char* dst = (w == 0 ? sKb : sVb) + par * TILEB;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int p = (2 * j + i) * 64 + lane;
        const int row = p >> 5, slot = p & 31;
        const int ch = (slot & ~15) | ((slot ^ row) & 15);
        const int n = n0 + row < a.N ? n0 + row : a.N - 1;
        const __bf16* src = base + ((int64_t)b * a.N + n) * D + ch * 8;
        const unsigned ldst = __builtin_amdgcn_readfirstlane(
            (unsigned)(uintptr_t)(lds_void_t*)(dst + (2 * j + i) * 1024));
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(ldst) : "memory");
      }
    }
  };
  if (T > 0) dma(0);
  for (int k = 0; k < T; ++k) {
    const int par = k & 1, n0 = n_lo + 32 * k, nlive = n_hi - n0;
    const char* sK = sKb + par * TILEB;
    const char* sV = sVb + par * TILEB;
    // tile k + 1 starts to stream in; this tile's DMA (one iteration old) must have landed: younger
    // are the 4 pieces just issued and the 4 stores of tile k - 1 (always a full tile)
    if (k + 1 < T) {
      dma(k + 1);
      if (k == 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    lds_barrier();                       // B0: Kp / Vp tiles complete; previous output tiles stored
    bf16x8 kr[2], vr[2];
    bool key_c[2];
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) {
      const int n = n0 + 16 * pb + r;
      key_c[pb] = n < len;
      kr[pb] = *reinterpret_cast<const bf16x8*>(sK + oK + 8192 * pb);
      vr[pb] = *reinterpret_cast<const bf16x8*>(sV + oK + 8192 * pb);
      if (n >= n_hi) {                     // rows past the range: the DMA fetched a duplicate
#pragma unroll
        for (int e = 0; e < 8; ++e) { kr[pb][e] = (__bf16)0.f; vr[pb][e] = (__bf16)0.f; }
      }
      bf16x4 lo4, hi4;
#pragma unroll
      for (int e = 0; e < 4; ++e) { lo4[e] = kr[pb][e]; hi4[e] = kr[pb][4 + e]; }
      *reinterpret_cast<bf16x4*>(myK + (16 * pb + r) * PV + 16 * g) = lo4;
      *reinterpret_cast<bf16x4*>(myK + (16 * pb + r) * PV + 16 * g + 8) = hi4;
    }
    bf16x8 kt[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) kt[tt] = tr_frag_small(myK, PV, 16 * tt, lane);
    // ---- orientation A: dQp ----
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      f32x4 ds[2];
#pragma unroll
      for (int pb = 0; pb < 2; ++pb) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const f32x4 sv = mfma32(kr[pb], qs[qt], z);
        const f32x4 da = mfma32(vr[pb], dof[qt], z);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool key = n0 + 16 * pb + 4 * g + e < len;
          const float p = key ? __builtin_amdgcn_exp2f(sv[e] - lse_c[qt]) : 0.f;
          ds[pb][e] = p * (da[e] - del_c[qt]) * a.scale;
        }
      }
      const bf16x8 dsb = pack8(ds[0], ds[1]);
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) dq[tt][qt] = mfma32(kt[tt], dsb, dq[tt][qt]);
    }
    // ---- orientation B: dKp, dVp -> own slices of the output tiles ----
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) {
      f32x4 pq[QT], dsq[QT];
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const f32x4 sv = mfma32(qs[qt], kr[pb], z);          // rows q, column pt
        const f32x4 da = mfma32(dof[qt], vr[pb], z);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float p = key_c[pb] ? __builtin_amdgcn_exp2f(sv[e] - lse_r[qt][e]) : 0.f;
          pq[qt][e] = p;
          dsq[qt][e] = p * (da[e] - del_r[qt][e]) * a.scale;
        }
      }
      const bf16x8 pb8 = pack8(pq[0], pq[1]), ds8 = pack8(dsq[0], dsq[1]);
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const f32x4 dv = mfma32(doT[tt], pb8, z);
        const f32x4 dk = mfma32(qnT[tt], ds8, z);
        *reinterpret_cast<bf16x4*>(sDV + oD[tt] + 8192 * pb) = pack4(dv);
        *reinterpret_cast<bf16x4*>(sDK + oD[tt] + 8192 * pb) = pack4(dk);
      }
    }
    lds_barrier();                       // B1: dKp / dVp tiles complete
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = tid + 512 * i, row = c >> 5, ch = c & 31;
      if (row < nlive) {
        const int64_t o = ((int64_t)b * a.N + n0 + row) * D + ch * 8;
        *reinterpret_cast<uint4*>(a.dKp + o) = *reinterpret_cast<const uint4*>(sDK + oC + 8192 * i);
        *reinterpret_cast<uint4*>(a.dVp + o) = *reinterpret_cast<const uint4*>(sDV + oC + 8192 * i);
      }
    }
  }
  const int64_t pb0 = ((int64_t)b * a.S + sp);
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int q = 16 * qt + r;
    if (q < a.m) {
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
        *reinterpret_cast<float4*>(a.dQpPart + (pb0 * a.m + q) * D + 32 * j + 16 * tt + 4 * g) =
            float4{dq[tt][qt][0], dq[tt][qt][1], dq[tt][qt][2], dq[tt][qt][3]};
    }
  }
}

// merge of the forward partials + residual: O[b][q][f] = Qp[q][f] + sum_s w_s Op_s / sum_s w_s L_s,
// LSE[b][h][q] = M + log2 L; Oa (= A Vp, for Delta) is O - Qp
__global__ __launch_bounds__(256) void k_fq_merge(const float* __restrict__ Op,
                                                  const float* __restrict__ Mp,
                                                  const float* __restrict__ Lp,
                                                  const float* __restrict__ Qp, int B, int S,
                                                  int m, int D, int MQ, float* __restrict__ O,
                                                  float* __restrict__ LSE) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)B * m * D) return;
  const int f = (int)(i % D);
  const int q = (int)((i / D) % m);
  const int64_t b = i / ((int64_t)D * m);
  const int H = D / 32, j = f / 32;
  float M = -INFINITY;
  for (int s = 0; s < S; ++s) M = fmaxf(M, Mp[((b * S + s) * H + j) * MQ + q]);
  float L = 0.f, t = 0.f;
  for (int s = 0; s < S; ++s) {
    const float ms = Mp[((b * S + s) * H + j) * MQ + q];
    if (ms == -INFINITY) continue;
    const float w = exp2f(ms - M);
    L += w * Lp[((b * S + s) * H + j) * MQ + q];
    t += w * Op[((b * S + s) * m + q) * D + f];
  }
  O[i] = Qp[(int64_t)q * D + f] + t / L;
  if ((f & 31) == 0) LSE[(b * H + j) * MQ + q] = M + log2f(L);
}

// Delta[b][h][q] = sum_{f in head} dO[b][q][f] (O[b][q][f] - Qp[q][f])
__global__ __launch_bounds__(256) void k_fq_delta(const float* __restrict__ dO,
                                                  const float* __restrict__ O,
                                                  const float* __restrict__ Qp, int B, int m,
                                                  int D, int MQ, float* __restrict__ Delta) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int H = D / 32;
  if (i >= (int64_t)B * H * m) return;
  const int q = (int)(i % m);
  const int j = (int)((i / m) % H);
  const int64_t b = i / ((int64_t)m * H);
  float t = 0.f;
  for (int f = 0; f < 32; ++f) {
    const int64_t o = (b * m + q) * D + 32 * j + f;
    t += dO[o] * (O[o] - Qp[(int64_t)q * D + 32 * j + f]);
  }
  Delta[(b * H + j) * MQ + q] = t;
}

// dOt[b][q][f] = dO[b][q][f] + sum_s dQpPart[b][s][q][f]   (gradient w.r.t. Qp of set b)
__global__ __launch_bounds__(256) void k_fq_dq_sum(const float* __restrict__ dO,
                                                   const float* __restrict__ part, int B, int S,
                                                   int md, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)B * md) return;
  const int64_t b = i / md, o = i - b * md;
  float t = dO[i];
  for (int s = 0; s < S; ++s) t += part[(b * S + s) * md + o];
  out[i] = t;
}

__global__ __launch_bounds__(256) void k_cvt_f32_bf16(const float* __restrict__ s,
                                                      __bf16* __restrict__ d, int64_t n4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const float4 v = reinterpret_cast<const float4*>(s)[i];
  reinterpret_cast<bf16x4*>(d)[i] = bf16x4{(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
}
__global__ __launch_bounds__(256) void k_cvt_bf16_f32(const __bf16* __restrict__ s,
                                                      float* __restrict__ d, int64_t n4,
                                                      int accumulate) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const bf16x4 v = reinterpret_cast<const bf16x4*>(s)[i];
  float4 o = float4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
  if (accumulate) {
    const float4 p = reinterpret_cast<const float4*>(d)[i];
    o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w;
  }
  reinterpret_cast<float4*>(d)[i] = o;
}

// Kp / Vp of a layer whose input has dk <= 4 columns (layer 1): exact fp32 FMAs, bf16 out
__global__ __launch_bounds__(256) void k_kv_proj_small(const float* __restrict__ X, int64_t M,
                                                       int dk, const float* __restrict__ Wk,
                                                       const float* __restrict__ bk,
                                                       const float* __restrict__ Wv,
                                                       const float* __restrict__ bv, int D,
                                                       __bf16* __restrict__ Kp,
                                                       __bf16* __restrict__ Vp) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;      // one thread: 4 features of a row
  const int per = D / 4;
  if (i >= M * per) return;
  const int64_t row = i / per;
  const int f0 = (int)(i - row * per) * 4;
  float x[4] = {0.f, 0.f, 0.f, 0.f};
  for (int c = 0; c < dk; ++c) x[c] = X[row * dk + c];
  bf16x4 k4, v4;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float ak = bk[f0 + e], av = bv[f0 + e];
    for (int c = 0; c < dk; ++c) {
      ak = fmaf(x[c], Wk[(f0 + e) * dk + c], ak);
      av = fmaf(x[c], Wv[(f0 + e) * dk + c], av);
    }
    k4[e] = (__bf16)ak;
    v4[e] = (__bf16)av;
  }
  *reinterpret_cast<bf16x4*>(Kp + row * D + f0) = k4;
  *reinterpret_cast<bf16x4*>(Vp + row * D + f0) = v4;
}


// dW[256 x dq] += G[M x 256]^T X[M x dq] (dq <= 4, fp32 X), db += colsum(G): layer-1 fc_q.
// HBM-bound read of G: a thread owns 8 features (one 16-byte load per row) of every 8th row of
// the workgroup's range, four rows in flight; the 8 row lanes meet in LDS, then one atomic per
// element per workgroup.
__global__ __launch_bounds__(256) void k_wgrad_small256(const __bf16* __restrict__ G,
                                                        const float* __restrict__ X, int64_t M,
                                                        int dq, int rows_per_wg,
                                                        float* __restrict__ slabs) {
  constexpr int D = 256;
  __shared__ __attribute__((aligned(16))) float red[8][D][5];
  const int fc = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_wg;
  const int64_t r1 = r0 + rows_per_wg < M ? r0 + rows_per_wg : M;
  float acc[8][4], bs[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    bs[k] = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[k][c] = 0.f;
  }
  // the range's points, [row][4] (zero padded), staged once: fetched per row by every thread
  // they were 12 four-byte loads per 4 rows next to the 4 sixteen-byte loads of G - the kernel
  // was bound by load instructions, not bytes (2.2 TB/s)
  float* sXs = &red[0][0][0];
  for (int i = threadIdx.x; i < rows_per_wg * 4; i += 256) {
    const int64_t rr = r0 + (i >> 2);
    const int c = i & 3;
    const float v = X[(rr < r1 ? rr : r1 - 1) * dq + (c < dq ? c : 0)];
    sXs[i] = (rr < r1 && c < dq) ? v : 0.f;
  }
  __syncthreads();
  // (rows past the end are fetched from the last row and weighted by zero: a load under a
  //  divergent `if` gets its own basic block and its own s_waitcnt vmcnt(0) - four serialised
  //  round trips per iteration instead of four loads in flight)
  for (int64_t row = r0 + rl; row < r1; row += 32) {
    bf16x8 gv[4];
    float xv[4][4], wv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t rr = row + 8 * u;
      const int64_t rc = rr < r1 ? rr : r1 - 1;
      wv[u] = rr < r1 ? 1.f : 0.f;
      gv[u] = *reinterpret_cast<const bf16x8*>(G + rc * D + 8 * fc);
      const float4 x4 = *reinterpret_cast<const float4*>(sXs + (rc - r0) * 4);
      xv[u][0] = x4.x * wv[u]; xv[u][1] = x4.y * wv[u];
      xv[u][2] = x4.z * wv[u]; xv[u][3] = x4.w * wv[u];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float g = (float)gv[u][k];
        bs[k] = fmaf(g, wv[u], bs[k]);
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[k][c] = fmaf(g, xv[u][c], acc[k][c]);
      }
  }
  __syncthreads();                       // the staged points share their LDS with `red`
#pragma unroll
  for (int k = 0; k < 8; ++k) {
#pragma unroll
    for (int c = 0; c < 4; ++c) red[rl][8 * fc + k][c] = acc[k][c];
    red[rl][8 * fc + k][4] = bs[k];
  }
  __syncthreads();
  const int f = threadIdx.x;
  float t[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < 8; ++q)
#pragma unroll
    for (int c = 0; c < 5; ++c) t[c] += red[q][f][c];
  // partial sums leave as a slab [workgroup][256][5]; k_wgrad_small256_sum adds the slabs in a
  // fixed order (1026 workgroups x 1024 fp32 atomics onto the same 1024 addresses measured
  // 30 us of this kernel's 117, and made the result depend on the arrival order)
  float* slab = slabs + (int64_t)blockIdx.x * D * 5;
#pragma unroll
  for (int c = 0; c < 5; ++c) slab[f * 5 + c] = t[c];
}

// out[o] += sum over slabs, o = 5 f + c: c < dq -> dW[f][c], c == 4 -> db[f].  One workgroup per 64
// outputs, 16 slab groups of 64 lanes, eight loads in flight each.
__global__ __launch_bounds__(1024) void k_wgrad_small256_sum(const float* __restrict__ slabs,
                                                            int nslabs, int dq,
                                                            float* __restrict__ dW,
                                                            float* __restrict__ db) {
  constexpr int NO = 256 * 5;
  __shared__ float red[16][64];
  const int sg = threadIdx.x >> 6, c = threadIdx.x & 63, o = blockIdx.x * 64 + c;
  const float* s = slabs + o;
  float t = 0.f;
  int w = sg;
  for (; w + 112 < nslabs; w += 128) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = s[(int64_t)(w + 16 * u) * NO];
#pragma unroll
    for (int u = 0; u < 8; ++u) t += v[u];
  }
  for (; w < nslabs; w += 16) t += s[(int64_t)w * NO];
  red[sg][c] = t;
  __syncthreads();
  if (sg == 0) {
    float v = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) v += red[q][c];
    const int f = o / 5, k = o - 5 * f;
    if (k < dq) dW[f * dq + k] += v;
    else if (k == 4 && db != nullptr) db[f] += v;
  }
}

// ---- per-set epilogue pieces of the few-queries block whose keys have dk <= 4 columns --------
// O[b][q][f] = Qp[q][f] + bv[f] + sum_c T[b][j m + q][c] Wv[f][c]      (j = head of f)
__global__ __launch_bounds__(256) void k_epi_small_fwd(const float* __restrict__ T,
                                                       const float* __restrict__ Qp,
                                                       const float* __restrict__ Wv,
                                                       const float* __restrict__ bv, int B, int m,
                                                       int D, int dk, float* __restrict__ O) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)B * m * D) return;
  const int f = (int)(i % D), q = (int)((i / D) % m);
  const int64_t b = i / ((int64_t)D * m);
  const int j = f / 32, R = (D / 32) * m;
  float acc = Qp[(int64_t)q * D + f] + bv[f];
  for (int c = 0; c < dk; ++c) acc = fmaf(T[(b * R + j * m + q) * dk + c], Wv[f * dk + c], acc);
  O[i] = acc;
}
// the same for dk = 256 (PMA): one wave per output, lanes over the contraction - a thread per
// output walks Wv[f][:] with a 1 KiB stride between neighbouring lanes (46 us for 32 K outputs)
__global__ __launch_bounds__(256) void k_epi_wide_fwd(const float* __restrict__ T,
                                                      const float* __restrict__ Qp,
                                                      const float* __restrict__ Wv,
                                                      const float* __restrict__ bv, int B, int m,
                                                      float* __restrict__ O) {
  constexpr int D = 256, DKW = 256, PER = 8;             // outputs per wave
  const int lane = threadIdx.x & 63;
  const int64_t w = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * PER;
  const int R = (D / 32) * m;
  float4 tv[PER], wv[PER];
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int64_t i = w + u < (int64_t)B * m * D ? w + u : (int64_t)B * m * D - 1;
    const int f = (int)(i % D), q = (int)((i / D) % m);
    const int64_t b = i / ((int64_t)D * m);
    tv[u] = *reinterpret_cast<const float4*>(T + (b * R + (f / 32) * m + q) * DKW + 4 * lane);
    wv[u] = *reinterpret_cast<const float4*>(Wv + (int64_t)f * DKW + 4 * lane);
  }
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    float acc = tv[u].x * wv[u].x + tv[u].y * wv[u].y + tv[u].z * wv[u].z + tv[u].w * wv[u].w;
#pragma unroll
    for (int sh = 1; sh < 64; sh <<= 1) acc += __shfl_xor(acc, sh);
    const int64_t i = w + u;
    if (lane == 0 && i < (int64_t)B * m * D) {
      const int f = (int)(i % D), q = (int)((i / D) % m);
      O[i] = acc + Qp[(int64_t)q * D + f] + bv[f];
    }
  }
}
// dT[b][r][c] = sum_{f in head j} dO[b][q][f] Wv[f][c] ; Delta[b][r] = sum_c dT T   (r = j m + q)
__global__ __launch_bounds__(256) void k_epi_small_bwd(const float* __restrict__ dO,
                                                       const float* __restrict__ T,
                                                       const float* __restrict__ Wv, int B, int m,
                                                       int D, int dk, float* __restrict__ dT,
                                                       float* __restrict__ Delta) {
  const int R = (D / 32) * m;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)B * R) return;
  const int rr = (int)(i % R), j = rr / m, q = rr - j * m;
  const int64_t b = i / R;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int f = 32 * j; f < 32 * j + 32; ++f) {
    const float g = dO[(b * m + q) * D + f];
    for (int c = 0; c < dk; ++c) acc[c] = fmaf(g, Wv[f * dk + c], acc[c]);
  }
  float del = 0.f;
  for (int c = 0; c < dk; ++c) {
    dT[i * dk + c] = acc[c];
    del = fmaf(acc[c], T[i * dk + c], del);
  }
  Delta[i] = del;
}
// dWv[f][c] += sum_{b,q} dO[b][q][f] T[b][j(f) m + q][c] ; dbv[f] += sum_{b,q} dO[b][q][f]
__global__ __launch_bounds__(256) void k_epi_small_wv(const float* __restrict__ dO,
                                                      const float* __restrict__ T, int B, int m,
                                                      int dk, int rows_per_wg,
                                                      float* __restrict__ slabs) {
  constexpr int D = 256;
  const int f = threadIdx.x, j = f / 32, R = (D / 32) * m;
  const int64_t M = (int64_t)B * m;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_wg;
  const int64_t r1 = r0 + rows_per_wg < M ? r0 + rows_per_wg : M;
  float acc[4] = {0.f, 0.f, 0.f, 0.f}, bs = 0.f;
  for (int64_t row0 = r0; row0 < r1; row0 += 8) {           // 8 rows in flight
    float gv[8], tv[8][4];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t row = row0 + u < r1 ? row0 + u : r1 - 1;
      const int64_t b = row / m;
      const int q = (int)(row - b * m);
      gv[u] = row0 + u < r1 ? dO[row * D + f] : 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) tv[u][c] = c < dk ? T[(b * R + j * m + q) * dk + c] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      bs += gv[u];
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] = fmaf(gv[u], tv[u][c], acc[c]);
    }
  }
  // (no atomics: a slab [workgroup][256][5] in the format of k_wgrad_small256, summed in order)
  float* slab = slabs + (int64_t)blockIdx.x * D * 5;
#pragma unroll
  for (int c = 0; c < 4; ++c) slab[f * 5 + c] = acc[c];
  slab[f * 5 + 4] = bs;
}


// =====================================================================================
// PMA at dk = 256 (R = h*m <= 16 score rows): reassociated form of mab0_bf16.hip -
//   G' = sl2e Qp_h Wk_h (batch invariant), S = G' X^T, A = softmax_N(S), T = A X -
// so the N keys are never projected and X is read once.  One 16-row score tile; a wave streams
// its own 32-point tiles of X through a private LDS tile (row fragments + transposed fragments).
// =====================================================================================
struct PmaArgs {
  const __bf16* X;             // [B*N][256]
  const __bf16* Gb;            // [>= 16][256] rows r (sl2e folded in; rows >= R zero)
  float *Tp, *Mp, *Lp;         // forward partials [B][S][16][256], [B][S][16]
  const __bf16* dTb;           // [B][16][256]
  const __bf16* TG;            // [B][256][32]: k-slot 8 g + j = j < 4 ? dT[4g+j][c] : G'[4g+j-4][c]
  const float *LSEp, *Delta;   // [B][16]
  __bf16* dX;                  // [B*N][256] or null
  float* DG;                   // backward: slabs [B][S][16][256] of per-workgroup sums
  int B, N, R, S, accumulate_dx;
  const int32_t* lengths;
};

__global__ __launch_bounds__(256, 1) void k_pma_fwd256(const PmaArgs a) {
  constexpr int DK = 256, FT = DK / 16, KS = DK / 32, TB = 32 * DK * 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sX = smem;                                             // 4 waves x 16 KiB, later slabs
  float* sAl = reinterpret_cast<float*>(smem + 4 * TB);        // 4 x 16
  float* sM = sAl + 64;
  float* sL = sM + 64;
  float* sT = reinterpret_cast<float*>(sX);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int b = blockIdx.x, sp = blockIdx.y;
  const int per = (int)(((int64_t)(a.N + 127) / 128 + a.S - 1) / a.S) * 128;
  int len = a.N;
  if (a.lengths != nullptr) len = a.lengths[b] < a.N ? a.lengths[b] : a.N;
  const int n_lo = sp * per, n_hi = (n_lo + per < len) ? n_lo + per : len;
  bf16x8 gB[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
    gB[ks] = *reinterpret_cast<const bf16x8*>(a.Gb + (int64_t)r * DK + 32 * ks + 8 * g);
  char* myX = sX + wave * TB;
  float* myAl = sAl + wave * 16;
  float mrow = -INFINITY, lrow = 0.f;
  f32x4 T[FT];
#pragma unroll
  for (int ft = 0; ft < FT; ++ft) T[ft] = f32x4{0.f, 0.f, 0.f, 0.f};
  // The wave's next tile is fetched into registers (64 VGPRs - the kernel runs one wave per SIMD,
  // there are 512) while the current one is worked on: without it every tile starts with an
  // exposed round trip to memory.
  bf16x8 nx[16];
  auto fetch_tile = [&](int n0) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int c = lane + 64 * e, row = c >> 5, ch = c & 31;
      const int nr = n0 + row;
      // (rows past the end of the range read its last row and are zeroed when the tile goes to LDS,
      //  one iteration later: zeroed here, hipcc waits for every load right behind its issue -
      //  vmcnt(15) ... vmcnt(0) - and the tile that was meant to arrive during the current tile's
      //  arithmetic is waited for before that arithmetic starts)
      nx[e] = *reinterpret_cast<const bf16x8*>(
          a.X + ((int64_t)b * a.N + (nr < n_hi ? nr : n_hi - 1)) * DK + ch * 8);
    }
  };
  if (n_lo + wave * 32 < n_hi) fetch_tile(n_lo + wave * 32);
  for (int n0 = n_lo + wave * 32; n0 < n_hi; n0 += 128) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int c = lane + 64 * e, row = c >> 5, ch = c & 31;
      bf16x8 v = nx[e];
      if (n0 + row >= n_hi) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (__bf16)0.f;
      }
      *reinterpret_cast<bf16x8*>(myX + tr_off256(row, ch)) = v;
    }
    if (n0 + 128 < n_hi) fetch_tile(n0 + 128);
    f32x4 s[2];
    float mt = -INFINITY;
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) {
      s[pb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        s[pb] = mfma32(*reinterpret_cast<const bf16x8*>(myX + tr_off256(16 * pb + r, 4 * ks + g)),
                       gB[ks], s[pb]);
      // rows of s = points 16 pb + 4 g + e ; column = score row r
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (n0 + 16 * pb + 4 * g + e >= n_hi) s[pb][e] = -INFINITY;
        mt = fmaxf(mt, s[pb][e]);
      }
    }
    mt = wave16_max(mt);
    const float mnew = fmaxf(mrow, mt);            // finite: the tile has >= 1 live point
    const float alpha = __builtin_amdgcn_exp2f(mrow - mnew);
    float ls = 0.f;
#pragma unroll
    for (int pb = 0; pb < 2; ++pb)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s[pb][e] = __builtin_amdgcn_exp2f(s[pb][e] - mnew);
        ls += s[pb][e];
      }
    ls = wave16_sum(ls);
    lrow = lrow * alpha + ls;
    mrow = mnew;
    // the T tiles hold score rows 4g+e on their accumulator rows: fetch their alphas
    if (g == 0) myAl[r] = alpha;
    const float4 a4 = *reinterpret_cast<const float4*>(&myAl[4 * g]);
    const bf16x8 pa = pack8(s[0], s[1]);
#pragma unroll
    for (int ft = 0; ft < FT; ++ft) {
      T[ft][0] *= a4.x; T[ft][1] *= a4.y; T[ft][2] *= a4.z; T[ft][3] *= a4.w;
      T[ft] = mfma32(pa, tr_frag256(myX, ft, lane), T[ft]);
    }
  }
  // ---- merge the four waves' partial (m, l, T): per-wave slabs over the dead X tiles ----
  __syncthreads();
  if (g == 0) {
    sM[wave * 16 + r] = mrow;
    sL[wave * 16 + r] = lrow;
  }
  __syncthreads();
  float* mySlab = sT + wave * 16 * DK;
  {
    float f4[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int rr = 4 * g + e;
      const float M = fmaxf(fmaxf(sM[rr], sM[16 + rr]), fmaxf(sM[32 + rr], sM[48 + rr]));
      const float mine = sM[wave * 16 + rr];
      f4[e] = (mine == -INFINITY) ? 0.f : exp2f(mine - M);
    }
#pragma unroll
    for (int ft = 0; ft < FT; ++ft)
#pragma unroll
      for (int e = 0; e < 4; ++e) mySlab[(4 * g + e) * DK + 16 * ft + r] = T[ft][e] * f4[e];
  }
  __syncthreads();
  const int64_t pbase = ((int64_t)b * a.S + sp) * 16;
  for (int i = tid; i < 16 * DK; i += 256) {
    const int rr = i / DK;
    a.Tp[pbase * DK + i] = sT[i] + sT[16 * DK + i] + sT[2 * 16 * DK + i] + sT[3 * 16 * DK + i];
    if (i == rr * DK) {
      float M = -INFINITY;
#pragma unroll
      for (int w = 0; w < 4; ++w) M = fmaxf(M, sM[w * 16 + rr]);
      float L = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const float mw = sM[w * 16 + rr];
        if (mw != -INFINITY) L += sL[w * 16 + rr] * exp2f(mw - M);
      }
      a.Mp[pbase + rr] = M;
      a.Lp[pbase + rr] = L;
    }
  }
}

// T[b][r][c] = sum_s w_s Tp / sum_s w_s Lp ; LSE[b][r] = M + log2 L   (r < R)
__global__ __launch_bounds__(256) void k_pma_merge(const float* __restrict__ Tp,
                                                   const float* __restrict__ Mp,
                                                   const float* __restrict__ Lp, int B, int S,
                                                   int R, float* __restrict__ T,
                                                   float* __restrict__ LSE) {
  constexpr int DK = 256;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)B * R * DK) return;
  const int c = (int)(i % DK), rr = (int)((i / DK) % R);
  const int64_t b = i / ((int64_t)DK * R);
  float M = -INFINITY;
  for (int s = 0; s < S; ++s) M = fmaxf(M, Mp[(b * S + s) * 16 + rr]);
  float L = 0.f, t = 0.f;
  for (int s = 0; s < S; ++s) {
    const float ms = Mp[(b * S + s) * 16 + rr];
    if (ms == -INFINITY) continue;
    const float w = exp2f(ms - M);
    L += w * Lp[(b * S + s) * 16 + rr];
    t += w * Tp[((b * S + s) * 16 + rr) * DK + c];
  }
  T[i] = t / L;
  if (c == 0) LSE[b * R + rr] = M + log2f(L);
}

// per set: dT[r][c] = sum_{f in head j} dO[q][f] Wv[f][c], Delta[r] = <dT[r], T[r]>, and the
// operand images of k_pma_bwd256 (dTb natural rows, TG = [dT | G'] interleaved per column)
__global__ __launch_bounds__(256) void k_pma_epi_bwd(const float* __restrict__ dO,
                                                     const float* __restrict__ T,
                                                     const float* __restrict__ LSE,
                                                     const float* __restrict__ Wv,
                                                     const float* __restrict__ Gf, int m, int R,
                                                     __bf16* __restrict__ dTb,
                                                     __bf16* __restrict__ TG,
                                                     float* __restrict__ Delta,
                                                     float* __restrict__ LSEp) {
  constexpr int D = 256;
  __shared__ float sdO[2 * D];            // m <= 2 query rows
  __shared__ float sDel[16][4];
  const int b = blockIdx.x, c = threadIdx.x;
  for (int i = c; i < m * D; i += 256) sdO[i] = dO[(int64_t)b * m * D + i];
  __syncthreads();
  float dT[16];
#pragma unroll
  for (int rr = 0; rr < 16; ++rr) dT[rr] = 0.f;
#pragma unroll
  for (int rr = 0; rr < 16; ++rr) {
    if (rr < R) {
      const int j = rr / m, q = rr - j * m;
      float acc = 0.f;
      for (int f = 0; f < 32; ++f) acc = fmaf(sdO[q * D + 32 * j + f], Wv[(32 * j + f) * D + c], acc);
      dT[rr] = acc;
    }
  }
  // Delta: block reduction per row (wave shuffle, then 4 partials)
#pragma unroll
  for (int rr = 0; rr < 16; ++rr) {
    float v = rr < R ? dT[rr] * T[((int64_t)b * R + rr) * D + c] : 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((c & 63) == 0) sDel[rr][c >> 6] = v;
  }
  __syncthreads();
  if (c < 16) {
    Delta[(int64_t)b * 16 + c] = c < R ? sDel[c][0] + sDel[c][1] + sDel[c][2] + sDel[c][3] : 0.f;
    LSEp[(int64_t)b * 16 + c] = c < R ? LSE[(int64_t)b * R + c] : 1.0e30f;
  }
#pragma unroll
  for (int rr = 0; rr < 16; ++rr) dTb[((int64_t)b * 16 + rr) * D + c] = (__bf16)dT[rr];
  bf16x8 row[4];
#pragma unroll
  for (int gq = 0; gq < 4; ++gq)
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
      const int rr = 4 * gq + (jj & 3);
      row[gq][jj] = jj < 4 ? (__bf16)dT[rr] : (__bf16)(rr < R ? Gf[rr * D + c] : 0.f);
    }
#pragma unroll
  for (int gq = 0; gq < 4; ++gq)
    *reinterpret_cast<bf16x8*>(TG + ((int64_t)b * D + c) * 32 + 8 * gq) = row[gq];
}

// dWv[f][c] += sum_{b,q} dO[b][q][f] T[b][j(f) m + q][c]     (deterministic: one thread per (f, c))
__global__ __launch_bounds__(256) void k_pma_dwv(const float* __restrict__ dO,
                                                 const float* __restrict__ T, int B, int m, int R,
                                                 float* __restrict__ dWv) {
  constexpr int D = 256;
  const int f = blockIdx.x, c = threadIdx.x, j = f / 32;
  // (fixed summation order; 8 terms fetched at a time - one by one the B*m dependent round trips
  //  of this loop were 35 us at B = 128)
  float acc = 0.f;
  const int n = B * m;
  int i = 0;
  for (; i + 8 <= n; i += 8) {
    float x[8], y[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int b = (i + u) / m, q = (i + u) - b * m;
      x[u] = dO[((int64_t)b * m + q) * D + f];
      y[u] = T[((int64_t)b * R + j * m + q) * D + c];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = fmaf(x[u], y[u], acc);
  }
  for (; i < n; ++i) {
    const int b = i / m, q = i - b * m;
    acc = fmaf(dO[((int64_t)b * m + q) * D + f], T[((int64_t)b * R + j * m + q) * D + c], acc);
  }
  dWv[f * D + c] += acc;
}

__global__ __launch_bounds__(256, 1) void k_pma_bwd256(const PmaArgs a) {
  constexpr int DK = 256, FT = DK / 16, KS = DK / 32, TB = 32 * DK * 2, PD = 40;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sX = smem;                                 // 4 x 16 KiB ; later the dG slabs (4 x 16 KiB)
  char* sTG = sX + 4 * TB;                         // [256][64 B]
  char* sDS = sTG + DK * 64;                       // 4 x 32 x PD
  float* sLSE = reinterpret_cast<float*>(sDS + 4 * 32 * PD);
  float* sDel = sLSE + 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int b = blockIdx.x, sp = blockIdx.y;
  const int per = (int)(((int64_t)(a.N + 127) / 128 + a.S - 1) / a.S) * 128;
  const int n_lo = sp * per, n_hi = (n_lo + per < a.N) ? n_lo + per : a.N;
  int len = a.N;
  if (a.lengths != nullptr) len = a.lengths[b] < a.N ? a.lengths[b] : a.N;
  for (int i = tid; i < DK * 4; i += 256)
    reinterpret_cast<uint4*>(sTG)[i] = reinterpret_cast<const uint4*>(a.TG + (int64_t)b * DK * 32)[i];
  if (tid < 16) {
    sLSE[tid] = a.LSEp[(int64_t)b * 16 + tid];
    sDel[tid] = a.Delta[(int64_t)b * 16 + tid];
  }
  bf16x8 gA[KS], tA[KS];                   // rows r of G' / dT as A operands [row = r][k = c]
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    gA[ks] = *reinterpret_cast<const bf16x8*>(a.Gb + (int64_t)r * DK + 32 * ks + 8 * g);
    tA[ks] = *reinterpret_cast<const bf16x8*>(a.dTb + ((int64_t)b * 16 + r) * DK + 32 * ks + 8 * g);
  }
  __syncthreads();
  const float4 l4 = *reinterpret_cast<const float4*>(&sLSE[4 * g]);
  const float4 d4 = *reinterpret_cast<const float4*>(&sDel[4 * g]);
  const float lse[4] = {l4.x, l4.y, l4.z, l4.w}, del[4] = {d4.x, d4.y, d4.z, d4.w};
  char* myX = sX + wave * TB;
  char* myDS = sDS + wave * 32 * PD;
  f32x4 dG[FT];
#pragma unroll
  for (int ft = 0; ft < FT; ++ft) dG[ft] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr float LN2 = 0.6931471805599453f;
  // The wave's next tile is fetched into registers (64 VGPRs - the kernel runs one wave per SIMD,
  // there are 512) while the current one is worked on: without it every tile starts with an
  // exposed round trip to memory.
  bf16x8 nx[16];
  auto fetch_tile = [&](int n0) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int c = lane + 64 * e, row = c >> 5, ch = c & 31;
      const int nr = n0 + row;
      // (rows past the end of the range read its last row and are zeroed when the tile goes to LDS,
      //  one iteration later: zeroed here, hipcc waits for every load right behind its issue -
      //  vmcnt(15) ... vmcnt(0) - and the tile that was meant to arrive during the current tile's
      //  arithmetic is waited for before that arithmetic starts)
      nx[e] = *reinterpret_cast<const bf16x8*>(
          a.X + ((int64_t)b * a.N + (nr < n_hi ? nr : n_hi - 1)) * DK + ch * 8);
    }
  };
  if (n_lo + wave * 32 < n_hi) fetch_tile(n_lo + wave * 32);
  for (int n0 = n_lo + wave * 32; n0 < n_hi; n0 += 128) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int c = lane + 64 * e, row = c >> 5, ch = c & 31;
      bf16x8 v = nx[e];
      if (n0 + row >= n_hi) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (__bf16)0.f;
      }
      *reinterpret_cast<bf16x8*>(myX + tr_off256(row, ch)) = v;
    }
    if (n0 + 128 < n_hi) fetch_tile(n0 + 128);
    bf16x8 pds[2];                          // B operand [k = (P rows | dS rows)][col = point]
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) {
      f32x4 sv = {0.f, 0.f, 0.f, 0.f}, da = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 xr =
            *reinterpret_cast<const bf16x8*>(myX + tr_off256(16 * pb + r, 4 * ks + g));
        sv = mfma32(gA[ks], xr, sv);        // rows = score rows 4g+e, column = point 16 pb + r
        da = mfma32(tA[ks], xr, da);
      }
      const bool key = n0 + 16 * pb + r < len;
      f32x4 p, ds;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        p[e] = key ? __builtin_amdgcn_exp2f(sv[e] - lse[e]) : 0.f;
        ds[e] = LN2 * p[e] * (da[e] - del[e]);
      }
      pds[pb] = pack8(p, ds);
      *reinterpret_cast<bf16x4*>(myDS + (16 * pb + r) * PD + 8 * g) = pack4(ds);
    }
    // dG[r][c] += sum_points dS[r][pt] X[pt][c]   (before the X tile is re-used for dX)
    const bf16x8 dsa = tr_frag_small(myDS, PD, 0, lane);
#pragma unroll
    for (int ft = 0; ft < FT; ++ft) dG[ft] = mfma32(dsa, tr_frag256(myX, ft, lane), dG[ft]);
    if (a.dX != nullptr) {
      // dX tile [32 points][256] assembled in the wave's own LDS tile (its X is no longer needed)
      // and stored / accumulated in 16-byte pieces of full rows - straight from the accumulator
      // layout it was 32 store instructions of 8 bytes per lane, 16 rows x 32 bytes each
#pragma unroll
      for (int ft = 0; ft < FT; ++ft) {
        const bf16x8 tg = *reinterpret_cast<const bf16x8*>(sTG + (16 * ft + r) * 64 + 16 * g);
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
          const f32x4 dx = mfma32(tg, pds[pb], f32x4{0.f, 0.f, 0.f, 0.f});
          *reinterpret_cast<bf16x4*>(myX + swz(16 * pb + r, 2 * ft + (g >> 1), 2 * DK) + 8 * (g & 1)) =
              pack4(dx);
        }
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int c = lane + 64 * e, row = c >> 5, ch = c & 31;
        const int n = n0 + row;
        bf16x8 v = *reinterpret_cast<const bf16x8*>(myX + swz(row, ch, 2 * DK));
        __bf16* pd = a.dX + ((int64_t)b * a.N + (n < n_hi ? n : n_hi - 1)) * DK + ch * 8;
        if (a.accumulate_dx) {              // (wave-uniform; the load itself is unconditional)
          const bf16x8 o = *reinterpret_cast<const bf16x8*>(pd);
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] = (__bf16)((float)v[k] + (float)o[k]);
        }
        if (n < n_hi) *reinterpret_cast<bf16x8*>(pd) = v;
      }
    }
  }
  __syncthreads();
  float* slab = reinterpret_cast<float*>(sX) + wave * 16 * DK;
#pragma unroll
  for (int ft = 0; ft < FT; ++ft)
#pragma unroll
    for (int e = 0; e < 4; ++e) slab[(4 * g + e) * DK + 16 * ft + r] = dG[ft][e];
  __syncthreads();
  const float* s0 = reinterpret_cast<const float*>(sX);
  // (no atomics: one [16][256] slab per workgroup, k_slab_sum adds them in a fixed order)
  float* out = a.DG + ((int64_t)b * a.S + sp) * 16 * DK;
  for (int i = tid; i < 16 * DK; i += 256)
    out[i] = s0[i] + s0[16 * DK + i] + s0[2 * 16 * DK + i] + s0[3 * 16 * DK + i];
}

}  // namespace

// ---- launchers (declared in d256_bf16.hpp) ------------------------------------------------
int rowgemm256_proj(const __bf16* X, const __bf16* WP, const float* bias, __bf16* Y, int B, int N,
                    hipStream_t st) {
  RowGemmArgs a{X, WP, bias, nullptr, nullptr, nullptr, Y, nullptr, B, N, (int)cdiv(N, TP), 0};
  return launch_rowgemm<RG_PROJ>(a, st);
}
int rowgemm256_bwd_o(const __bf16* dY, const uint32_t* mask, const __bf16* WoTP, __bf16* dZ,
                     __bf16* dO, int B, int N, hipStream_t st) {
  RowGemmArgs a{dY, WoTP, nullptr, nullptr, mask, nullptr, dO, dZ, B, N, (int)cdiv(N, TP), 0};
  return launch_rowgemm<RG_BWD_O>(a, st);
}
int rowgemm256_dx(const __bf16* G, const __bf16* WTP, __bf16* dX, int B, int N, int accumulate,
                  hipStream_t st) {
  RowGemmArgs a{G, WTP, nullptr, nullptr, nullptr, nullptr, dX, nullptr, B, N, (int)cdiv(N, TP),
                accumulate};
  return launch_rowgemm<RG_BWD_Q>(a, st);
}
int rowgemm256_fwd_o(const __bf16* O, const __bf16* WoP, const float* bo, __bf16* Y, uint32_t* mask,
                     int B, int N, hipStream_t st) {
  RowGemmArgs a{O, WoP, bo, nullptr, nullptr, mask, Y, nullptr, B, N, (int)cdiv(N, TP), 0};
  return launch_rowgemm<RG_FWD_O>(a, st);
}
// fp8 (e4m3) operands: W8 = image of s * W written by prep_weight_f8, inv_scale[0] = 1 / s
int rowgemm256_proj_f8(const __bf16* X, const void* W8, const float* inv_scale, const float* bias,
                       __bf16* Y, int B, int N, hipStream_t st) {
  RowGemmArgs a{X, reinterpret_cast<const __bf16*>(W8), bias, inv_scale, nullptr, nullptr, Y,
                nullptr, B, N, (int)cdiv(N, TP), 0};
  return launch_rowgemm_nb<RG_PROJ, 2, true>(a, st);
}
int rowgemm256_fwd_o_f8(const __bf16* O, const void* W8, const float* inv_scale, const float* bo,
                        __bf16* Y, uint32_t* mask, int B, int N, hipStream_t st) {
  RowGemmArgs a{O, reinterpret_cast<const __bf16*>(W8), bo, inv_scale, nullptr, mask, Y, nullptr, B,
                N, (int)cdiv(N, TP), 0};
  return launch_rowgemm_nb<RG_FWD_O, 2, true>(a, st);
}

int attn1_bwd256_parts(int B, int N) {
  // point ranges per set so that B * parts workgroups (2 per CU) cover the chip
  int parts = 1;
  const int tiles = (int)cdiv(N, 32);
  while (parts * 2 <= tiles && B * parts < 512) parts *= 2;
  return parts;
}
int attn1_bwd256(const __bf16* dO, const __bf16* QpS, const __bf16* KpP, const __bf16* VpP,
                 const __bf16* Kt, __bf16* dQp, float* dKpPart, float* dVpPart, float* dKp,
                 float* dVp, int B, int N, hipStream_t st) {
  constexpr int D = 256;
  int parts = attn1_bwd256_parts(B, N);
  // (k_attn1_bwd2 holds 152 KiB of LDS: one workgroup per CU, so half the ranges of the
  //  two-per-CU form; the partial buffers are sized for the larger count)
  if (B * parts > 256 && parts > 1) parts /= 2;
  const int ppp = (int)cdiv(cdiv(N, 32), parts) * 32;
  Attn1BwdArgs a{dO, QpS, KpP, VpP, Kt, dQp, dKpPart, dVpPart, B, N, parts, ppp,
                 1.0f / sqrtf((float)D), 1.4426950408889634f / sqrtf((float)D)};
  static std::once_flag once;
  std::call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn1_bwd2<D>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  hipLaunchKernelGGL((k_attn1_bwd2<D>), dim3(B * parts), dim3(64 * (D / 32)),
                     (size_t)5 * 32 * D * 2 + (size_t)(D / 32) * 4 * 32 * 72, st, a);
  PCA_TRY(check_launch("k_attn1_bwd"));
  hipLaunchKernelGGL(k_sum_parts256, dim3((unsigned)cdiv((int64_t)B * 32 * D, 256)), dim3(256), 0,
                     st, dKpPart, dVpPart, dKp, dVp, B, parts, 32 * D);
  return check_launch("k_sum_parts256");
}

// fc_o adjoint + attention adjoint in one launch (k_attn1_bwd3); WoT: transposed natural image
int attn1_bwd256_fused(const __bf16* dY, const uint32_t* mask, const __bf16* WoT, const __bf16* QpS,
                       const __bf16* KpP, const __bf16* VpP, const __bf16* Kt, __bf16* dZ,
                       __bf16* dQp, float* dKpPart, float* dVpPart, float* dKp, float* dVp, int B,
                       int N, hipStream_t st, const float* Xs, const float* WqF, const float* bq,
                       int dq) {
  constexpr int D = 256;
  int parts = attn1_bwd256_parts(B, N);
  if (B * parts > 256 && parts > 1) parts /= 2;            // 155 KiB of LDS: one workgroup per CU
  const int ppp = (int)cdiv(cdiv(N, 32), parts) * 32;
  Attn1Bwd3Args a{};
  a.base = Attn1BwdArgs{nullptr, QpS, KpP, VpP, Kt, dQp, dKpPart, dVpPart, B, N, parts, ppp,
                        1.0f / sqrtf((float)D), 1.4426950408889634f / sqrtf((float)D)};
  a.dY = dY; a.mask = mask; a.WoT = WoT; a.dZ = dZ;
  a.tiles128 = (int)cdiv(N, 128);
  a.Xs = Xs; a.WqF = WqF; a.bq = bq; a.dq = dq;
  const bool smallq = QpS == nullptr;
  PCA_REQUIRE(!smallq || (Xs && WqF && bq && dq >= 1 && dq <= 4),
              "attn1_bwd256_fused: no saved Qp and no points to recompute it from");
  static std::once_flag once;
  std::call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn1_bwd3<D, false, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn1_bwd3<D, true, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn1_bwd3<D, false, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn1_bwd3<D, true, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  const size_t lds = (size_t)5 * 32 * D * 2 + 2048 + (size_t)8 * 4 * 32 * 72;
  const dim3 grid(B * parts), block(512);
  // dZ == nullptr: dZ stays inside the kernel (the weight-gradient job applies the mask to dY itself)
  if (dZ != nullptr) {
    if (smallq) hipLaunchKernelGGL((k_attn1_bwd3<D, true, true>), grid, block, lds, st, a);
    else hipLaunchKernelGGL((k_attn1_bwd3<D, false, true>), grid, block, lds, st, a);
  } else {
    if (smallq) hipLaunchKernelGGL((k_attn1_bwd3<D, true, false>), grid, block, lds, st, a);
    else hipLaunchKernelGGL((k_attn1_bwd3<D, false, false>), grid, block, lds, st, a);
  }
  PCA_TRY(check_launch("k_attn1_bwd3"));
  hipLaunchKernelGGL(k_sum_parts256, dim3((unsigned)cdiv((int64_t)B * 32 * D, 256)), dim3(256), 0,
                     st, dKpPart, dVpPart, dKp, dVp, B, parts, 32 * D);
  return check_launch("k_sum_parts256");
}

// workgroups per job: enough to stream from every CU, few enough that the slab pass (256 KiB per
// workgroup written + read) stays small against the 1 KiB per row the job reads
int wgrad256_nwg(int64_t maxM) {
  // long jobs: one workgroup per CU; short ones ([B*m] rows): 64 rows each
  int nwg = (int)cdiv(maxM, maxM >= 65536 ? 1024 : 64);
  if (nwg > 256) nwg = 256;
  return nwg < 1 ? 1 : nwg;
}
// Room for `njobs` jobs of AT MOST maxM rows each.  wgrad256_nwg() is not monotonic in the row
// count (64 rows per workgroup below 65 536 rows, 1024 above), and one workspace serves the long
// [B N]-row jobs as well as the short [B m]-row ones of the same block: size for the largest
// workgroup count any job of up to maxM rows can get.
size_t wgrad256_ws_bytes(int njobs, int64_t maxM) {
  size_t nwg = (size_t)cdiv(maxM < 1 ? 1 : maxM, 64);
  if (nwg > 256) nwg = 256;
  return align256((size_t)njobs * nwg * 256 * 256 * sizeof(float)) +
         align256((size_t)njobs * nwg * 256 * sizeof(float));
}
int wgrad256_launch(const Wgrad256Jobs& jobs, void* ws, hipStream_t st) {
  return wgrad256_launch_t(jobs, ws, false, st);
}
static bool wgrad256_use_dma() {      // PCA_WGRAD256_DMA=0: the register-staged kernel (A/B measurements)
  static const bool on = [] { const char* e = getenv("PCA_WGRAD256_DMA"); return !(e && e[0] == '0'); }();
  return on;
}
// May a bf16 job hand over dY + the forward's ReLU mask instead of dZ?  (PCA_D256_DZ_MASK=0: no)
bool wgrad256_masked_ok(int64_t rows_per_set) {
  static const bool on = [] { const char* e = getenv("PCA_D256_DZ_MASK"); return !(e && e[0] == '0'); }();
  return on && wgrad256_use_dma() && rows_per_set % 128 == 0;
}
int wgrad256_launch_t(const Wgrad256Jobs& jobs, void* ws, bool f32_operands, hipStream_t st) {
  if (jobs.n == 0) return PCA_OK;
  int64_t maxM = 0;
  for (int i = 0; i < jobs.n; ++i) {
    maxM = jobs.j[i].M > maxM ? jobs.j[i].M : maxM;
    PCA_REQUIRE(((uintptr_t)jobs.j[i].dW & 15) == 0, "wgrad256: dW must be 16-byte aligned");
  }
  if (maxM == 0) return PCA_OK;
  const int nwg = wgrad256_nwg(maxM);
  int rpw = (int)cdiv(cdiv(maxM, nwg), 64) * 64;
  Carver c(ws);
  float* slabs = c.take<float>((size_t)jobs.n * nwg * 256 * 256);
  float* bslabs = c.take<float>((size_t)jobs.n * nwg * 256);
  double rows = 0;
  for (int i = 0; i < jobs.n; ++i) rows += (double)jobs.j[i].M;
  bool shared = jobs.n > 1;
  for (int i = 1; i < jobs.n; ++i) shared = shared && jobs.j[i].A == jobs.j[0].A;
  // algorithmic bytes: both operands of every job once (a shared A operand once for all jobs)
  const double eb = f32_operands ? 4.0 : 2.0;
  const double opbytes = shared ? eb * 256 * (rows + (double)jobs.j[0].M) : 2.0 * eb * 256 * rows;
  ProfScope ps(PCA_K_WGRAD, st, 2.0 * rows * 256 * 256, opbytes);
  if (shared) rpw = -rpw;
  const bool use_dma = wgrad256_use_dma();
  for (int i = 0; i < jobs.n; ++i)
    PCA_REQUIRE(jobs.j[i].mask == nullptr || (use_dma && !f32_operands),
                "wgrad256: a masked job needs the LDS-DMA kernel");
  if (f32_operands) {
    hipLaunchKernelGGL(k_wgrad256<float>, dim3(nwg * jobs.n), dim3(512), 0, st, jobs, rpw, slabs,
                       bslabs);
  } else if (use_dma) {
    static std::once_flag once;
    std::call_once(once, [] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_wgrad256_dma),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    for (int i = 0; i < jobs.n; ++i)
      PCA_REQUIRE(jobs.j[i].mask == nullptr || jobs.j[i].M % 32 == 0, "wgrad256: masked job rows");
    hipLaunchKernelGGL(k_wgrad256_dma, dim3(nwg * jobs.n), dim3(512),
                       (size_t)4 * 2 * 32 * 256 * 2 + 4 * 1024, st, jobs, rpw, slabs, bslabs);
  } else {
    hipLaunchKernelGGL(k_wgrad256<__bf16>, dim3(nwg * jobs.n), dim3(512), 0, st, jobs, rpw, slabs,
                       bslabs);
  }
  ps.end();
  PCA_TRY(check_launch("k_wgrad256"));
  hipLaunchKernelGGL(k_wgrad256_sum, dim3((256 * 256 + 256 + 255) / 256, jobs.n), dim3(256), 0, st,
                     jobs, nwg, rpw, slabs, bslabs);
  return check_launch("k_wgrad256_sum");
}

int cvt_f32_bf16(const float* s, __bf16* d, int64_t n, hipStream_t st) {
  hipLaunchKernelGGL(k_cvt_f32_bf16, dim3((unsigned)cdiv(n / 4, 256)), dim3(256), 0, st, s, d,
                     n / 4);
  return check_launch("k_cvt_f32_bf16");
}
int cvt_bf16_f32(const __bf16* s, float* d, int64_t n, int accumulate, hipStream_t st) {
  hipLaunchKernelGGL(k_cvt_bf16_f32, dim3((unsigned)cdiv(n / 4, 256)), dim3(256), 0, st, s, d,
                     n / 4, accumulate);
  return check_launch("k_cvt_bf16_f32");
}
int kv_proj_small256(const float* X, int64_t M, int dk, const float* Wk, const float* bk,
                     const float* Wv, const float* bv, __bf16* Kp, __bf16* Vp, hipStream_t st) {
  hipLaunchKernelGGL(k_kv_proj_small, dim3((unsigned)cdiv(M * 64, 256)), dim3(256), 0, st, X, M, dk,
                     Wk, bk, Wv, bv, 256, Kp, Vp);
  return check_launch("k_kv_proj_small");
}

int fq_splits256(int B, int N) {
  int S = 1;
  const int tiles = (int)cdiv(N, 32);
  while (S * 2 <= tiles && B * S < 1024 && S < 16) S *= 2;
  return S;
}
int fq_attn_fwd256(const __bf16* Kp, const __bf16* Vp, const float* Qp, int B, int N, int m,
                   const int32_t* lengths, float* Op, float* Mp, float* Lp, float* O, float* LSE,
                   hipStream_t st) {
  constexpr int D = 256;
  const int S = fq_splits256(B, N), QT = m > 16 ? 2 : 1, MQ = 16 * QT;
  FqArgs a{};
  a.Kp = Kp; a.Vp = Vp; a.Qp = Qp; a.Op = Op; a.Mp = Mp; a.Lp = Lp;
  a.B = B; a.N = N; a.m = m; a.S = S; a.lengths = lengths;
  a.scale = 1.0f / sqrtf((float)D);
  a.scale_log2e = 1.4426950408889634f * a.scale;
  const size_t lds = (size_t)(D / 32) * 32 * 72;
  constexpr bool v1 = false;       // (the per-wave global-traffic form: only for m <= 16 below)
  int S2 = S;
  if (QT == 2 && !v1) {
    while (S2 > 1 && B * S2 > 256) S2 /= 2;       // 82 KiB of LDS: one workgroup per CU
    a.S = S2;
    static std::once_flag once;
    std::call_once(once, [] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_fq_attn_fwd2),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    hipLaunchKernelGGL(k_fq_attn_fwd2, dim3(B, S2), dim3(512), (size_t)4 * 32 * D * 2 + lds, st, a);
  } else if (QT == 2) {
    hipLaunchKernelGGL((k_fq_attn_fwd<D, 2>), dim3(B, S), dim3(512), lds, st, a);
  } else {
    hipLaunchKernelGGL((k_fq_attn_fwd<D, 1>), dim3(B, S), dim3(512), lds, st, a);
  }
  PCA_TRY(check_launch("k_fq_attn_fwd"));
  hipLaunchKernelGGL(k_fq_merge, dim3((unsigned)cdiv((int64_t)B * m * D, 256)), dim3(256), 0, st,
                     Op, Mp, Lp, Qp, B, S2, m, D, MQ, O, LSE);
  return check_launch("k_fq_merge");
}
// fc_k / fc_v over the keys + the attention in one launch (k_fq_proj_fwd), then the merge; m = 32
int fq_proj_attn_fwd256(const __bf16* X, const __bf16* WkB, const __bf16* WvB, const float* bk,
                        const float* bv, const float* Qp, int B, int N, int m,
                        const int32_t* lengths, __bf16* Kp, __bf16* Vp, float* Op, float* Mp,
                        float* Lp, float* O, float* LSE, hipStream_t st, const float* inv_scale) {
  constexpr int D = 256;
  PCA_REQUIRE(m > 16 && m <= 32, "fq_proj_attn_fwd256: m = %d", m);
  int S2 = fq_splits256(B, N);
  while (S2 > 1 && B * S2 > 256) S2 /= 2;         // 98 KiB of LDS: one workgroup per CU
  FqProjArgs a{};
  a.f.Qp = Qp; a.f.Op = Op; a.f.Mp = Mp; a.f.Lp = Lp;
  a.f.B = B; a.f.N = N; a.f.m = m; a.f.S = S2; a.f.lengths = lengths;
  a.f.scale = 1.0f / sqrtf((float)D);
  a.f.scale_log2e = 1.4426950408889634f * a.f.scale;
  a.X = X; a.WkB = WkB; a.WvB = WvB; a.bk = bk; a.bv = bv; a.KpO = Kp; a.VpO = Vp;
  a.inv_scale = inv_scale;
  static std::once_flag once;
  std::call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_fq_proj_fwd<false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_fq_proj_fwd<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  const size_t lds = (size_t)5 * 32 * D * 2 + (size_t)(D / 32) * 32 * 72 +
                     (inv_scale != nullptr ? (size_t)32 * D : 0);
  if (inv_scale != nullptr) hipLaunchKernelGGL(k_fq_proj_fwd<true>, dim3(B, S2), dim3(512), lds, st, a);
  else hipLaunchKernelGGL(k_fq_proj_fwd<false>, dim3(B, S2), dim3(512), lds, st, a);
  PCA_TRY(check_launch("k_fq_proj_fwd"));
  hipLaunchKernelGGL(k_fq_merge, dim3((unsigned)cdiv((int64_t)B * m * D, 256)), dim3(256), 0, st,
                     Op, Mp, Lp, Qp, B, S2, m, D, 32, O, LSE);
  return check_launch("k_fq_merge");
}
int fq_attn_bwd256(const __bf16* Kp, const __bf16* Vp, const float* Qp, const float* dO,
                   const float* O, const float* LSE, float* Delta, int B, int N, int m,
                   const int32_t* lengths, __bf16* dKp, __bf16* dVp, float* dQpPart, float* dOt,
                   hipStream_t st) {
  constexpr int D = 256;
  const int S = fq_splits256(B, N), QT = m > 16 ? 2 : 1, MQ = 16 * QT;
  hipLaunchKernelGGL(k_fq_delta, dim3((unsigned)cdiv((int64_t)B * (D / 32) * m, 256)), dim3(256), 0,
                     st, dO, O, Qp, B, m, D, MQ, Delta);
  PCA_TRY(check_launch("k_fq_delta"));
  FqArgs a{};
  a.Kp = Kp; a.Vp = Vp; a.Qp = Qp; a.dOa = dO; a.LSE = LSE; a.Delta = Delta;
  a.dKp = dKp; a.dVp = dVp; a.dQpPart = dQpPart;
  a.B = B; a.N = N; a.m = m; a.S = S; a.lengths = lengths;
  a.scale = 1.0f / sqrtf((float)D);
  a.scale_log2e = 1.4426950408889634f * a.scale;
  const size_t lds = (size_t)(D / 32) * 32 * 72;
  constexpr bool v1 = false;       // (the per-wave global-traffic form: only for m <= 16 below)
  int S2 = S;
  if (QT == 2 && !v1) {
    // full-line traffic through LDS tiles: 114 KiB per workgroup, one per CU - fewer point ranges
    // (the partial buffers are sized for S)
    while (S2 > 1 && B * S2 > 256) S2 /= 2;
    a.S = S2;
    static std::once_flag once;
    std::call_once(once, [] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_fq_attn_bwd2),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    hipLaunchKernelGGL(k_fq_attn_bwd2, dim3(B, S2), dim3(512), (size_t)6 * 32 * D * 2 + lds, st, a);
  } else if (QT == 2) {
    hipLaunchKernelGGL((k_fq_attn_bwd<D, 2>), dim3(B, S), dim3(512), lds, st, a);
  } else {
    hipLaunchKernelGGL((k_fq_attn_bwd<D, 1>), dim3(B, S), dim3(512), lds, st, a);
  }
  PCA_TRY(check_launch("k_fq_attn_bwd"));
  hipLaunchKernelGGL(k_fq_dq_sum, dim3((unsigned)cdiv((int64_t)B * m * D, 256)), dim3(256), 0, st,
                     dO, dQpPart, B, S2, m * D, dOt);
  return check_launch("k_fq_dq_sum");
}

size_t wgrad_small256_ws_bytes(int64_t M) {
  return align256((size_t)cdiv(M, M >= 65536 ? 512 : 128) * 256 * 5 * sizeof(float));
}
int wgrad_small256(const __bf16* G, const float* X, int64_t M, int dq, float* dW, float* db,
                   void* ws, hipStream_t st) {
  const int rpw = M >= 65536 ? 512 : 128;
  const int nwg = (int)cdiv(M, rpw);
  float* slabs = reinterpret_cast<float*>(ws);
  hipLaunchKernelGGL(k_wgrad_small256, dim3((unsigned)nwg), dim3(256), 0, st, G, X, M, dq, rpw,
                     slabs);
  PCA_TRY(check_launch("k_wgrad_small256"));
  hipLaunchKernelGGL(k_wgrad_small256_sum, dim3(256 * 5 / 64), dim3(1024), 0, st, slabs, nwg, dq, dW,
                     db);
  return check_launch("k_wgrad_small256_sum");
}
int epi_small_fwd256(const float* T, const float* Qp, const float* Wv, const float* bv, int B, int m,
                     int dk, float* O, hipStream_t st) {
  if (dk == 256) {
    hipLaunchKernelGGL(k_epi_wide_fwd, dim3((unsigned)cdiv((int64_t)B * m * 256, 32)), dim3(256), 0,
                       st, T, Qp, Wv, bv, B, m, O);
    return check_launch("k_epi_wide_fwd");
  }
  hipLaunchKernelGGL(k_epi_small_fwd, dim3((unsigned)cdiv((int64_t)B * m * 256, 256)), dim3(256), 0,
                     st, T, Qp, Wv, bv, B, m, 256, dk, O);
  return check_launch("k_epi_small_fwd");
}
size_t epi_small_bwd256_ws_bytes(int B, int m) {
  return align256((size_t)cdiv((int64_t)B * m, 16) * 256 * 5 * sizeof(float));
}
int epi_small_bwd256(const float* dO, const float* T, const float* Wv, int B, int m, int dk,
                     float* dT, float* Delta, float* dWv, float* dbv, void* ws, hipStream_t st) {
  hipLaunchKernelGGL(k_epi_small_bwd, dim3((unsigned)cdiv((int64_t)B * 8 * m, 256)), dim3(256), 0,
                     st, dO, T, Wv, B, m, 256, dk, dT, Delta);
  PCA_TRY(check_launch("k_epi_small_bwd"));
  const int nwg = (int)cdiv((int64_t)B * m, 16);
  float* slabs = reinterpret_cast<float*>(ws);
  hipLaunchKernelGGL(k_epi_small_wv, dim3((unsigned)nwg), dim3(256), 0, st, dO, T, B, m, dk, 16, slabs);
  PCA_TRY(check_launch("k_epi_small_wv"));
  hipLaunchKernelGGL(k_wgrad_small256_sum, dim3(256 * 5 / 64), dim3(1024), 0, st, slabs, nwg, dk, dWv,
                     dbv);
  return check_launch("k_wgrad_small256_sum");
}

size_t pma_bwd256_slab_bytes(int B) {
  return align256((size_t)(B > 256 ? B : 256) * 16 * 256 * sizeof(float));
}
int slab_sum(const float* slabs, int S, int n, float* out, int accumulate, hipStream_t st) {
  PCA_REQUIRE(n % 4 == 0 && ((uintptr_t)out & 15) == 0 && ((uintptr_t)slabs & 15) == 0,
              "slab_sum: alignment");
  hipLaunchKernelGGL(k_slab_sum, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, slabs, S, n, out,
                     accumulate);
  return check_launch("k_slab_sum");
}
namespace {
// several slab sums in one launch (blockIdx.y = job); same walk and summation order as k_slab_sum
__global__ __launch_bounds__(256) void k_slab_sum_jobs(const SlabSumJobs jobs) {
  __shared__ float4 red[4 * 64];
  slab_sum_body(jobs.j[blockIdx.y], blockIdx.x, threadIdx.x, red);
}
}  // namespace
int slab_sum_jobs(const SlabSumJobs& J, hipStream_t st) {
  if (J.n == 0) return PCA_OK;
  int nmax = 0;
  for (int i = 0; i < J.n; ++i) {
    PCA_REQUIRE(slab_sum_job_ok(J.j[i]), "slab_sum_jobs: alignment");
    nmax = J.j[i].n > nmax ? J.j[i].n : nmax;
  }
  hipLaunchKernelGGL(k_slab_sum_jobs, dim3((unsigned)cdiv(nmax, 256), (unsigned)J.n), dim3(256), 0,
                     st, J);
  return check_launch("k_slab_sum_jobs");
}
int pma_splits256(int B, int N) {
  int S = 1;
  const int tiles = (int)cdiv(N, 128);
  while (S * 2 <= tiles && B * S < 512 && S < 16) S *= 2;
  return S;
}
int pma_attn_fwd256(const __bf16* X, const __bf16* Gb, int B, int N, int R, const int32_t* lengths,
                    float* Tp, float* Mp, float* Lp, float* T, float* LSE, hipStream_t st) {
  PmaArgs a{};
  a.X = X; a.Gb = Gb; a.Tp = Tp; a.Mp = Mp; a.Lp = Lp;
  a.B = B; a.N = N; a.R = R; a.S = pma_splits256(B, N); a.lengths = lengths;
  static std::once_flag once;
  std::call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_pma_fwd256),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  hipLaunchKernelGGL(k_pma_fwd256, dim3(B, a.S), dim3(256), 4 * 32 * 256 * 2 + 3 * 64 * sizeof(float),
                     st, a);
  PCA_TRY(check_launch("k_pma_fwd256"));
  hipLaunchKernelGGL(k_pma_merge, dim3((unsigned)cdiv((int64_t)B * R * 256, 256)), dim3(256), 0, st,
                     Tp, Mp, Lp, B, a.S, R, T, LSE);
  return check_launch("k_pma_merge");
}
int pma_epi_bwd256(const float* dO, const float* T, const float* LSE, const float* Wv,
                   const float* Gf, int B, int m, int R, __bf16* dTb, __bf16* TG, float* Delta,
                   float* LSEp, float* dWv, hipStream_t st) {
  hipLaunchKernelGGL(k_pma_epi_bwd, dim3(B), dim3(256), 0, st, dO, T, LSE, Wv, Gf, m, R, dTb, TG,
                     Delta, LSEp);
  PCA_TRY(check_launch("k_pma_epi_bwd"));
  hipLaunchKernelGGL(k_pma_dwv, dim3(256), dim3(256), 0, st, dO, T, B, m, R, dWv);
  return check_launch("k_pma_dwv");
}
int pma_attn_bwd256(const __bf16* X, const __bf16* Gb, const __bf16* dTb, const __bf16* TG,
                    const float* LSEp, const float* Delta, int B, int N, int R,
                    const int32_t* lengths, __bf16* dX, int accumulate_dx, float* DG,
                    float* DGslabs, hipStream_t st) {
  PmaArgs a{};
  a.X = X; a.Gb = Gb; a.dTb = dTb; a.TG = TG; a.LSEp = LSEp; a.Delta = Delta;
  a.dX = dX; a.DG = DGslabs; a.accumulate_dx = accumulate_dx;
  a.B = B; a.N = N; a.R = R; a.lengths = lengths;
  int S = pma_splits256(B, N);
  while (S > 1 && B * S > 256) S /= 2;            // ~100 KiB of LDS: one workgroup per CU
  a.S = S;
  static std::once_flag once;
  std::call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_pma_bwd256),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  const size_t lds = 4 * 32 * 256 * 2 + 256 * 64 + 4 * 32 * 40 + 32 * sizeof(float);
  hipLaunchKernelGGL(k_pma_bwd256, dim3(B, S), dim3(256), lds, st, a);
  PCA_TRY(check_launch("k_pma_bwd256"));
  hipLaunchKernelGGL(k_slab_sum, dim3(16), dim3(256), 0, st, DGslabs, B * S, 16 * 256, DG, 0);
  return check_launch("k_slab_sum");
}

}  // namespace pca
