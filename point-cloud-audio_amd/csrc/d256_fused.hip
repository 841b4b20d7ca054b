// Single-launch forward of the many-queries block at d = 256 / 8 heads / m = 32 keys (ISAB's
// mab1(X, H), set_transformer-master/modules.py:53 with :19-33 inside):
//     Qp = fc_q(x) ; per head A = softmax(Qp_h Kp_h^T / sqrt d) ; O = Qp + A Vp ; Y = O + relu(fc_o(O))
//
// WAVE = HEAD, WEIGHTS IN REGISTERS.  The two 128 KiB weight images do not fit the 160 KiB of LDS
// next to anything else - but the register file of a CU is 512 KiB.  The workgroup is 8 waves;
// wave j owns head j, i.e. output features 32 j .. 32 j + 31 of BOTH projections, and keeps its
// [32 x 256] slices of Wq and Wo as MFMA A operands in 128 VGPRs for the whole launch.  What
// moves through LDS is only activations, in full tiles shared by the 8 waves:
//     X tile [64 points][256] (LDS-DMA, double buffered)  --GEMM1-->  Qp_h (registers, fp32)
//     --scores / softmax / A V, all inside the wave (its head's K / V slices: 16 VGPRs)-->  O_h
//     --own 32-column slice written to the O tile--  barrier  --GEMM2 over the whole O tile-->
//     Z_h ; Y_h = O_h + relu(Z_h + bo) --own slice of the Y tile-- barrier -- coalesced store.
// So X is read once and Y written once (the two-phase split it replaces read / wrote O in
// between: 2 extra passes over [B*N, 256]), no weight is ever re-read, and every global access is
// a full 128-byte line.  Roofline unit: 2 (dq d + d^2 + 2 m d) FLOP per point,
// 2 (dq + d) bytes per point.
#include "d256_bf16.hpp"

#include <math.h>

#include <mutex>

namespace pca {

namespace {

// 32-point tiles (2 blocks of 16): with 64 the kernel needs ~290 registers - over the 256 of two
// waves per SIMD, and every scratch reload is a VMEM access that drains the LDS-DMA queue
constexpr int D = 256, MI = 32, P = 32, NBK = P / 16;
constexpr int ROWB = D * 2, TILEB = P * ROWB;               // 16 KiB per tile
constexpr int DMA_PER_WAVE = TILEB / 1024 / 8;              // 1 KiB pieces per wave and tile

struct FusedFwdArgs {
  const void* X;          // [B*N][256] bf16, or fp32 [B*N][dq] for dq <= 4
  const __bf16* Wq;       // [256][256] bf16, natural (nullptr for dq <= 4)
  const float* WqF;       // dq <= 4: fp32 [256][dq]
  const float* bq;
  const __bf16* KpP;      // [B][32][256] (K-permuted features inside each head)
  const __bf16* Vt;       // [B][256][32] (keys in perm32 order)
  const __bf16* Wo;       // [256][256] bf16, natural (F8O: fp8 e4m3 bytes of s * Wo, natural)
  const float* inv_o;     // F8O: 1 / s
  const float* bo;
  __bf16* Y;              // [B*N][256]
  __bf16 *QpS, *OS;       // saved for the backward (nullable)
  uint32_t* mask;         // ReLU mask bits in the layout of mab1_mask_index<256> (nullable)
  int B, N, dq, tiles_per_set, units_per_wg;
  float scale_log2e;
  int ablate;             // measurement builds (-DPCA_FWD_ABLATE): PCA_AB_ABLATE bit mask, see the kernel
};

// 16-byte chunk c16 of row `row` of a [64][256] bf16 tile (swizzled like the weight images)
__device__ __forceinline__ int toff(int row, int c16) { return swz(row, c16, ROWB); }

typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

#define PCA_WAIT_VM_CASE(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
__device__ __forceinline__ void wait_vm(int n) {
  switch (n) {
    PCA_WAIT_VM_CASE(2) PCA_WAIT_VM_CASE(3) PCA_WAIT_VM_CASE(4) PCA_WAIT_VM_CASE(5)
    PCA_WAIT_VM_CASE(6) PCA_WAIT_VM_CASE(7) PCA_WAIT_VM_CASE(8) PCA_WAIT_VM_CASE(9)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}

#ifdef PCA_FWD_STAMPS
// (measurement build only, -DPCA_FWD_STAMPS: cycle stamps of workgroup 0, every wave, the first 64
//  tiles.  The stamps are global stores: they perturb every vmcnt wait of the stamped workgroup.)
__device__ long long g_fwd_stamps[8 * 64 * 16];
#define FWD_STAMP(i)                                                                          \
  do {                                                                                        \
    if (blockIdx.x == 0 && lane == 0 && unit - u0 < 64)                                       \
      g_fwd_stamps[(j * 64 + (unit - u0)) * 16 + (i)] = (long long)__builtin_readcyclecounter(); \
  } while (0)
#else
#define FWD_STAMP(i)
#endif

// F8O (PCA_MODE_FP8): fc_o with fp8 e4m3 operands - Wo slices as fp8 (32 registers instead of 64),
// the O fragments converted in registers, Z rescaled by 1 / s before the bias
template <bool SMALL, bool F8O = false>
__global__ __launch_bounds__(512, 2) void k_isab1_fwd256(const FusedFwdArgs a) {
  constexpr int KS = D / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // every tile exists twice (iteration parity): the global stores of tile i are issued in
  // iteration i + 1, so no wait of iteration i + 1 ever covers a store that was just issued
  char* sXb = smem;                         // X tiles          [2][TILEB]
  char* sOb = smem + 2 * TILEB;             // O tiles
  char* sQb = smem + 4 * TILEB;             // Qp tiles (training: saved for the backward)
  char* sYb = smem + 6 * TILEB;             // Y tiles
  uint32_t* sMaskb = reinterpret_cast<uint32_t*>(smem + 8 * TILEB);   // [2][NBK blocks][2 words][64]
  float* sBias = reinterpret_cast<float*>(sMaskb + 2 * NBK * 2 * 64); // bq [256], bo [256]
  const int tid = threadIdx.x, lane = tid & 63;
  const int j = __builtin_amdgcn_readfirstlane(tid >> 6);              // head of this wave (scalar)
  const int r = lane & 15, g = lane >> 4;

  // ---- this head's weight slices: A operands [row = feature 32 j + 16 t + r][k = 32 s + 8 g ..] ----
  bf16x8 wqa[SMALL ? 1 : KS][2], woa[F8O ? 1 : KS][2];
  f8x8 woa8[F8O ? KS : 1][2];
  const float inv_o = F8O ? a.inv_o[0] : 1.f;
#pragma unroll
  for (int s = 0; s < KS; ++s)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int64_t o = (int64_t)(32 * j + 16 * t + r) * D + 32 * s + 8 * g;
      if (F8O)
        woa8[F8O ? s : 0][t] =
            *reinterpret_cast<const f8x8*>(reinterpret_cast<const uint8_t*>(a.Wo) + o);
      else
        woa[F8O ? 0 : s][t] = *reinterpret_cast<const bf16x8*>(a.Wo + o);
      if (!SMALL) wqa[s][t] = *reinterpret_cast<const bf16x8*>(a.Wq + o);
    }
  // layer 1: fc_q rows of the features this lane holds in the accumulator layout
  float wqs[SMALL ? 2 : 1][4][4];
  // (the biases live in LDS: 16 registers less in a kernel at the 256-register limit of two waves
  //  per SIMD, and no ordinary global load inside the loop - hipcc drains the LDS-DMA queue with
  //  vmcnt(0) at the first use of any such load)
  if (tid < D) {
    sBias[tid] = a.bq[tid];
    sBias[D + tid] = a.bo[tid];
  }
  auto bias4 = [&](int which, int t) {
    const float4 q4 = *reinterpret_cast<const float4*>(sBias + which * D + 32 * j + 16 * t + 4 * g);
    return f32x4{q4.x, q4.y, q4.z, q4.w};
  };
  // workgroup barrier that orders LDS traffic only: __syncthreads() would also wait for every
  // outstanding global store and LDS-DMA (vmcnt(0)), i.e. expose a full memory round trip three
  // times per tile
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
  // LDS-DMA of a unit's X tile: LDS is written linearly (wave base + 16 lane); piece p of the
  // tile (row p / 32, slot p % 32) must hold chunk slot ^ (row & 15) of its row, so each lane
  // fetches THAT chunk (the swizzle is an involution: source-side permutation, rule 21)
  // ((set, tile) of the previous / current / next unit are carried through the loop: an integer
  //  division by a run-time value costs ~40 instructions, and there were three per tile and wave)
  auto dma_x = [&](int b, int tile, char* dst) {
    const int n0 = tile * P;
#pragma unroll
    for (int i = 0; i < DMA_PER_WAVE; ++i) {
      const int p = (DMA_PER_WAVE * j + i) * 64 + lane;    // this wave: consecutive 1 KiB pieces
      const int row = p >> 5, slot = p & 31;
      const int ch = (slot & ~15) | ((slot ^ row) & 15);
      const int n = n0 + row < a.N ? n0 + row : a.N - 1;   // padding rows: any valid line
      const __bf16* src = reinterpret_cast<const __bf16*>(a.X) + ((int64_t)b * a.N + n) * D + ch * 8;
      // The DMA is issued from inline asm on purpose: with the builtin, hipcc knows an LDS write is
      // pending on the VM counter and puts s_waitcnt vmcnt(0) in front of every later ds_read that
      // might alias it - i.e. the whole memory latency of the tile just requested would be waited
      // for at once.  Hidden from the compiler, the transfer is ordered by the counted vmcnt in the
      // loop (hidden VMEM operations can only make hipcc's own counted waits stricter: the counter
      // retires in order).  M0 = wave-uniform LDS base, written in the statement that uses it.
      const unsigned ldst = __builtin_amdgcn_readfirstlane(
          (unsigned)(uintptr_t)(lptr_t*)(dst + (DMA_PER_WAVE * j + i) * 1024));
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                   "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(src), "s"(ldst) : "memory");
    }
  };
  int cb = u0 / a.tiles_per_set, ct = u0 - cb * a.tiles_per_set;      // current unit
  int pb_ = cb, pt_ = ct;                                             // previous unit
  if (!SMALL && u0 < u1) dma_x(cb, ct, sXb);

  // Per-lane byte offsets into a tile, computed ONCE: the swizzle XORs the 16-byte chunk index with
  // the row, so an address is not "base + constant" across k-steps - left to the compiler, every
  // LDS access re-derived it (measured: 486 VALU instructions per wave and tile against 72 MFMAs).
  //   oB[k]: B-operand row fragment, row r, chunk 4 (s & 3) + g  (+ 256 (s >> 2) + 8192 nb)
  //   oD[t]: accumulator-layout 8 bytes, row r, features 32 j + 16 t + 4 g  (+ 8192 nb)
  //   oC   : coalesced 16-byte piece of row tid / 32, chunk tid % 32   (+ 8192 i)
  int oB[4], oD[2];
#pragma unroll
  for (int k = 0; k < 4; ++k) oB[k] = toff(r, 4 * k + g);
#pragma unroll
  for (int t = 0; t < 2; ++t) oD[t] = toff(r, 4 * j + 2 * t + (g >> 1)) + 8 * (g & 1);
  const int oC = toff(tid >> 5, tid & 31);

  // coalesced stores of a finished tile (Y; training: O, Qp, ReLU mask words) from its LDS tiles
  const int tiles128 = (a.tiles_per_set * P + 127) / 128;
  auto store_tile = [&](int b, int tile, int par) {
    const int n0 = tile * P, nlive = a.N - n0;
    const int64_t rowbase = (int64_t)b * a.N + n0;
#pragma unroll
    for (int i = 0; i < P / 16; ++i) {
      const int c = tid + 512 * i, row = c >> 5, ch = c & 31;
      if (row < nlive) {
        *reinterpret_cast<uint4*>(a.Y + (rowbase + row) * D + ch * 8) =
            *reinterpret_cast<const uint4*>(sYb + par * TILEB + oC + 8192 * i);
        if (a.OS != nullptr)
          *reinterpret_cast<uint4*>(a.OS + (rowbase + row) * D + ch * 8) =
              *reinterpret_cast<const uint4*>(sOb + par * TILEB + oC + 8192 * i);
        if (a.QpS != nullptr)
          *reinterpret_cast<uint4*>(a.QpS + (rowbase + row) * D + ch * 8) =
              *reinterpret_cast<const uint4*>(sQb + par * TILEB + oC + 8192 * i);
      }
    }
    if (a.mask != nullptr && tid < NBK * 128) {
      const int nb = tid >> 7, w = (tid >> 6) & 1;
      const int64_t blk = (int64_t)b * tiles128 * 8 + tile * NBK + nb;
      a.mask[(blk * 2 + w) * 64 + lane] = sMaskb[par * NBK * 128 + (nb * 2 + w) * 64 + lane];
    }
  };

  // (Tried: the two waves of each SIMD in antiphase - waves 0-3 one phase ahead of waves 4-7, half
  //  steps between the barriers, so that one wave's softmax / epilogue VALU run meets the other's
  //  MFMA run.  Same results, 6 % slower in training mode, equal in inference: the waves of a SIMD
  //  already interleave inside a phase, and the split store passes cost more than the overlap won.)
  // VM instructions per thread of store_tile for a full tile (the last tile of a set is ragged when
  // 32 does not divide N: its pass issues fewer, so the wait that would count it drains instead)
  const int n_store = (P / 16) * (1 + (a.OS != nullptr ? 1 : 0) + (a.QpS != nullptr ? 1 : 0)) +
                      ((a.mask != nullptr && j < NBK * 2) ? 1 : 0);
  const bool ragged = a.N % P != 0;
  auto full_tile = [&](int b, int tile) { (void)b; return !ragged || tile != a.tiles_per_set - 1; };
  int n_last = 0;          // store instructions of the store_tile pass issued in the previous iteration
  int cur_b = -1;
  bf16x8 kpa[2], vta[2];
  for (int unit = u0; unit < u1; ++unit) {
    const int par = (unit - u0) & 1;
    char* sXc = sXb + par * TILEB;
    char* sO = sOb + par * TILEB;
    char* sQ = sQb + par * TILEB;
    char* sY = sYb + par * TILEB;
    uint32_t* sMask = sMaskb + par * NBK * 128;
    const int b = cb, tile = ct;
    int nb_ = cb, nt_ = ct + 1;                                       // next unit
    if (nt_ == a.tiles_per_set) { nt_ = 0; ++nb_; }
    const int n0 = tile * P, nlive = a.N - n0;
    const int64_t rowbase = (int64_t)b * a.N + n0;
    (void)tile;
    FWD_STAMP(0);
    // layer 1: this tile's points straight to registers (issued before the deferred stores, so
    // that the counted wait for them does not cover the stores)
    float xv[SMALL ? NBK : 1][4];
    if (SMALL) {
#pragma unroll
      for (int nb = 0; nb < NBK; ++nb) {
        const int n = 16 * nb + r;
#pragma unroll
        for (int c = 0; c < 4; ++c)
          xv[SMALL ? nb : 0][c] = (n < nlive && c < a.dq)
                      ? reinterpret_cast<const float*>(a.X)[(rowbase + n) * a.dq + c] : 0.f;
      }
    }
    // The next tile starts to stream in now (the last readers of its buffer passed barrier B1 of
    // the previous iteration); this tile's DMA was issued one iteration ago: everything older
    // than the pieces just issued - it and the stores of two tiles back - must have landed.
    if (!SMALL) {
      if (unit + 1 < u1) {
        dma_x(nb_, nt_, sXb + (par ^ 1) * TILEB);
        static_assert(DMA_PER_WAVE == 2, "the counted wait below");
        // Younger than this tile's DMA are the 2 pieces just issued and the stores of tile unit - 2
        // (issued one iteration ago, after its barrier B0): counted, so that they may stay in flight.
        // (Measured: no difference to draining them - they are acknowledged within an iteration.  The
        //  12 % "stall" the cycle stamps of scripts/experiments/fwd_stamps.py showed here was the
        //  stamps' own global stores being waited for.)
        wait_vm(DMA_PER_WAVE + n_last);
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    FWD_STAMP(1);
    lds_barrier();                       // B0: X tile (all waves' pieces); previous tile's Y / mask
    FWD_STAMP(2);
    n_last = 0;
    if (unit > u0) {
      store_tile(pb_, pt_, par ^ 1);
      n_last = full_tile(pb_, pt_) ? n_store : 0;          // (ragged: count unknown -> drained next time)
    }
    pb_ = cb; pt_ = ct;
    cb = nb_; ct = nt_;
    FWD_STAMP(3);
    if (b != cur_b) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        kpa[kt] = *reinterpret_cast<const bf16x8*>(a.KpP + ((int64_t)b * MI + 16 * kt + r) * D +
                                                   32 * j + 8 * g);
        vta[kt] = *reinterpret_cast<const bf16x8*>(a.Vt + ((int64_t)b * D + 32 * j + 16 * kt + r) * MI +
                                                   8 * g);
      }
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
          const bf16x8 bx =
              *reinterpret_cast<const bf16x8*>(sXc + oB[s & 3] + 256 * (s >> 2) + 8192 * nb);
          acc[0][nb] = mfma32(wqa[SMALL ? 0 : s][0], bx, acc[0][nb]);
          acc[1][nb] = mfma32(wqa[SMALL ? 0 : s][1], bx, acc[1][nb]);
        }
    }
    FWD_STAMP(4);
    if (a.QpS != nullptr) {              // own 32-column slice of the Qp tile
#pragma unroll
      for (int nb = 0; nb < NBK; ++nb)
#pragma unroll
        for (int t = 0; t < 2; ++t)
          *reinterpret_cast<bf16x4*>(sQ + oD[t] + 8192 * nb) = pack4(acc[t][nb]);
    }
    // ---- attention over the 32 inducing keys, all inside the wave ----
#pragma unroll
    for (int nb = 0; nb < NBK; ++nb) {
      const bf16x8 qb = pack8(acc[0][nb], acc[1][nb]);
      const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
      f32x4 s0 = mfma32(kpa[0], qb, z4), s1 = mfma32(kpa[1], qb, z4);
      float mx = max_nn(max_nn(max_nn(s0[0], s0[1]), max_nn(s0[2], s0[3])),
                       max_nn(max_nn(s1[0], s1[1]), max_nn(s1[2], s1[3])));
      mx = wave16_max_nn(mx);
      float sum = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        // raw v_exp_f32: the arguments are <= 0 and finite, so exp2f's range handling (compare,
        // select, ldexp per element) buys nothing here
        s0[e] = __builtin_amdgcn_exp2f((s0[e] - mx) * a.scale_log2e);
        s1[e] = __builtin_amdgcn_exp2f((s1[e] - mx) * a.scale_log2e);
        sum += s0[e] + s1[e];
      }
      sum = wave16_sum(sum);
      const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
      for (int e = 0; e < 4; ++e) { s0[e] *= inv; s1[e] *= inv; }
      const bf16x8 pb = pack8(s0, s1);
      acc[0][nb] = mfma32(vta[0], pb, acc[0][nb]);
      acc[1][nb] = mfma32(vta[1], pb, acc[1][nb]);
    }
    // ---- own slice of the O tile ----
#pragma unroll
    for (int nb = 0; nb < NBK; ++nb)
#pragma unroll
      for (int t = 0; t < 2; ++t)
        *reinterpret_cast<bf16x4*>(sO + oD[t] + 8192 * nb) = pack4(acc[t][nb]);
    FWD_STAMP(5);
    lds_barrier();                       // B1: O (and Qp) tiles complete; X tile consumed
    FWD_STAMP(6);
    // ---- GEMM2: Z_h^T = Wo_h . O^T + bo ; Y_h = O_h + relu(Z_h) ----
    // (the accumulators of O_h are re-used for Z_h: the residual O_h is read back from the
    //  wave's own slice of the O tile - bf16, the rounding the two-launch form had as well)
    {
      const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
      const f32x4 bo0 = F8O ? z4 : bias4(1, 0), bo1 = F8O ? z4 : bias4(1, 1);
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
        const bf16x8 ob =
            *reinterpret_cast<const bf16x8*>(sO + oB[s & 3] + 256 * (s >> 2) + 8192 * nb);
        if (F8O) {
          const f8x8 o8 = bf_to_f8(ob);
          acc[0][nb] = mfma32_f8(woa8[F8O ? s : 0][0], o8, acc[0][nb]);
          acc[1][nb] = mfma32_f8(woa8[F8O ? s : 0][1], o8, acc[1][nb]);
        } else {
          acc[0][nb] = mfma32(woa[F8O ? 0 : s][0], ob, acc[0][nb]);
          acc[1][nb] = mfma32(woa[F8O ? 0 : s][1], ob, acc[1][nb]);
        }
      }
    FWD_STAMP(7);
#pragma unroll
    for (int nb = 0; nb < NBK; ++nb) {
      uint32_t bits = 0u;                // byte (j & 3) of mask word j >> 2: tiles 2 j, 2 j + 1
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const bf16x4 o4 = *reinterpret_cast<const bf16x4*>(sO + oD[t] + 8192 * nb);
        f32x4 y, bo4 = {0.f, 0.f, 0.f, 0.f};
        if (F8O) bo4 = bias4(1, t);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float zz = F8O ? acc[t][nb][e] * inv_o + bo4[e] : acc[t][nb][e];
          y[e] = (float)o4[e] + max_nn(zz, 0.f);
          if (zz > 0.f) bits |= 1u << (4 * t + e);
        }
        *reinterpret_cast<bf16x4*>(sY + oD[t] + 8192 * nb) = pack4(y);
      }
      if (a.mask != nullptr)
        reinterpret_cast<uint8_t*>(sMask)[((nb * 2 + (j >> 2)) * 64 + lane) * 4 + (j & 3)] =
            (uint8_t)bits;
    }
    FWD_STAMP(8);
    // (this tile's Y / O / Qp / mask leave in the next iteration, after its barrier B0)
  }
  if (u0 < u1) {
    lds_barrier();
    store_tile(pb_, pt_, (u1 - 1 - u0) & 1);
  }
}


// =====================================================================================
// Round 3: the same block as a PRODUCER / CONSUMER pair of wave groups in one 1024-thread
// workgroup (16 waves = 4 per SIMD, <= 128 registers each).  In k_isab1_fwd256 above every wave
// walks GEMM1 -> softmax -> GEMM2 -> epilogue between the same two barriers, so the two waves of a
// SIMD want the matrix pipe at the same time and leave it idle at the same time (MFMA utilisation
// 0.33, waves waiting 69 % of their cycles: profiles/r02_fwd256_fused_sq.txt).  Here
//     waves 0-7  (role A, wave = head j): X tile --GEMM1 (Wq_j in 64 registers)--> Qp_j
//                --scores, softmax, A V--> O_j  --own slice of the O tile of unit k
//     waves 8-15 (role B, wave = head j): O tile of unit k-1 --GEMM2 (Wo_j in 64 registers)-->
//                Z_j ; Y_j = O_j + relu(Z_j)  --own slice of the Y tile; coalesced stores of the
//                Y tile of unit k-2 (training: O, Qp tiles of unit k-1, mask words)
// with ONE workgroup barrier per unit.  Each SIMD hosts two A and two B waves, whose VALU-heavy
// (softmax, epilogue) and MFMA-heavy (the two GEMMs) stretches belong to different units and
// overlap by construction; a wave holds one weight slice instead of two, which is what brings it
// under 128 registers.  Tiles, swizzle, LDS-DMA and layouts are those of the kernel above.
#ifdef PCA_FWD_STAMPS
// measurement build: s_memtime stamps (low 32 bits) of workgroup 0 into LDS ([wave 16][iteration 32]
// [stamp 8], copied to g_fwd_stamps at the end) - no global store inside the loop
#define AB_STAMP(i)                                                                               \
  do {                                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    if (blockIdx.x == 0 && k < 32 && ((i) == 0 || stamp_sel == 15 || stamp_sel == (i))) {                                                           \
      const long long t_ = (long long)__builtin_readcyclecounter();                               \
      if (lane == 0) sStamp[(wv * 32 + k) * 8 + (i)] = (unsigned)t_;                                     \
    }                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                            \
  } while (0)
#else
#define AB_STAMP(i)
#endif

// Two 16-byte pieces to (uniform base + 32-bit lane offsets), as instructions in the scalar-base
// form.  (Written as C++, the full-tile and the ragged store paths get merged by hipcc into one
// exec-mask maze with the LDS reads under the row guards and a full wait in front of every store.)
// The s_nop covers the store-data hazard, which hipcc's hazard recognizer cannot see inside inline
// asm: without it the next instruction may overwrite a data register the store has not read yet.
__device__ __forceinline__ void store2x16_s(char* base, unsigned off0, unsigned off1, const uint4& v0,
                                            const uint4& v1) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 d0 = {v0.x, v0.y, v0.z, v0.w}, d1 = {v1.x, v1.y, v1.z, v1.w};
  asm volatile("global_store_dwordx4 %0, %2, %4\n\tglobal_store_dwordx4 %1, %3, %4\n\ts_nop 1"
               :: "v"(off0), "v"(off1), "v"(d0), "v"(d1), "s"(base) : "memory");
}

template <bool SMALL, bool F8O, bool TRAIN, bool ABREAST = false>
__global__ __launch_bounds__(1024) void k_isab1_fwd256_ab(const FusedFwdArgs a) {
  // (TRAIN: fewer fragments ahead - the kernel has to stay inside 128 registers without scratch)
  constexpr int KS = D / 32, PF = TRAIN ? (F8O ? 1 : 2) : 3;
  constexpr int PFB = PF;   // (role B 4-5 fragments ahead: 20 bytes of scratch and 102 us against 91.8)
  // ablation switches of the measurement build (results are garbage with any of them set):
  // 1 no softmax arithmetic, 2 no epilogue arithmetic, 4 no barrier, 8 no LDS-DMA, 16 no Y stores,
  // 32 no GEMM1, 64 no GEMM2, 128 the GEMMs' MFMAs without their B-operand reads (fragments re-used)
#ifdef PCA_FWD_ABLATE
  const int abl = a.ablate & 0xffff;
#else
  constexpr int abl = 0;
#endif
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // X tiles [XR]: one unit ahead.  (A ring of three tiles, two units ahead, measured no different:
  // 91.8 against 91.9 us at B = 128, N = 2048 - the LDS-DMA is not what role A waits for.)
  constexpr int XR = 2, AH = XR - 1;
  char* sXb = smem;                         // X tiles [XR][TILEB]   (role A only)
  char* sOb = smem + XR * TILEB;            // O tiles (A writes slices, B reads)
  char* sYb = sOb + 2 * TILEB;              // Y tiles (B only)
  char* sKb = sYb + 2 * TILEB;              // K slices of the current set, wave-private [8][2][64][16 B]
  char* sVb = sKb + TILEB;                  // V slices likewise (16 registers less in role A)
  constexpr bool VL = true;
  // F8O: the O tile once more as fp8 e4m3 [2][32 points][256 B] (16-byte chunks XOR-swizzled by
  // row): role A converts its slice once, role B's GEMM2 reads fp8 fragments for the K = 128 MFMA
  char* sO8b = sVb + TILEB;
  uint32_t* sMaskb = reinterpret_cast<uint32_t*>(sVb + (F8O ? 2 : 1) * TILEB);
  // bq [256], bo [256], and both again: point block nb reads copy nb, so that hipcc does not keep
  // one read alive (8 registers) across the attention of block 0
  float* sBias = reinterpret_cast<float*>(sMaskb + 2 * NBK * 2 * 64);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int role = wv >> 3, j = wv & 7;     // (waves w, w + 4, w + 8, w + 12 share a SIMD)
#ifdef PCA_FWD_STAMPS
  // (PCA_AB_STAMPSEL = 1 .. 6: only stamp 0 and that one are taken - a stamp costs ~280 cycles,
  //  seven of them per iteration distort what they measure; 15 = all)
  const int stamp_sel = a.ablate >> 16;
  unsigned* sStamp = reinterpret_cast<unsigned*>(sBias + 4 * D);
  for (int i = tid; i < 16 * 32 * 8; i += 1024) sStamp[i] = 0;
#endif
  const int r = lane & 15, g = lane >> 4;

  // ---- this wave's ONE weight slice: A operands [row = feature 32 j + 16 t + r][k = 32 s + 8 g ..],
  // loaded inside the role's branch (role B in F8O mode holds fp8 fragments of the K = 128 MFMA
  // instead: the two never live together)
  typedef int v8i __attribute__((ext_vector_type(8)));
  if (tid < 2 * D) {
    const float v = tid < D ? a.bq[tid] : a.bo[tid - D];
    sBias[tid] = v;
    sBias[2 * D + tid] = v;
  }
  auto bias4 = [&](int which, int t, int copy) {
    const float4 q4 = *reinterpret_cast<const float4*>(sBias + (2 * copy + which) * D + 32 * j +
                                                       16 * t + 4 * g);
    return f32x4{q4.x, q4.y, q4.z, q4.w};
  };
  auto unit_barrier = [&] {                 // LDS traffic only (no vmcnt: see the kernel above)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (!(abl & 4)) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  const int total_units = a.B * a.tiles_per_set;
  const int u0 = blockIdx.x * a.units_per_wg;
  const int u1 = (u0 + a.units_per_wg < total_units) ? u0 + a.units_per_wg : total_units;
  const int n = u1 - u0;
  int oB[4], oD[2];
#pragma unroll
  for (int k = 0; k < 4; ++k) oB[k] = toff(r, 4 * k + g);
#pragma unroll
  for (int t = 0; t < 2; ++t) oD[t] = toff(r, 4 * j + 2 * t + (g >> 1)) + 8 * (g & 1);
  int cb = u0 / a.tiles_per_set, ct = u0 - cb * a.tiles_per_set;

#ifdef PCA_FWD_ABLATE
  {   // static wave priority per role: bits 8-9 role A, bits 10-11 role B
    const int pr = role == 0 ? (abl >> 8) & 3 : (abl >> 10) & 3;
    if (pr == 1) __builtin_amdgcn_s_setprio(1);
    else if (pr == 2) __builtin_amdgcn_s_setprio(2);
    else if (pr == 3) __builtin_amdgcn_s_setprio(3);
  }
#endif
  if (role == 0) {
    // ================================ role A ================================
    // LDS-DMA of a unit's X tile: scalar tile base + a 32-bit lane offset (piece p of the tile,
    // row p / 32, slot p % 32, holds chunk slot ^ (row & 15) of its row)
    // (scalar work per piece kept to a minimum - the kernel's time follows its instruction count,
    //  the tile base is advanced, not recomputed with 64-bit multiplies; LDS
    //  addresses are integers off one base, not pointer casts with their null checks; M0 is read
    //  once before the loop and only restored after a piece)
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t*)smem;
    unsigned m0_keep;
    asm volatile("s_mov_b32 %0, m0" : "=s"(m0_keep));
    int poff[DMA_PER_WAVE], prow[DMA_PER_WAVE];
#pragma unroll
    for (int i = 0; i < DMA_PER_WAVE; ++i) {
      const int p = (DMA_PER_WAVE * j + i) * 64 + lane;
      const int row = p >> 5, slot = p & 31;
      prow[i] = row;
      poff[i] = row * ROWB + (((slot & ~15) | ((slot ^ row) & 15)) << 4);
    }
    auto dma_x = [&](const char* base, int nlive, int slot_x) {       // base: row 0 of the tile
      const unsigned l0 = lds0 + (unsigned)(slot_x * TILEB + DMA_PER_WAVE * j * 1024);
      if (nlive >= P) {                      // (uniform) a full tile: the lane offsets as they are
#pragma unroll
        for (int i = 0; i < DMA_PER_WAVE; ++i)
          asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t"
                       "global_load_lds_dwordx4 %0, %1\n\ts_mov_b32 m0, %3"
                       :: "v"((unsigned)poff[i]), "s"(base), "s"(l0 + i * 1024), "s"(m0_keep) : "memory");
      } else {                               // ragged: padding rows read the tile's last valid line
#pragma unroll
        for (int i = 0; i < DMA_PER_WAVE; ++i) {
          const unsigned off =
              (unsigned)(poff[i] - (prow[i] < nlive ? 0 : (prow[i] - nlive + 1) * ROWB));
          asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t"
                       "global_load_lds_dwordx4 %0, %1\n\ts_mov_b32 m0, %3"
                       :: "v"(off), "s"(base), "s"(l0 + i * 1024), "s"(m0_keep) : "memory");
        }
      }
    };
    auto load_points = [&](int b, int tile, float (&xv)[NBK][4]) {    // layer 1: points to registers
      const int nlive = a.N - tile * P;
      const float* base = reinterpret_cast<const float*>(a.X) + ((int64_t)b * a.N + tile * P) * a.dq;
#pragma unroll
      for (int nb = 0; nb < NBK; ++nb) {
        const int nn = 16 * nb + r < nlive ? 16 * nb + r : nlive - 1;
#pragma unroll
        for (int c = 0; c < 4; ++c) xv[nb][c] = base[nn * a.dq + (c < a.dq ? c : 0)];
      }
    };
    float wqs[SMALL ? 2 : 1][4][4];         // layer 1: fc_q rows of this lane's features
    if (SMALL) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int c = 0; c < 4; ++c)
            wqs[SMALL ? t : 0][e][c] = c < a.dq ? a.WqF[(32 * j + 16 * t + 4 * g + e) * a.dq + c] : 0.f;
    }
    bf16x8 wa[KS][2];
    if (!SMALL) {
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int t = 0; t < 2; ++t)
          wa[s][t] = *reinterpret_cast<const bf16x8*>(a.Wq + (32 * j + 16 * t + r) * D + 32 * s + 8 * g);
    }
    const int qoff = r * ROWB + (32 * j + 4 * g) * 2;      // TRAIN: this lane's 8 bytes of a Qp row
    // F8O: this lane's 4-byte piece of the fp8 O tile: row r, byte 32 j + 16 t + 4 g
    int o8[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) o8[t] = r * D + (((2 * j + t) ^ r) << 4) + 4 * g;
    float xv[NBK][4], xn[NBK][4];
    int fb = cb, ft = ct;                    // the next unit to fetch (AH units ahead of the current)
    // ... and the address of its first row: + 32 rows per unit, less across a ragged set end
    const char* fbase = reinterpret_cast<const char*>(a.X) + ((int64_t)cb * a.N + ct * P) * ROWB;
    const int last_rows = a.N - (a.tiles_per_set - 1) * P;             // rows of a set's last tile
    auto fetch_advance = [&] {
      if (++ft == a.tiles_per_set) { ft = 0; ++fb; fbase += (int64_t)last_rows * ROWB; }
      else fbase += P * ROWB;
    };
    auto fetch_live = [&] { return ft == a.tiles_per_set - 1 ? last_rows : P; };
    if (!SMALL) {
#pragma unroll
      for (int q = 0; q < AH; ++q)
        if (q < n) { dma_x(fbase, fetch_live(), q); fetch_advance(); }
    } else {
      load_points(cb, ct, xv);
      fetch_advance();
    }
    // weights, X tile 0 (hidden from hipcc: waited for explicitly) - as a BUILTIN wait, so that the
    // compiler's own counter bookkeeping knows no load is pending at the loop entry: left to
    // itself it re-waits for the 16 weight loads inside the loop (vmcnt(15) ... vmcnt(0) between
    // the MFMAs of GEMM1), and in steady state those waits hit the LDS-DMA just issued
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
    __syncthreads();                         // biases visible; X tile 0 landed in every wave
    int cur_b = -1;
    bf16x8 vta[2];
    char* sK = sKb + j * 2048 + lane * 16;   // this lane's 16 bytes of kpa[kt] at + 1024 kt
    char* sV = sVb + j * 2048 + lane * 16;
    int xs = 0;                              // slot of the current unit in the X ring
    for (int k = 0; k <= n + 1; ++k) {
      AB_STAMP(0);
      if (k < n) {
        const int par = k & 1;
        int nb_ = cb, nt_ = ct + 1;
        if (nt_ == a.tiles_per_set) { nt_ = 0; ++nb_; }
        const int b = cb;
        char* sXc = sXb + xs * TILEB;
        char* sO = sOb + par * TILEB;
        // TRAIN, d -> d blocks: Qp is saved for the backward - this wave's [32 points][32 features]
        // slice goes straight to memory, 8 bytes per lane (round 2 staged a Qp tile in LDS for
        // role-B-style coalesced rows: two tiles of LDS and a copy pass; the 64-byte row pieces of
        // the eight heads meet in L2 before the line leaves it)
        const int nliveA = a.N - ct * P;
        char* qbase = reinterpret_cast<char*>(a.QpS) + ((int64_t)cb * a.N + ct * P) * ROWB;   // (uniform)
        if (b != cur_b) {                    // this set's K / V slices of head j (once per set)
#pragma unroll
          for (int kt = 0; kt < 2; ++kt) {
            *reinterpret_cast<bf16x8*>(sK + 1024 * kt) = *reinterpret_cast<const bf16x8*>(
                a.KpP + ((int64_t)b * MI + 16 * kt + r) * D + 32 * j + 8 * g);
            const bf16x8 vt = *reinterpret_cast<const bf16x8*>(
                a.Vt + ((int64_t)b * D + 32 * j + 16 * kt + r) * MI + 8 * g);
            if (VL) *reinterpret_cast<bf16x8*>(sV + 1024 * kt) = vt;
            else vta[kt] = vt;
          }
          cur_b = b;
          __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): before any LDS-DMA of this iteration
        }
        const bool more = k + (SMALL ? 1 : AH) < n;
        if (more) {                          // the input of unit k + AH starts to arrive
          if (!SMALL) {
            const int fs = xs + AH >= XR ? xs + AH - XR : xs + AH;
            if (!(abl & 8)) dma_x(fbase, fetch_live(), fs);
          } else {
            load_points(fb, ft, xn);
          }
          fetch_advance();
        }
        AB_STAMP(1);
        if (ABREAST && !SMALL) {
          // ---- both point blocks of the unit ABREAST: every stage is written for block 0 and
          // block 1 side by side, so that the wave always has a second, independent chain to issue
          // from while the first waits on an MFMA result, an LDS read or a cross-lane step (with
          // the blocks one after the other a wave is stalled 80 % of its cycles, and four such
          // waves leave the SIMD idle about half the time)
          constexpr int PA = 2;
          const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
          f32x4 acc[NBK][2];
#pragma unroll
          for (int nb = 0; nb < NBK; ++nb) { acc[nb][0] = bias4(0, 0, nb); acc[nb][1] = bias4(0, 1, nb); }
          bf16x8 bx[PA + 1][NBK];
#pragma unroll
          for (int s = 0; s < PA; ++s)
#pragma unroll
            for (int nb = 0; nb < NBK; ++nb)
              bx[s][nb] = *reinterpret_cast<const bf16x8*>(sXc + oB[s & 3] + 256 * (s >> 2) + 8192 * nb);
#pragma unroll
          for (int s = 0; s < ((abl & 32) ? 1 : KS); ++s) {
            if (s + PA < KS) {
#pragma unroll
              for (int nb = 0; nb < NBK; ++nb)
                bx[(s + PA) % (PA + 1)][nb] = *reinterpret_cast<const bf16x8*>(
                    sXc + oB[(s + PA) & 3] + 256 * ((s + PA) >> 2) + 8192 * nb);
            }
#pragma unroll
            for (int nb = 0; nb < NBK; ++nb) {
              acc[nb][0] = mfma32(wa[s][0], bx[s % (PA + 1)][nb], acc[nb][0]);
              acc[nb][1] = mfma32(wa[s][1], bx[s % (PA + 1)][nb], acc[nb][1]);
            }
          }
          const bf16x8 kp0 = *reinterpret_cast<const bf16x8*>(sK);
          const bf16x8 kp1 = *reinterpret_cast<const bf16x8*>(sK + 1024);
          AB_STAMP(2);
          if (TRAIN && a.QpS != nullptr) {
#pragma unroll
            for (int nb = 0; nb < NBK; ++nb) {
              char* q0 = qbase + (unsigned)(qoff + 16 * ROWB * nb);
              if (nliveA >= P || 16 * nb + r < nliveA) {
                *reinterpret_cast<bf16x4*>(q0) = pack4(acc[nb][0]);
                *reinterpret_cast<bf16x4*>(q0 + 32) = pack4(acc[nb][1]);
              }
            }
          }
          f32x4 s0[NBK], s1[NBK];
#pragma unroll
          for (int nb = 0; nb < NBK; ++nb) {
            const bf16x8 qb = pack8(acc[nb][0], acc[nb][1]);
            s0[nb] = mfma32(kp0, qb, z4);
            s1[nb] = mfma32(kp1, qb, z4);
          }
          if (!(abl & 1)) {
            float mx[NBK];
#pragma unroll
            for (int nb = 0; nb < NBK; ++nb)
              mx[nb] = max_nn(max_nn(max_nn(s0[nb][0], s0[nb][1]), max_nn(s0[nb][2], s0[nb][3])),
                             max_nn(max_nn(s1[nb][0], s1[nb][1]), max_nn(s1[nb][2], s1[nb][3])));
#pragma unroll
            for (int nb = 0; nb < NBK; ++nb) mx[nb] = wave16_max_nn(mx[nb]);
#pragma unroll
            for (int nb = 0; nb < NBK; ++nb) mx[nb] = -mx[nb] * a.scale_log2e;
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
              for (int nb = 0; nb < NBK; ++nb) {
                s0[nb][e] = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[nb][e], a.scale_log2e, mx[nb]));
                s1[nb][e] = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[nb][e], a.scale_log2e, mx[nb]));
              }
          }
          bf16x8 pb[NBK];
#pragma unroll
          for (int nb = 0; nb < NBK; ++nb) pb[nb] = pack8(s0[nb], s1[nb]);
          bf16x8 ones;
#pragma unroll
          for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
          const bf16x8 v0 = *reinterpret_cast<const bf16x8*>(sV);
          const bf16x8 v1 = *reinterpret_cast<const bf16x8*>(sV + 1024);
#pragma unroll
          for (int nb = 0; nb < NBK; ++nb) {
            const f32x4 sm = mfma32(ones, pb[nb], z4);
            const f32x4 o0 = mfma32(v0, pb[nb], z4);
            const f32x4 o1 = mfma32(v1, pb[nb], z4);
            const float inv = (abl & 1) ? sm[0] : __builtin_amdgcn_rcpf(sm[0]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              acc[nb][0][e] = __builtin_fmaf(o0[e], inv, acc[nb][0][e]);
              acc[nb][1][e] = __builtin_fmaf(o1[e], inv, acc[nb][1][e]);
            }
            *reinterpret_cast<bf16x4*>(sO + oD[0] + 8192 * nb) = pack4(acc[nb][0]);
            *reinterpret_cast<bf16x4*>(sO + oD[1] + 8192 * nb) = pack4(acc[nb][1]);
            if (F8O) {
              char* sO8 = sO8b + par * (P * D) + (16 * D) * nb;
              *reinterpret_cast<uint32_t*>(sO8 + o8[0]) =
                  cvt4_f8(acc[nb][0][0], acc[nb][0][1], acc[nb][0][2], acc[nb][0][3]);
              *reinterpret_cast<uint32_t*>(sO8 + o8[1]) =
                  cvt4_f8(acc[nb][1][0], acc[nb][1][1], acc[nb][1][2], acc[nb][1][3]);
            }
          }
          AB_STAMP(5);
        } else
#pragma unroll
        for (int nb = 0; nb < NBK; ++nb) {
          f32x4 acc0 = bias4(0, 0, nb), acc1 = bias4(0, 1, nb);
          bf16x8 kp0, kp1;
          if (SMALL) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              acc0[e] += wqs[0][e][0] * xv[nb][0] + wqs[0][e][1] * xv[nb][1] +
                         wqs[0][e][2] * xv[nb][2] + wqs[0][e][3] * xv[nb][3];
              acc1[e] += wqs[SMALL ? 1 : 0][e][0] * xv[nb][0] + wqs[SMALL ? 1 : 0][e][1] * xv[nb][1] +
                         wqs[SMALL ? 1 : 0][e][2] * xv[nb][2] + wqs[SMALL ? 1 : 0][e][3] * xv[nb][3];
            }
          } else {
            // B-operand fragments PF k-steps ahead of the MFMAs that consume them: a k-step is
            // 32 cycles of MFMA issue, an LDS read returns after ~100-130
            bf16x8 bx[PF + 1];
#pragma unroll
            for (int s = 0; s < PF; ++s)
              bx[s] = *reinterpret_cast<const bf16x8*>(sXc + oB[s & 3] + 256 * (s >> 2) + 8192 * nb);
#pragma unroll
            for (int s = 0; s < ((abl & 32) ? 1 : KS); ++s) {
              if (s + PF < KS && !(abl & 128))
                bx[(s + PF) % (PF + 1)] = *reinterpret_cast<const bf16x8*>(
                    sXc + oB[(s + PF) & 3] + 256 * ((s + PF) >> 2) + 8192 * nb);
              if (s == KS - PF) {          // this head's K slices: requested under the last MFMAs
                kp0 = *reinterpret_cast<const bf16x8*>(sK);
                kp1 = *reinterpret_cast<const bf16x8*>(sK + 1024);
              }
              acc0 = mfma32(wa[s][0], bx[s % (PF + 1)], acc0);
              acc1 = mfma32(wa[s][1], bx[s % (PF + 1)], acc1);
            }
          }
          if (SMALL || (abl & 32)) {
            kp0 = *reinterpret_cast<const bf16x8*>(sK);
            kp1 = *reinterpret_cast<const bf16x8*>(sK + 1024);
          }
          AB_STAMP(2 + 2 * nb);
          if (TRAIN && a.QpS != nullptr) {
            char* q0 = qbase + (unsigned)(qoff + 16 * ROWB * nb);
            if (nliveA >= P) {               // (uniform: a full tile stores unguarded, see the wait below)
              *reinterpret_cast<bf16x4*>(q0) = pack4(acc0);
              *reinterpret_cast<bf16x4*>(q0 + 32) = pack4(acc1);
            } else if (16 * nb + r < nliveA) {
              *reinterpret_cast<bf16x4*>(q0) = pack4(acc0);
              *reinterpret_cast<bf16x4*>(q0 + 32) = pack4(acc1);
            }
          }
          // ---- attention over the 32 inducing keys, inside the wave ----
          const bf16x8 qb = pack8(acc0, acc1);
          const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
          f32x4 s0 = mfma32(kp0, qb, z4);
          f32x4 s1 = mfma32(kp1, qb, z4);
          if (!(abl & 1)) {
            float mx = max_nn(max_nn(max_nn(s0[0], s0[1]), max_nn(s0[2], s0[3])),
                             max_nn(max_nn(s1[0], s1[1]), max_nn(s1[2], s1[3])));
            mx = wave16_max_nn(mx);
            const float mc = -mx * a.scale_log2e;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              s0[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[e], a.scale_log2e, mc));
              s1[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[e], a.scale_log2e, mc));
            }
          }
          // Un-normalised probabilities go to the MFMA; their sum over the 32 keys comes from the
          // matrix pipe as well - an A operand of ones gives every accumulator row the column sum,
          // i.e. every lane the denominator of ITS point (one MFMA instead of 7 dependent adds and
          // two cross-lane butterfly steps; the sum is that of the bf16 values actually multiplied)
          const bf16x8 pb = pack8(s0, s1);
          bf16x8 ones;
#pragma unroll
          for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
          const f32x4 sm = mfma32(ones, pb, z4);
          const f32x4 o0 = mfma32(VL ? *reinterpret_cast<const bf16x8*>(sV) : vta[0], pb, z4);
          const f32x4 o1 = mfma32(VL ? *reinterpret_cast<const bf16x8*>(sV + 1024) : vta[1], pb, z4);
          const float inv = (abl & 1) ? sm[0] : __builtin_amdgcn_rcpf(sm[0]);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc0[e] = __builtin_fmaf(o0[e], inv, acc0[e]);
            acc1[e] = __builtin_fmaf(o1[e], inv, acc1[e]);
          }
          *reinterpret_cast<bf16x4*>(sO + oD[0] + 8192 * nb) = pack4(acc0);
          *reinterpret_cast<bf16x4*>(sO + oD[1] + 8192 * nb) = pack4(acc1);
          if (F8O) {
            char* sO8 = sO8b + par * (P * D) + (16 * D) * nb;
            *reinterpret_cast<uint32_t*>(sO8 + o8[0]) = cvt4_f8(acc0[0], acc0[1], acc0[2], acc0[3]);
            *reinterpret_cast<uint32_t*>(sO8 + o8[1]) = cvt4_f8(acc1[0], acc1[1], acc1[2], acc1[3]);
          }
          AB_STAMP(3 + 2 * nb);
        }
        if (SMALL) {
#pragma unroll
          for (int nb = 0; nb < NBK; ++nb)
#pragma unroll
            for (int c = 0; c < 4; ++c) xv[nb][c] = xn[nb][c];
        }
        cb = nb_; ct = nt_;
        if (++xs == XR) xs = 0;
        // the next unit's X tile (this wave's pieces) has landed before the barrier releases it;
        // the pieces of the units after it, issued later, stay in flight (vmcnt retires in order)
        // (TRAIN: the four Qp stores of a full tile were issued after the LDS-DMA and may stay in
        //  flight - vmcnt retires in issue order; a ragged tile's guarded stores may have been
        //  skipped by whole waves, so its count is unknown: drain)
        if (!SMALL) {
          if (TRAIN && a.QpS != nullptr && nliveA >= P) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
      }
      AB_STAMP(6);
      unit_barrier();
    }
  } else {
    // ================================ role B ================================
    const int tb = tid & 511;                 // thread index inside the role
    const float inv_o = F8O ? a.inv_o[0] : 1.f;
    bf16x8 wa[F8O ? 1 : KS][2];
    v8i wf[F8O ? 2 : 1][2];                   // F8O: [K step of 128][feature tile]: 32 bytes per lane
    int o8r[2];                               // F8O: the lane's two 16-byte chunks of an fp8 O row
#pragma unroll
    for (int c = 0; c < 2; ++c) o8r[c] = r * D + (((2 * g + c) ^ r) << 4);
    if (F8O) {
#pragma unroll
      for (int S = 0; S < 2; ++S)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const uint4* pw = reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(a.Wo) +
                                                           (32 * j + 16 * t + r) * D + 128 * S + 32 * g);
          const uint4 lo = pw[0], hi = pw[1];
          wf[F8O ? S : 0][t] = v8i{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w,
                                   (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
        }
    } else {
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int t = 0; t < 2; ++t)
          wa[F8O ? 0 : s][t] =
              *reinterpret_cast<const bf16x8*>(a.Wo + (32 * j + 16 * t + r) * D + 32 * s + 8 * g);
    }
    const int oC = toff(tb >> 5, tb & 31);
    const int tiles128 = (a.tiles_per_set * P + 127) / 128;
    int b1 = cb, t1 = ct, b2 = cb, t2 = ct;   // units k - 1 (GEMM2) and k - 2 (deferred Y stores)
    // byte offset of row 0 of the current unit (cb, ct) in a [B*N][256] bf16 tensor, advanced with
    // the unit (as role A's fbase: no 64-bit multiplies per unit), and of units k - 1 / k - 2
    const int last_rows = a.N - (a.tiles_per_set - 1) * P;
    int64_t ro0 = ((int64_t)cb * a.N + ct * P) * ROWB, ro1 = ro0, ro2 = ro0;
    __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0): the weight slice (see role A)
    __syncthreads();                          // pairs with role A's
    for (int k = 0; k <= n + 1; ++k) {
      const int par = k & 1;
      AB_STAMP(0);
      // Order inside an iteration: GEMM2 of BOTH point blocks first, then the epilogues, then the
      // memory passes.  Role A's iteration is [input issue, GEMM1(0), softmax(0), GEMM1(1),
      // softmax(1)]; with role B as [stores, GEMM2(0), epilogue(0), GEMM2(1), epilogue(1)] the two
      // roles asked for the matrix pipe in the same stretches and left it idle in the same
      // stretches (ablation: the costs of GEMM1, GEMM2, the stores and the LDS-DMA issue ADDED up,
      // scripts/experiments/ab_ablate.sh).  Now B's 32 MFMAs run while A issues its input and
      // does softmax(0), and B's VALU / memory work runs under A's GEMM1(1).
      AB_STAMP(1);
      if (k >= 1 && k <= n) {                 // unit k - 1
        const int pq = par ^ 1;
        char* sO = sOb + pq * TILEB;
        char* sY = sYb + pq * TILEB;
        uint32_t* sMask = sMaskb + pq * NBK * 128;
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        f32x4 acc[NBK][2];
        bf16x4 o4[NBK][2];
#pragma unroll
        for (int nb = 0; nb < NBK; ++nb) {
          acc[nb][0] = F8O ? z4 : bias4(1, 0, nb);
          acc[nb][1] = F8O ? z4 : bias4(1, 1, nb);
          if (F8O) {
            // GEMM2 on v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3, block scales 2^0): 4 instructions
            // of 32 cycles per point block instead of 16 of 16 cycles, fragments straight from the
            // fp8 O tile (32 bytes per lane and K step: lane (r, g) holds k = 32 g .. 32 g + 31 of
            // its row - verified by scripts/probe/mfma_f8_probe.hip, which also measured the rate)
            const char* sO8 = sO8b + pq * (P * D) + (16 * D) * nb;
            v8i fb8[2];
#pragma unroll
            for (int S = 0; S < 2; ++S) {
              const uint4 lo = *reinterpret_cast<const uint4*>(sO8 + (o8r[0] ^ (S << 7)));
              const uint4 hi = *reinterpret_cast<const uint4*>(sO8 + (o8r[1] ^ (S << 7)));
              fb8[S] = v8i{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w,
                           (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
            }
            if (!TRAIN && nb == NBK - 1) {
#pragma unroll
              for (int q = 0; q < NBK; ++q) {
                o4[q][0] = *reinterpret_cast<const bf16x4*>(sO + oD[0] + 8192 * q);
                o4[q][1] = *reinterpret_cast<const bf16x4*>(sO + oD[1] + 8192 * q);
              }
            }
#pragma unroll
            for (int S = 0; S < 2; ++S)
#pragma unroll
              for (int t = 0; t < 2; ++t)
                acc[nb][t] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
                    wf[F8O ? S : 0][t], fb8[S], acc[nb][t], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
          } else {
          bf16x8 bo_[PFB + 1];
#pragma unroll
          for (int s = 0; s < PFB; ++s)
            bo_[s] = *reinterpret_cast<const bf16x8*>(sO + oB[s & 3] + 256 * (s >> 2) + 8192 * nb);
#pragma unroll
          for (int s = 0; s < ((abl & 64) ? 1 : KS); ++s) {
            if (s + PFB < KS && !(abl & 128))
              bo_[(s + PFB) % (PFB + 1)] = *reinterpret_cast<const bf16x8*>(
                  sO + oB[(s + PFB) & 3] + 256 * ((s + PFB) >> 2) + 8192 * nb);
            // the residual O_j of both blocks: requested under the last MFMAs, not after them
            if (!TRAIN && nb == NBK - 1 && s == KS - PFB) {       // (TRAIN: no registers to spare)
#pragma unroll
              for (int q = 0; q < NBK; ++q) {
                o4[q][0] = *reinterpret_cast<const bf16x4*>(sO + oD[0] + 8192 * q);
                o4[q][1] = *reinterpret_cast<const bf16x4*>(sO + oD[1] + 8192 * q);
              }
            }
            acc[nb][0] = mfma32(wa[F8O ? 0 : s][0], bo_[s % (PFB + 1)], acc[nb][0]);
            acc[nb][1] = mfma32(wa[F8O ? 0 : s][1], bo_[s % (PFB + 1)], acc[nb][1]);
          }
          }
          AB_STAMP(2 + nb);
        }
#pragma unroll
        for (int nb = 0; nb < NBK; ++nb) {
          uint32_t bits = 0u;
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            f32x4 bo4 = z4;
            if (F8O) bo4 = bias4(1, t, nb);
            if (TRAIN) o4[nb][t] = *reinterpret_cast<const bf16x4*>(sO + oD[t] + 8192 * nb);
            f32x4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float zz = F8O ? acc[nb][t][e] * inv_o + bo4[e] : acc[nb][t][e];
              y[e] = (abl & 2) ? zz : (float)o4[nb][t][e] + max_nn(zz, 0.f);
              if (TRAIN && zz > 0.f) bits |= 1u << (4 * t + e);
            }
            *reinterpret_cast<bf16x4*>(sY + oD[t] + 8192 * nb) = pack4(y);
          }
          if (TRAIN)
            reinterpret_cast<uint8_t*>(sMask)[((nb * 2 + (j >> 2)) * 64 + lane) * 4 + (j & 3)] =
                (uint8_t)bits;
          AB_STAMP(4 + nb);
        }
        if (TRAIN) {                          // O and Qp tiles of the unit, saved for the backward
          const int nlive = a.N - t1 * P;
          char* baseO = reinterpret_cast<char*>(a.OS) + ro1;          // (uniform)
          // (LDS reads unconditional, all ahead of the stores: a read under the divergent row guard
          //  would get a basic block and a full wait of its own)
          uint4 vo[P / 16];
#pragma unroll
          for (int i = 0; i < P / 16; ++i) vo[i] = *reinterpret_cast<const uint4*>(sO + oC + 8192 * i);
          if (nlive >= P) {                   // (uniform) a full tile: no row guards
            static_assert(P / 16 == 2, "two pieces per thread");
            store2x16_s(baseO, (unsigned)(tb * 16), (unsigned)((tb + 512) * 16), vo[0], vo[1]);
          } else {
#pragma unroll
            for (int i = 0; i < P / 16; ++i) {
              const int c = tb + 512 * i;
              if ((c >> 5) < nlive) *reinterpret_cast<uint4*>(baseO + (unsigned)(c * 16)) = vo[i];
            }
          }
        }
      }
      if (k >= 2 && !(abl & 16)) {            // coalesced stores of the Y tile of unit k - 2
        const int nlive = a.N - t2 * P;
        char* base = reinterpret_cast<char*>(a.Y) + ro2;               // (uniform)
        uint4 vy[P / 16];
#pragma unroll
        for (int i = 0; i < P / 16; ++i)
          vy[i] = *reinterpret_cast<const uint4*>(sYb + par * TILEB + oC + 8192 * i);
        if (nlive >= P) {                     // (uniform) a full tile: no row guards
          store2x16_s(base, (unsigned)(tb * 16), (unsigned)((tb + 512) * 16), vy[0], vy[1]);
        } else {
#pragma unroll
          for (int i = 0; i < P / 16; ++i) {
            const int c = tb + 512 * i;
            if ((c >> 5) < nlive) *reinterpret_cast<uint4*>(base + (unsigned)(c * 16)) = vy[i];
          }
        }
        if (TRAIN && tb < NBK * 128) {
          const int nb = tb >> 7, w = (tb >> 6) & 1;
          const int64_t blk = (int64_t)b2 * tiles128 * 8 + t2 * NBK + nb;
          a.mask[(blk * 2 + w) * 64 + lane] = sMaskb[par * NBK * 128 + (nb * 2 + w) * 64 + lane];
        }
      }
      AB_STAMP(6);
      b2 = b1; t2 = t1; ro2 = ro1;
      b1 = cb; t1 = ct; ro1 = ro0;
      if (k < n) {
        if (++ct == a.tiles_per_set) { ct = 0; ++cb; ro0 += (int64_t)last_rows * ROWB; }
        else ro0 += P * ROWB;
      }
      unit_barrier();
    }
  }
#ifdef PCA_FWD_STAMPS
  __syncthreads();
  if (blockIdx.x == 0)
    for (int i = tid; i < 16 * 32 * 8; i += 1024) g_fwd_stamps[i] = sStamp[i];
#endif
}

}  // namespace

int isab1_fwd256_fused(const void* X, int dq, const __bf16* WqB, const float* WqF, const float* bq,
                       const __bf16* KpP, const __bf16* Vt, const __bf16* WoB, const float* bo,
                       __bf16* Y, __bf16* QpS, __bf16* OS, uint32_t* mask, int B, int N,
                       hipStream_t st, const float* inv_o) {
  FusedFwdArgs a{};
  a.inv_o = inv_o;
  a.X = X; a.Wq = WqB; a.WqF = WqF; a.bq = bq; a.KpP = KpP; a.Vt = Vt; a.Wo = WoB; a.bo = bo;
  a.Y = Y; a.QpS = QpS; a.OS = OS; a.mask = mask;
  a.B = B; a.N = N; a.dq = dq;
  a.tiles_per_set = (int)cdiv(N, P);
  a.scale_log2e = 1.4426950408889634f / sqrtf((float)D);
#if defined(PCA_FWD_ABLATE) || defined(PCA_FWD_STAMPS)      // diagnostic builds only (scripts/experiments)
  a.ablate = (getenv("PCA_AB_ABLATE") ? atoi(getenv("PCA_AB_ABLATE")) : 0) |
             ((getenv("PCA_AB_STAMPSEL") ? atoi(getenv("PCA_AB_STAMPSEL")) : 15) << 16);
#else
  a.ablate = 15 << 16;
#endif
  const int total = B * a.tiles_per_set;
  int grid = total < 256 ? total : 256;
  a.units_per_wg = (int)cdiv(total, grid);
  grid = (int)cdiv(total, a.units_per_wg);
  const size_t lds = 8 * (size_t)TILEB + 2 * NBK * 2 * 64 * sizeof(uint32_t) + 2 * D * sizeof(float);
  static std::once_flag once;
  std::call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_isab1_fwd256<false, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_isab1_fwd256<true, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_isab1_fwd256<false, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_isab1_fwd256<true, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  const double pts = (double)B * N;
  ProfScope ps(PCA_K_MAB1_FWD, st, 2.0 * pts * ((double)dq * D + (double)D * D + 2.0 * MI * D),
               pts * ((dq <= 4 ? 4.0 : 2.0) * dq + 2.0 * D));
  // PCA_D256_AB=0: the one-role kernel (every wave walks the whole chain; A/B measurements)
  static const bool ab = [] { const char* e = getenv("PCA_D256_AB"); return !(e && e[0] == '0'); }();
  if (ab) {
    const bool train = OS != nullptr;
    PCA_REQUIRE(!train || mask != nullptr, "isab1_fwd256_fused: training needs the mask buffer");
    const size_t lds2 = (inv_o != nullptr ? 9 : 8) * (size_t)TILEB + 2 * NBK * 2 * 64 * sizeof(uint32_t) +
                        4 * D * sizeof(float)
#ifdef PCA_FWD_STAMPS
                        + 16 * 32 * 8 * 4
#endif
        ;
#define PCA_AB_LAUNCH(S, F, T)                                                                   \
  do {                                                                                           \
    static std::once_flag o2;                                                                    \
    std::call_once(o2, [] {                                                                      \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_isab1_fwd256_ab<S, F, T>),       \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);        \
    });                                                                                          \
    hipLaunchKernelGGL((k_isab1_fwd256_ab<S, F, T>), dim3(grid), dim3(1024), lds2, st, a);       \
  } while (0)
    const bool sm = dq <= 4, f8 = inv_o != nullptr;
    // (role A with the two point blocks of a unit side by side: measured, no gain - DESIGN.md 4.5.1; the
    //  kernel variant is kept for the ablation scripts only)
#ifdef PCA_FWD_ABLATE
    static const bool abreast = [] { const char* e = getenv("PCA_AB_ABREAST"); return e && e[0] == '1'; }();
#else
    constexpr bool abreast = false;
#endif
    if (abreast && !sm && !f8 && !train) {
      static std::once_flag o3;
      std::call_once(o3, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_isab1_fwd256_ab<false, false, false, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      });
      hipLaunchKernelGGL((k_isab1_fwd256_ab<false, false, false, true>), dim3(grid), dim3(1024), lds2,
                         st, a);
      ps.end();
      return check_launch("k_isab1_fwd256_ab<abreast>");
    }
    if (sm && f8 && train) PCA_AB_LAUNCH(true, true, true);
    else if (sm && f8) PCA_AB_LAUNCH(true, true, false);
    else if (sm && train) PCA_AB_LAUNCH(true, false, true);
    else if (sm) PCA_AB_LAUNCH(true, false, false);
    else if (f8 && train) PCA_AB_LAUNCH(false, true, true);
    else if (f8) PCA_AB_LAUNCH(false, true, false);
    else if (train) PCA_AB_LAUNCH(false, false, true);
    else PCA_AB_LAUNCH(false, false, false);
#undef PCA_AB_LAUNCH
    ps.end();
    return check_launch("k_isab1_fwd256_ab");
  }
  if (inv_o != nullptr) {
    if (dq <= 4) hipLaunchKernelGGL((k_isab1_fwd256<true, true>), dim3(grid), dim3(512), lds, st, a);
    else hipLaunchKernelGGL((k_isab1_fwd256<false, true>), dim3(grid), dim3(512), lds, st, a);
  } else {
    if (dq <= 4) hipLaunchKernelGGL((k_isab1_fwd256<true, false>), dim3(grid), dim3(512), lds, st, a);
    else hipLaunchKernelGGL((k_isab1_fwd256<false, false>), dim3(grid), dim3(512), lds, st, a);
  }
  ps.end();
  return check_launch("k_isab1_fwd256");
}

#ifdef PCA_FWD_STAMPS
extern "C" int pca_debug_fwd_stamps(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fwd_stamps), sizeof(g_fwd_stamps));
}
#endif

}  // namespace pca
