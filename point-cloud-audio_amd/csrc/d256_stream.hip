// Streaming row-GEMMs of the d = 256 path with the WEIGHTS IN REGISTERS and the activations in a
// deep LDS-DMA ring:
//     PROJ2   Kp = X Wk^T + bk ; Vp = X Wv^T + bv        (fc_k / fc_v of the few-queries block over
//                                                          the N keys, modules.py:21; one pass over X)
//     DX2     dX (+)= dKp Wk + dVp Wv                     (their adjoint w.r.t. the keys; one pass)
//     DX1     dX = dQp Wq                                 (adjoint of fc_q, modules.py:20)
//     DX3     dX = dQp Wq + dKp Wk + dVp Wv               (both of the above for one ISAB input)
// k_rowgemm (d256_bf16.hip) keeps the 128 KiB weight image in LDS, which leaves 32 KiB for
// activations: 32 KiB in flight per CU is 4 TB/s by Little's law, and that is what it measured
// (3.3 TB/s; PROJ and the dKp / dVp adjoint also ran as two launches each, reading X / re-reading
// dX twice).  Here, as in k_isab1_fwd256, wave j of 8 owns output features 32 j .. 32 j + 31 and
// holds its [32 x 256] slice of each weight as MFMA A operands (64 registers per weight); LDS holds
// only [32 points][256] activation tiles: a ring of NBUF tiles per input stream filled by
// global_load_lds_dwordx4 (1 KiB per wave instruction, swizzled at the source), NBUF - 1 tiles
// ahead, and one tile per output that leaves in 16-byte pieces of full rows.
// Roofline unit: 2 d^2 FLOP per point and weight, 2 d bytes per point and tensor.
#include "d256_bf16.hpp"

#include <math.h>

#include <mutex>

namespace pca {

namespace {

constexpr int D = 256, P = 32, KS = D / 32;
constexpr int ROWB = D * 2, TILEB = P * ROWB;               // 16 KiB per tile

struct RowStreamArgs {
  const __bf16* In[3];      // [B*N][256] input streams
  const __bf16* W[3];       // [256][256] bf16 A-operand images: row = output feature, col = k
  const float* bias[2];     // per output (nullable)
  const float* inv_scale;   // F8: 1 / s of the two fp8 weight images (W holds fp8 bytes of s * W)
  __bf16* Out[2];           // [B*N][256]
  const __bf16* Acc;        // ACC: tensor added to output 0 (may alias Out[0])
  int B, N, tiles_per_set, units_per_wg;
};

typedef __attribute__((address_space(3))) void lds_void_t;
__device__ __forceinline__ f32x4 tof(bf16x4 v) {
  return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}

#define PCA_WAIT_VM_CASE(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
__device__ __forceinline__ void wait_vm(int n) {
  switch (n) {
    PCA_WAIT_VM_CASE(1) PCA_WAIT_VM_CASE(2) PCA_WAIT_VM_CASE(3) PCA_WAIT_VM_CASE(4)
    PCA_WAIT_VM_CASE(5) PCA_WAIT_VM_CASE(6) PCA_WAIT_VM_CASE(7) PCA_WAIT_VM_CASE(8)
    PCA_WAIT_VM_CASE(9) PCA_WAIT_VM_CASE(10) PCA_WAIT_VM_CASE(11) PCA_WAIT_VM_CASE(12)
    PCA_WAIT_VM_CASE(13) PCA_WAIT_VM_CASE(14) PCA_WAIT_VM_CASE(15) PCA_WAIT_VM_CASE(16)
    PCA_WAIT_VM_CASE(17) PCA_WAIT_VM_CASE(18) PCA_WAIT_VM_CASE(19) PCA_WAIT_VM_CASE(20)
    PCA_WAIT_VM_CASE(21) PCA_WAIT_VM_CASE(22) PCA_WAIT_VM_CASE(23) PCA_WAIT_VM_CASE(24)
    default:
      if (n > 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}

// NIN input streams, NOUT outputs, NG = max(NIN, NOUT) weights: GEMM g multiplies input
// (NIN == 2 ? g : 0) and lands in output (NOUT == 2 ? g : 0) - the two products of DX2 share one
// accumulator.  ACC: output 0 also gets the tensor a.Acc (a third DMA stream).
// F8 (PROJ2): the weights are fp8 e4m3 images of s * W, the activation fragments are converted to
// fp8 in registers, the accumulators are rescaled by 1 / s before the bias (PCA_MODE_FP8)
template <int NIN, int NOUT, bool ACC, int NBUF, bool F8 = false>
__global__ __launch_bounds__(512, 2) void k_rowstream(const RowStreamArgs a) {
  constexpr int NG = NIN > NOUT ? NIN : NOUT;
  constexpr int NS = NIN + (ACC ? 1 : 0);       // DMA streams
  constexpr int PD = NBUF - 1;                  // tiles ahead
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sIn = smem;                             // [NS][NBUF][TILEB]
  char* sOut = smem + NS * NBUF * TILEB;        // [NOUT][TILEB]
  const int tid = threadIdx.x, lane = tid & 63;
  const int j = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;

  // A operands: [row = feature 32 j + 16 t + r][k = 32 s + 8 g ..]
  bf16x8 wa[F8 ? 1 : NG][KS][2];
  f8x8 wa8[F8 ? NG : 1][KS][2];
  float inv_s[NG];
#pragma unroll
  for (int q = 0; q < NG; ++q) {
    inv_s[q] = F8 ? a.inv_scale[q] : 1.f;
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int64_t o = (int64_t)(32 * j + 16 * t + r) * D + 32 * s + 8 * g;
        if (F8)
          wa8[F8 ? q : 0][s][t] =
              *reinterpret_cast<const f8x8*>(reinterpret_cast<const uint8_t*>(a.W[q]) + o);
        else
          wa[F8 ? 0 : q][s][t] = *reinterpret_cast<const bf16x8*>(a.W[q] + o);
      }
  }
  f32x4 bz[NOUT][2];
#pragma unroll
  for (int o = 0; o < NOUT; ++o)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      bz[o][t] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (a.bias[o] != nullptr) {
        const float4 b4 = *reinterpret_cast<const float4*>(a.bias[o] + 32 * j + 16 * t + 4 * g);
        bz[o][t] = f32x4{b4.x, b4.y, b4.z, b4.w};
      }
    }

  const int total_units = a.B * a.tiles_per_set;
  const int u0 = blockIdx.x * a.units_per_wg;
  const int u1 = (u0 + a.units_per_wg < total_units) ? u0 + a.units_per_wg : total_units;
  const int T = u1 - u0;
  const bool ragged = a.N % P != 0;
  auto full_tile = [&](int k) {
    return !ragged || (u0 + k) % a.tiles_per_set != a.tiles_per_set - 1;
  };
  // per-lane offsets into a tile (k_isab1_fwd256): B-operand fragments, accumulator-layout 8 bytes,
  // coalesced 16-byte piece
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
  // LDS-DMA of tile k of every stream (2 pieces of 1 KiB per wave and stream; issued from inline
  // asm so that hipcc does not serialise it against the ds_reads: see k_isab1_fwd256)
  auto dma = [&](int k) {
    const int unit = u0 + k;
    const int b = unit / a.tiles_per_set, n0 = (unit - b * a.tiles_per_set) * P;
#pragma unroll
    for (int w = 0; w < NS; ++w) {
      const __bf16* base = w < NIN ? a.In[w] : a.Acc;
      char* dst = sIn + (w * NBUF + k % NBUF) * TILEB;
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
#pragma unroll 1
  for (int k = 0; k < PD && k < T; ++k) dma(k);

#pragma unroll 1
  for (int k = 0; k < T; ++k) {
    const int unit = u0 + k;
    const int b = unit / a.tiles_per_set, n0 = (unit - b * a.tiles_per_set) * P;
    const int nlive = a.N - n0;
    // Tile k + PD starts to stream in (its buffer was last read in iteration k - 1).  The DMA of
    // tile k must have landed: younger in the queue are the DMAs of tiles k + 1 .. k + PD and the
    // stores of tiles k - PD .. k - 1 (2 NOUT instructions per thread for a full tile).
    if (k + PD < T) dma(k + PD);
    {
      const int ahead = (T - 1 - k) < PD ? (T - 1 - k) : PD;
      int younger = ahead * 2 * NS;
#pragma unroll
      for (int i = 1; i <= PD; ++i)
        if (k - i >= 0) younger += full_tile(k - i) ? 2 * NOUT : -1000;
      wait_vm(younger);
    }
    lds_barrier();                       // B0: tile k of every stream; the previous outputs are read
    f32x4 acc[NOUT][2][2];               // [output][feature tile][point block]
#pragma unroll
    for (int o = 0; o < NOUT; ++o)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acc[o][t][nb] = F8 ? f32x4{0.f, 0.f, 0.f, 0.f} : bz[o][t];
#pragma unroll
    for (int q = 0; q < NG; ++q) {
      const char* sX = sIn + ((NIN >= 2 ? q : 0) * NBUF + k % NBUF) * TILEB;
      constexpr int dummy = 0;
      (void)dummy;
      const int o = NOUT == 2 ? q : 0;
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          const bf16x8 bx =
              *reinterpret_cast<const bf16x8*>(sX + oB[s & 3] + 256 * (s >> 2) + 8192 * nb);
          if (F8) {
            const f8x8 b8 = bf_to_f8(bx);
            acc[o][0][nb] = mfma32_f8(wa8[F8 ? q : 0][s][0], b8, acc[o][0][nb]);
            acc[o][1][nb] = mfma32_f8(wa8[F8 ? q : 0][s][1], b8, acc[o][1][nb]);
          } else {
            acc[o][0][nb] = mfma32(wa[F8 ? 0 : q][s][0], bx, acc[o][0][nb]);
            acc[o][1][nb] = mfma32(wa[F8 ? 0 : q][s][1], bx, acc[o][1][nb]);
          }
        }
    }
#pragma unroll
    for (int o = 0; o < NOUT; ++o)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          f32x4 v = acc[o][t][nb];
          if (F8) {                      // (PROJ2 only: output o is GEMM o)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] * inv_s[o] + bz[o][t][e];
          }
          if (ACC && o == 0) {
            const f32x4 old = tof(*reinterpret_cast<const bf16x4*>(
                sIn + (NIN * NBUF + k % NBUF) * TILEB + oD[t] + 8192 * nb));
            v[0] += old[0]; v[1] += old[1]; v[2] += old[2]; v[3] += old[3];
          }
          *reinterpret_cast<bf16x4*>(sOut + o * TILEB + oD[t] + 8192 * nb) = pack4(v);
        }
    lds_barrier();                       // B1: output tiles complete; input tiles consumed
#pragma unroll
    for (int o = 0; o < NOUT; ++o)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int c = tid + 512 * i, row = c >> 5, ch = c & 31;
        if (row < nlive)
          *reinterpret_cast<uint4*>(a.Out[o] + ((int64_t)b * a.N + n0 + row) * D + ch * 8) =
              *reinterpret_cast<const uint4*>(sOut + o * TILEB + oC + 8192 * i);
      }
  }
}

template <int NIN, int NOUT, bool ACC, int NBUF, bool F8 = false>
int launch_rowstream(RowStreamArgs a, hipStream_t st) {
  a.tiles_per_set = (int)cdiv(a.N, P);
  const int total = a.B * a.tiles_per_set;
  int grid = total < 256 ? total : 256;
  a.units_per_wg = (int)cdiv(total, grid);
  grid = (int)cdiv(total, a.units_per_wg);
  constexpr int NS = NIN + (ACC ? 1 : 0);
  const size_t lds = (size_t)(NS * NBUF + NOUT) * TILEB;
  static_assert((NS * NBUF + NOUT) * TILEB <= 160 * 1024, "LDS");
  static std::once_flag once;
  std::call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_rowstream<NIN, NOUT, ACC, NBUF, F8>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  hipLaunchKernelGGL((k_rowstream<NIN, NOUT, ACC, NBUF, F8>), dim3(grid), dim3(512), lds, st, a);
  return check_launch("k_rowstream");
}

}  // namespace

// Kp = X Wk^T + bk, Vp = X Wv^T + bv ; WkB / WvB: natural bf16 images ([out][in], prep mode 0)
int rowstream256_proj2(const __bf16* X, const __bf16* WkB, const __bf16* WvB, const float* bk,
                       const float* bv, __bf16* Kp, __bf16* Vp, int B, int N, hipStream_t st) {
  RowStreamArgs a{};
  a.In[0] = X; a.W[0] = WkB; a.W[1] = WvB; a.bias[0] = bk; a.bias[1] = bv;
  a.Out[0] = Kp; a.Out[1] = Vp; a.B = B; a.N = N;
  return launch_rowstream<1, 2, false, 4>(a, st);
}
// the same with fp8 e4m3 operands: Wk8 / Wv8 = natural fp8 images of s * W (prep_weight_f8, mode 0),
// inv_scale[0 / 1] = 1 / s
int rowstream256_proj2_f8(const __bf16* X, const void* Wk8, const void* Wv8, const float* inv_scale,
                          const float* bk, const float* bv, __bf16* Kp, __bf16* Vp, int B, int N,
                          hipStream_t st) {
  RowStreamArgs a{};
  a.In[0] = X; a.W[0] = reinterpret_cast<const __bf16*>(Wk8);
  a.W[1] = reinterpret_cast<const __bf16*>(Wv8); a.bias[0] = bk; a.bias[1] = bv;
  a.inv_scale = inv_scale; a.Out[0] = Kp; a.Out[1] = Vp; a.B = B; a.N = N;
  return launch_rowstream<1, 2, false, 4, true>(a, st);
}
// dX (+)= dKp Wk + dVp Wv ; WkT / WvT: transposed bf16 images ([in][out], prep mode 3)
int rowstream256_dx2(const __bf16* dKp, const __bf16* dVp, const __bf16* WkT, const __bf16* WvT,
                     __bf16* dX, int B, int N, int accumulate, hipStream_t st) {
  RowStreamArgs a{};
  a.In[0] = dKp; a.In[1] = dVp; a.W[0] = WkT; a.W[1] = WvT;
  a.Out[0] = dX; a.Acc = dX; a.B = B; a.N = N;
  if (accumulate) return launch_rowstream<2, 1, true, 2>(a, st);
  return launch_rowstream<2, 1, false, 3>(a, st);
}
// dX = dQp Wq + dKp Wk + dVp Wv in one pass (the ISAB input's whole gradient: mab1's fc_q adjoint
// and the few-queries block's key / value adjoints; three weights = 192 registers)
int rowstream256_dx3(const __bf16* dQp, const __bf16* dKp, const __bf16* dVp, const __bf16* WqT,
                     const __bf16* WkT, const __bf16* WvT, __bf16* dX, int B, int N, hipStream_t st) {
  RowStreamArgs a{};
  a.In[0] = dQp; a.In[1] = dKp; a.In[2] = dVp; a.W[0] = WqT; a.W[1] = WkT; a.W[2] = WvT;
  a.Out[0] = dX; a.B = B; a.N = N;
  return launch_rowstream<3, 1, false, 2>(a, st);
}
// dX = dQp Wq ; WqT: transposed bf16 image
int rowstream256_dx1(const __bf16* dQp, const __bf16* WqT, __bf16* dX, int B, int N,
                     hipStream_t st) {
  RowStreamArgs a{};
  a.In[0] = dQp; a.W[0] = WqT; a.Out[0] = dX; a.B = B; a.N = N;
  return launch_rowstream<1, 1, false, 4>(a, st);
}

}  // namespace pca
