// Generic batched / strided GEMM with bf16 MFMA operands and fp32 accumulation: what
// PCA_MODE_BF16 uses for the blocks that have no fused kernel (d = 256 / m = 32 of BASELINE
// configs[3], the shipped d = 64 / 8-head models).  Same descriptor, strides, batching, split-K
// and epilogue as k_gemm_f32 (gemm_f32.hip); the inner product is v_mfma_f32_16x16x32_bf16.
//
// A 256-thread workgroup computes one tile of C, its four waves 2 x 2; the tile shape follows the
// problem (128x128 in general, 256x32 / 256x64 / 32x256 / 64x256 / 32x32 for the skinny
// per-head products of the attention, see tile_variant()).  fp32 operands are rounded to bf16
// while they are staged: As[m][k] and Bs[n][k] (k contiguous, 80-byte rows: 16-byte fragment
// reads, rows spread over the LDS banks).  Staging is vectorised along whichever index is
// contiguous in memory and the next K tile is prefetched into registers under the MFMAs.
#include "pca_common.h"

#include <stdint.h>
#include "mfma_common.hpp"

namespace pca {

namespace {
constexpr int BK = 32, PITCH = 80;   // bytes per LDS row (32 bf16 + pad)

// Stage an [R rows][32 k] operand tile into LDS as bf16, rows = the non-contracted index.
//   elem(i, k) = P[i * s_row + k * s_k]; rows >= n_rows and k >= k_left read as zero.
// KVEC (0): s_k == 1   -> float4 along k, one 8-byte LDS store
// RVEC (1): s_row == 1 -> float4 along the rows; R >= 128: 4x4 register transpose and 8-byte
//                         LDS stores, R < 128: R/32 float4 per thread, 2-byte stores
// else (2) scalar.  VEC needs 16-byte aligned addresses (checked by the host).
template <int MODE, int R>
struct Stage {
  // floats per thread
  static constexpr int NV = R / 8;

  static __device__ __forceinline__ void load(const float* __restrict__ P, int64_t s_row,
                                              int64_t s_k, int64_t n_rows, int64_t k_left,
                                              int tid, bool vec, float (&v)[NV]) {
    if constexpr (MODE == 0) {              // R x 8 float4; thread -> R/32 of them
#pragma unroll
      for (int e = 0; e < R / 32; ++e) {
        const int idx = tid + e * 256;
        const int i = idx >> 3, k4 = (idx & 7) * 4;
        const float* src = P + i * s_row + k4;
        if (vec && i < n_rows && k4 + 3 < k_left) {
          const float4 t = *reinterpret_cast<const float4*>(src);
          v[4 * e] = t.x; v[4 * e + 1] = t.y; v[4 * e + 2] = t.z; v[4 * e + 3] = t.w;
        } else {
#pragma unroll
          for (int u = 0; u < 4; ++u)
            v[4 * e + u] = (i < n_rows && k4 + u < k_left) ? src[u] : 0.f;
        }
      }
    } else if constexpr (MODE == 1 && R >= 128) {   // 4 rows x 4 k blocks, float4 along the rows
#pragma unroll
      for (int e = 0; e < R / 128; ++e) {
        const int bi = tid + e * 256;      // 8 row groups x 8 k groups x R/32: LDS banks spread
        const int i4 = ((bi >> 6) * 8 + (bi & 7)) * 4, k4 = ((bi >> 3) & 7) * 4;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float* src = P + i4 + (k4 + u) * s_k;
          if (vec && k4 + u < k_left && i4 + 3 < n_rows) {
            const float4 t = *reinterpret_cast<const float4*>(src);
            v[16 * e + 4 * u] = t.x; v[16 * e + 4 * u + 1] = t.y;
            v[16 * e + 4 * u + 2] = t.z; v[16 * e + 4 * u + 3] = t.w;
          } else {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
              v[16 * e + 4 * u + rr] = (k4 + u < k_left && i4 + rr < n_rows) ? src[rr] : 0.f;
          }
        }
      }
    } else if constexpr (MODE == 1) {       // R < 128: R/4 float4 along the rows x 32 k
#pragma unroll
      for (int e = 0; e < R / 32; ++e) {
        const int idx = tid + e * 256;
        const int i4 = (idx % (R / 4)) * 4, k = idx / (R / 4);
        const float* src = P + i4 + k * s_k;
        if (vec && k < k_left && i4 + 3 < n_rows) {
          const float4 t = *reinterpret_cast<const float4*>(src);
          v[4 * e] = t.x; v[4 * e + 1] = t.y; v[4 * e + 2] = t.z; v[4 * e + 3] = t.w;
        } else {
#pragma unroll
          for (int rr = 0; rr < 4; ++rr)
            v[4 * e + rr] = (k < k_left && i4 + rr < n_rows) ? src[rr] : 0.f;
        }
      }
    } else {
#pragma unroll
      for (int e = 0; e < R / 8; ++e) {
        const int idx = tid + e * 256;
        const int i = idx >> 5, k = idx & 31;
        v[e] = (i < n_rows && k < k_left) ? P[i * s_row + k * s_k] : 0.f;
      }
    }
  }

  static __device__ __forceinline__ void store(char* S, int tid, const float (&v)[NV]) {
    if constexpr (MODE == 0) {
#pragma unroll
      for (int e = 0; e < R / 32; ++e) {
        const int idx = tid + e * 256;
        const int i = idx >> 3, k4 = (idx & 7) * 4;
        bf16x4 h;
        h[0] = (__bf16)v[4 * e]; h[1] = (__bf16)v[4 * e + 1];
        h[2] = (__bf16)v[4 * e + 2]; h[3] = (__bf16)v[4 * e + 3];
        *reinterpret_cast<bf16x4*>(S + i * PITCH + k4 * 2) = h;
      }
    } else if constexpr (MODE == 1 && R >= 128) {   // transposed in registers, k contiguous
#pragma unroll
      for (int e = 0; e < R / 128; ++e) {
        const int bi = tid + e * 256;      // 8 row groups x 8 k groups x R/32: LDS banks spread
        const int i4 = ((bi >> 6) * 8 + (bi & 7)) * 4, k4 = ((bi >> 3) & 7) * 4;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          bf16x4 h;
          h[0] = (__bf16)v[16 * e + rr]; h[1] = (__bf16)v[16 * e + 4 + rr];
          h[2] = (__bf16)v[16 * e + 8 + rr]; h[3] = (__bf16)v[16 * e + 12 + rr];
          *reinterpret_cast<bf16x4*>(S + (i4 + rr) * PITCH + k4 * 2) = h;
        }
      }
    } else if constexpr (MODE == 1) {
#pragma unroll
      for (int e = 0; e < R / 32; ++e) {
        const int idx = tid + e * 256;
        const int i4 = (idx % (R / 4)) * 4, k = idx / (R / 4);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
          *reinterpret_cast<__bf16*>(S + (i4 + rr) * PITCH + k * 2) = (__bf16)v[4 * e + rr];
      }
    } else {
#pragma unroll
      for (int e = 0; e < R / 8; ++e) {
        const int idx = tid + e * 256;
        const int i = idx >> 5, k = idx & 31;
        *reinterpret_cast<__bf16*>(S + i * PITCH + k * 2) = (__bf16)v[e];
      }
    }
  }
};

// Workgroup tile (32 WM) x (32 WN): the four waves sit 2 x 2, each owns WM x WN MFMA tiles.
//   (4,4) 128x128 general | (8,1) 256x32 and (8,2) 256x64: few output columns (attention of the
//   "many queries" block: N = m or dh) | (1,8), (2,8): few output rows | (1,1): both small,
//   long K (the per-head products that reduce over the points)
// The MFMA runs transposed (B fragment as the first operand), so a lane ends up with four
// consecutive columns of one row of C: 16-byte stores when C allows them (vec_c).
// HL: both operands enter as hi + lo bf16 pairs (hi = bf16(x), lo = bf16(x - hi)) and a K step is
// three MFMAs (hi hi, hi lo, lo hi): 16 significant bits per operand, i.e. fp32-level results at
// the launch cost of the bf16 product - for the [B m]-row epilogue GEMMs whose ReLU mask must not
// depend on operand rounding (d256_host.hip).
template <int MA, int MB, int WM, int WN, bool HL = false>
__global__ __launch_bounds__(256) void k_gemm_bf16(pca_gemm_desc g, const float* __restrict__ A,
                                                    const float* __restrict__ B,
                                                    const float* __restrict__ bias,
                                                    float* __restrict__ C, int split_k,
                                                    int64_t kchunk, int vec_c, int remap,
                                                    int tiles_m, int tiles_n, int nz,
                                                    int vec_ab) {
  constexpr int BM = 32 * WM, BN = 32 * WN;
  using SA = Stage<MA, BM>;
  using SB = Stage<MB, BN>;
  __shared__ __attribute__((aligned(16))) char smem[(BM + BN) * PITCH * (HL ? 2 : 1)];
  char* const As = smem;
  char* const Bs = smem + BM * PITCH;
  char* const AsL = smem + (BM + BN) * PITCH;          // HL only: the lo halves
  char* const BsL = AsL + BM * PITCH;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, gq = lane >> 4;
  // Workgroup -> (m tile, n tile, batch/split z).  Workgroups go to the eight XCDs round-robin
  // by their linear id, each XCD with an L2 of its own; the remapped orders put the tiles that
  // read the same slab of the streamed operand eight ids apart, i.e. on one XCD back to back:
  //   1: all n tiles of an m tile (tall A, few column tiles: the [B N, d] x [d, d] products)
  //   2: all tiles of one K split (weight gradients: both operands streamed along K)
  int bm = blockIdx.x, bn = blockIdx.y, bz = blockIdx.z;
  if (remap == 1) {
    const int grp = blockIdx.x / (8 * tiles_n), rem = blockIdx.x % (8 * tiles_n);
    bn = rem >> 3;
    bm = grp * 8 + (rem & 7);
    if (bm >= tiles_m) return;
  } else if (remap == 2) {
    const int T = tiles_m * tiles_n;
    const int grp = blockIdx.x / (8 * T), rem = blockIdx.x % (8 * T);
    const int t = rem >> 3;
    bz = grp * 8 + (rem & 7);
    if (bz >= nz) return;
    bm = t % tiles_m;
    bn = t / tiles_m;
  }
  const int zb = bz / split_k;
  const int ks = bz % split_k;
  const int z1 = zb / g.nb2, z2 = zb % g.nb2;
  A += z1 * g.sa_b1 + z2 * g.sa_b2;
  B += z1 * g.sb_b1 + z2 * g.sb_b2;
  C += z1 * g.sc_b1 + z2 * g.sc_b2;

  const int64_t m0 = (int64_t)bm * BM;
  const int64_t n0 = (int64_t)bn * BN;
  const int64_t k_begin = (int64_t)ks * kchunk;
  const int64_t k_end = (k_begin + kchunk < g.K) ? (k_begin + kchunk) : g.K;
  const int wm = 16 * WM * (wave >> 1), wn = 16 * WN * (wave & 1);
  const float* Ab = A + m0 * g.sa_m;
  const float* Bb = B + n0 * g.sb_n;

  f32x4 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  float av[SA::NV], bv[SB::NV];
  const bool va = vec_ab & 1, vb = vec_ab & 2;     // 16-byte loads allowed (alignment)
  SA::load(Ab + k_begin * g.sa_k, g.sa_m, g.sa_k, g.M - m0, k_end - k_begin, tid, va, av);
  SB::load(Bb + k_begin * g.sb_k, g.sb_n, g.sb_k, g.N - n0, k_end - k_begin, tid, vb, bv);
  for (int64_t kt = k_begin; kt < k_end; kt += BK) {
    __syncthreads();                     // previous tile consumed
    SA::store(As, tid, av);
    SB::store(Bs, tid, bv);
    if constexpr (HL) {
#pragma unroll
      for (int e = 0; e < SA::NV; ++e) av[e] -= (float)(__bf16)av[e];
#pragma unroll
      for (int e = 0; e < SB::NV; ++e) bv[e] -= (float)(__bf16)bv[e];
      SA::store(AsL, tid, av);
      SB::store(BsL, tid, bv);
    }
    __syncthreads();
    if (kt + BK < k_end) {               // next tile's loads fly under the MFMAs
      SA::load(Ab + (kt + BK) * g.sa_k, g.sa_m, g.sa_k, g.M - m0, k_end - kt - BK, tid, va, av);
      SB::load(Bb + (kt + BK) * g.sb_k, g.sb_n, g.sb_k, g.N - n0, k_end - kt - BK, tid, vb, bv);
    }
    bf16x8 af[WM];
#pragma unroll
    for (int i = 0; i < WM; ++i)
      af[i] = *reinterpret_cast<const bf16x8*>(As + (wm + 16 * i + r) * PITCH + 16 * gq);
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const bf16x8 bf = *reinterpret_cast<const bf16x8*>(Bs + (wn + 16 * j + r) * PITCH + 16 * gq);
#pragma unroll
      for (int i = 0; i < WM; ++i) acc[i][j] = mfma32(bf, af[i], acc[i][j]);
      if constexpr (HL) {
        const bf16x8 bl = *reinterpret_cast<const bf16x8*>(BsL + (wn + 16 * j + r) * PITCH + 16 * gq);
#pragma unroll
        for (int i = 0; i < WM; ++i) {
          const bf16x8 al = *reinterpret_cast<const bf16x8*>(AsL + (wm + 16 * i + r) * PITCH + 16 * gq);
          acc[i][j] = mfma32(bl, af[i], acc[i][j]);
          acc[i][j] = mfma32(bf, al, acc[i][j]);
        }
      }
    }
  }

  // transposed product: acc[i][j][e] = C[row = 16 i + r][col = 16 j + 4 gq + e]
  if constexpr (WM == 4 && WN == 4) {
    // 128x128 tile: each wave turns its 16 x 64 strips through LDS (its own 16 x 68 floats of
    // the staging buffers) so that one store instruction covers four rows of 256 contiguous
    // bytes instead of sixteen rows of 64
    constexpr int SP = 68;
    __syncthreads();                     // every wave is done with As / Bs
    float* scr = reinterpret_cast<float*>(smem) + wave * (16 * SP);
    const int c4 = (lane & 15) * 4, rq = lane >> 4;
    const int64_t col = n0 + wn + c4;
    float bia[4] = {0.f, 0.f, 0.f, 0.f};
    if (bias != nullptr && ks == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (col + e < g.N) bia[e] = bias[col + e];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<f32x4*>(scr + r * SP + 16 * j + 4 * gq) = acc[i][j];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int rl = rq + 4 * q;
        const f32x4 t = *reinterpret_cast<const f32x4*>(scr + rl * SP + c4);
        const int64_t row = m0 + wm + 16 * i + rl;
        if (row >= g.M || col >= g.N) continue;
        float* dst = C + row * g.sc_m + col;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = g.alpha * t[e] + bia[e];
        if (split_k > 1) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (col + e < g.N) atomicAdd(dst + e, v[e]);
        } else if (vec_c && col + 3 < g.N) {
          float4 o = float4{v[0], v[1], v[2], v[3]};
          if (g.accumulate) {
            const float4 c = *reinterpret_cast<const float4*>(dst);
            o.x += c.x; o.y += c.y; o.z += c.z; o.w += c.w;
          }
          *reinterpret_cast<float4*>(dst) = o;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (col + e < g.N) {
              if (g.accumulate) dst[e] += v[e];
              else dst[e] = v[e];
            }
        }
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < WM; ++i) {
    const int64_t row = m0 + wm + 16 * i + r;
    if (row >= g.M) continue;
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int64_t col = n0 + wn + 16 * j + 4 * gq;
      if (col >= g.N) continue;
      float* dst = C + row * g.sc_m + col;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = g.alpha * acc[i][j][e];
        if (bias != nullptr && ks == 0 && col + e < g.N) v[e] += bias[col + e];
      }
      if (split_k > 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (col + e < g.N) atomicAdd(dst + e, v[e]);
      } else if (vec_c && col + 3 < g.N) {
        float4 o = float4{v[0], v[1], v[2], v[3]};
        if (g.accumulate) {
          const float4 c = *reinterpret_cast<const float4*>(dst);
          o.x += c.x; o.y += c.y; o.z += c.z; o.w += c.w;
        }
        *reinterpret_cast<float4*>(dst) = o;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (col + e < g.N) {
            if (g.accumulate) dst[e] += v[e];
            else dst[e] = v[e];
          }
      }
    }
  }
}

// staging mode of an operand: 0 along k, 1 along the rows, 2 neither stride is 1;
// *vec: the 16-byte loads of modes 0 / 1 are aligned
inline int stage_mode(const float* p, int64_t s_row, int64_t s_k, int64_t b1, int64_t b2,
                      bool* vec) {
  const bool al = (reinterpret_cast<uintptr_t>(p) & 15) == 0 && b1 % 4 == 0 && b2 % 4 == 0;
  *vec = false;
  if (s_k == 1) { *vec = al && s_row % 4 == 0; return 0; }
  if (s_row == 1) { *vec = al && s_k % 4 == 0; return 1; }
  return 2;
}

// tile shape for an (M, N) problem: index into {(4,4), (8,1), (8,2), (1,8), (2,8), (1,1)}
inline int tile_variant(int64_t M, int64_t N, int64_t K, int64_t nbatch) {
  // one unbatched product with a short contraction whose 128 x 128 tiling leaves most CUs idle
  // (the [B*m, d] x [d, d] projections of the per-set epilogues: 64 workgroups at configs[3]):
  // 32 x 32 tiles - 16x the workgroups, operands re-read from L2 - measured 19.5 / 15.7 -> 10.0 / 9.2 us
  if (nbatch == 1 && K <= 1024 && cdiv(M, 128) * cdiv(N, 128) < 128 && M * N >= 256 * 1024)
    return 5;
  if (M <= 32 && N <= 32) return 5;
  if (N <= 32 && M >= 256) return 1;
  if (N <= 64 && M >= 256) return 2;
  if (M <= 32 && N >= 256) return 3;
  if (M <= 64 && N >= 256) return 4;
  return 0;
}
constexpr int kTileM[6] = {128, 256, 256, 32, 64, 32};
constexpr int kTileN[6] = {128, 32, 64, 256, 256, 32};

template <int WM, int WN>
void launch_vec(int ma, int mb, dim3 grid, hipStream_t st, const pca_gemm_desc& g, const float* A,
                const float* B, const float* bias, float* C, int split, int64_t kchunk, int vc,
                int remap, int tm, int tn, int nz, int vab) {
#define PCA_GEMM_CASE(X, Y)                                                                   \
  if (ma == X && mb == Y)                                                                     \
    hipLaunchKernelGGL((k_gemm_bf16<X, Y, WM, WN>), grid, dim3(256), 0, st, g, A, B, bias, C, \
                       split, kchunk, vc, remap, tm, tn, nz, vab)
  PCA_GEMM_CASE(0, 0); PCA_GEMM_CASE(0, 1); PCA_GEMM_CASE(1, 0); PCA_GEMM_CASE(1, 1);
#undef PCA_GEMM_CASE
}
}  // namespace

static int gemm_bf16_impl(const pca_gemm_desc& gin, const float* A, const float* B,
                          const float* bias, float* C, hipStream_t st, bool hl) {
  pca_gemm_desc g = gin;
  PCA_REQUIRE(A && B && C, "gemm_bf16: null operand");
  PCA_REQUIRE(g.M >= 0 && g.N >= 0 && g.K >= 0, "gemm_bf16: negative extent");
  if (g.nb1 <= 0) g.nb1 = 1;
  if (g.nb2 <= 0) g.nb2 = 1;
  if (g.M == 0 || g.N == 0) return PCA_OK;
  // K chunks start at multiples of 32 floats, so the vector paths stay aligned
  bool va = false, vb = false;
  const int ma = stage_mode(A, g.sa_m, g.sa_k, g.sa_b1, g.sa_b2, &va);
  const int mb = stage_mode(B, g.sb_n, g.sb_k, g.sb_b1, g.sb_b2, &vb);
  const int vab = (va ? 1 : 0) | (vb ? 2 : 0);
  // the narrow tiles exist for the staging modes 0 / 1 only
  if (hl && (ma == 2 || mb == 2)) return gemm_f32(gin, A, B, bias, C, st);   // exact anyway
  // (hi + lo operands: 32 x 32 tiles only - the problems it serves are [B m] x 256 x 256)
  const int tv = hl ? 5 : (ma == 2 || mb == 2) ? 0 : tile_variant(g.M, g.N, g.K, (int64_t)g.nb1 * g.nb2);
  const int64_t tiles_m = cdiv(g.M, kTileM[tv]), tiles_n = cdiv(g.N, kTileN[tv]);
  const int64_t nbatch = (int64_t)g.nb1 * g.nb2;
  int split = g.split_k;
  if (split <= 0) {          // same policy as gemm_f32: only an initialised C can take atomics
    split = 1;
    const int64_t wgs = tiles_m * tiles_n * nbatch;
    if (g.accumulate && wgs < 256 && g.K >= 256) {
      int64_t want = 512 / wgs;
      int64_t maxs = g.K / 64;
      split = (int)(want < maxs ? want : maxs);
      if (split < 1) split = 1;
    }
  }
  int64_t ksteps = cdiv(g.K > 0 ? g.K : 1, BK);
  if (split > ksteps) split = (int)ksteps;
  const int64_t kchunk = cdiv(ksteps, split) * BK;
  split = (int)cdiv(g.K > 0 ? g.K : 1, kchunk);
  PCA_REQUIRE(tiles_n <= 65535 && nbatch * split <= 65535,
              "gemm_bf16: grid too large (N tiles %lld, batch*split %lld)",
              (long long)tiles_n, (long long)(nbatch * split));
  dim3 grid((unsigned)tiles_m, (unsigned)tiles_n, (unsigned)(nbatch * split));
  // XCD-aware orders (see the kernel): sharers of a streamed slab on one L2, back to back
  const int64_t nz = nbatch * split, T = tiles_m * tiles_n;
  int remap = 0;
  if (tiles_n >= 2 && tiles_n <= 8 && tiles_m >= 64 && cdiv(tiles_m, 8) * 8 * tiles_n < (1ll << 31)) {
    remap = 1;
    grid = dim3((unsigned)(cdiv(tiles_m, 8) * 8 * tiles_n), 1, (unsigned)nz);
  } else if (T >= 2 && T <= 8 && nz >= 16 && kchunk >= 1024) {
    remap = 2;
    grid = dim3((unsigned)(cdiv(nz, 8) * 8 * T), 1, 1);
  }
  const int tm = (int)tiles_m, tn = (int)tiles_n;
  const int vc = (reinterpret_cast<uintptr_t>(C) & 15) == 0 && g.sc_m % 4 == 0 &&
                 g.sc_b1 % 4 == 0 && g.sc_b2 % 4 == 0;
  if (hl) {
#define PCA_GEMM_CASE(X, Y)                                                                   \
  if (ma == X && mb == Y)                                                                     \
    hipLaunchKernelGGL((k_gemm_bf16<X, Y, 1, 1, true>), grid, dim3(256), 0, st, g, A, B, bias, \
                       C, split, kchunk, vc, remap, tm, tn, (int)nz, vab)
    PCA_GEMM_CASE(0, 0); PCA_GEMM_CASE(0, 1); PCA_GEMM_CASE(1, 0); PCA_GEMM_CASE(1, 1);
#undef PCA_GEMM_CASE
    return check_launch("k_gemm_bf16<hl>");
  }
  switch (tv) {
    case 1: launch_vec<8, 1>(ma, mb, grid, st, g, A, B, bias, C, split, kchunk, vc, remap, tm, tn, (int)nz, vab); break;
    case 2: launch_vec<8, 2>(ma, mb, grid, st, g, A, B, bias, C, split, kchunk, vc, remap, tm, tn, (int)nz, vab); break;
    case 3: launch_vec<1, 8>(ma, mb, grid, st, g, A, B, bias, C, split, kchunk, vc, remap, tm, tn, (int)nz, vab); break;
    case 4: launch_vec<2, 8>(ma, mb, grid, st, g, A, B, bias, C, split, kchunk, vc, remap, tm, tn, (int)nz, vab); break;
    case 5: launch_vec<1, 1>(ma, mb, grid, st, g, A, B, bias, C, split, kchunk, vc, remap, tm, tn, (int)nz, vab); break;
    default: {
#define PCA_GEMM_CASE(X, Y)                                                                   \
  if (ma == X && mb == Y)                                                                     \
    hipLaunchKernelGGL((k_gemm_bf16<X, Y, 4, 4>), grid, dim3(256), 0, st, g, A, B, bias, C,   \
                       split, kchunk, vc, remap, tm, tn, (int)nz, vab)
      PCA_GEMM_CASE(0, 0); PCA_GEMM_CASE(0, 1); PCA_GEMM_CASE(0, 2);
      PCA_GEMM_CASE(1, 0); PCA_GEMM_CASE(1, 1); PCA_GEMM_CASE(1, 2);
      PCA_GEMM_CASE(2, 0); PCA_GEMM_CASE(2, 1); PCA_GEMM_CASE(2, 2);
#undef PCA_GEMM_CASE
    }
  }
  return check_launch("k_gemm_bf16");
}

int gemm_bf16(const pca_gemm_desc& g, const float* A, const float* B, const float* bias,
              float* C, hipStream_t st) {
  return gemm_bf16_impl(g, A, B, bias, C, st, false);
}
int gemm_bf16_hl(const pca_gemm_desc& g, const float* A, const float* B, const float* bias,
                 float* C, hipStream_t st) {
  return gemm_bf16_impl(g, A, B, bias, C, st, true);
}

}  // namespace pca

extern "C" int pca_gemm_bf16(const pca_gemm_desc* g, const float* A, const float* B,
                             const float* bias, float* C, void* stream) {
  PCA_REQUIRE(g != nullptr, "pca_gemm_bf16: null descriptor");
  return pca::gemm_bf16(*g, A, B, bias, C, pca::as_stream(stream));
}
