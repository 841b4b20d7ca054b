// Shared device helpers of the fused bf16 MFMA kernels (gfx950).
//
// Layout convention of the "transposed chain" (verified by scripts/probe/mfma_probe.hip):
// every intermediate lives in accumulator tiles T[t][nb] (f32x4) that hold a 16x16 block of
// X^T: row = feature 16t + 4g + e (e = register index), column = point 16nb + (lane & 15),
// g = lane >> 4.  With v_mfma_f32_16x16x32_bf16 (A[row = lane&15][k = 8g+j],
// B[k = 8g+j][col = lane&15], D[row = 4g+e][col = lane&15]) two such tiles (features
// 32s .. 32s+31) convert in registers into the B operand of the NEXT product that sums over
// features; the k-slot (g, j) then means feature 32s + perm32(8g + j), so the A operand
// (always a weight-like matrix) is stored with its K axis permuted the same way.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pca {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef float f32x2 __attribute__((ext_vector_type(2)));   // packed-fp32 operand pair (v_pk_*_f32)

// position p = 8g + j inside a 32-wide K block -> index of the element that k-slot (g, j)
// carries when the B operand is built from two accumulator tiles
__host__ __device__ __forceinline__ int perm32(int p) {
  const int g = p >> 3, j = p & 7;
  return j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4);
}

__device__ __forceinline__ f32x4 mfma32(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(bf16x4 a, bf16x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a),
                                                   __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
}

__device__ __forceinline__ bf16x8 pack8(f32x4 lo, f32x4 hi) {
  bf16x8 r;
  r[0] = (__bf16)lo[0]; r[1] = (__bf16)lo[1]; r[2] = (__bf16)lo[2]; r[3] = (__bf16)lo[3];
  r[4] = (__bf16)hi[0]; r[5] = (__bf16)hi[1]; r[6] = (__bf16)hi[2]; r[7] = (__bf16)hi[3];
  return r;
}
__device__ __forceinline__ bf16x4 pack4(f32x4 v) {
  bf16x4 r;
  r[0] = (__bf16)v[0]; r[1] = (__bf16)v[1]; r[2] = (__bf16)v[2]; r[3] = (__bf16)v[3];
  return r;
}

// ---- fp8 (OCP e4m3) operands: same 16x16x32 shape and (g, j) k-slot structure as bf16, 8 values
// per lane in two VGPRs (element j = byte j) --------------------------------------------------
typedef long f8x8;
__device__ __forceinline__ f32x4 mfma32_f8(f8x8 a, f8x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float clamp_f8(float x) { return fminf(fmaxf(x, -448.f), 448.f); }
__device__ __forceinline__ uint32_t cvt4_f8(float a, float b, float c, float d) {
  int p = __builtin_amdgcn_cvt_pk_fp8_f32(clamp_f8(a), clamp_f8(b), 0, false);
  p = __builtin_amdgcn_cvt_pk_fp8_f32(clamp_f8(c), clamp_f8(d), p, true);
  return (uint32_t)p;
}
__device__ __forceinline__ f8x8 pack8_f8(f32x4 lo, f32x4 hi) {
  const uint64_t l = cvt4_f8(lo[0], lo[1], lo[2], lo[3]), h = cvt4_f8(hi[0], hi[1], hi[2], hi[3]);
  return (f8x8)(l | (h << 32));
}
__device__ __forceinline__ f8x8 bf_to_f8(bf16x8 v) {
  const uint64_t l = cvt4_f8((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
  const uint64_t h = cvt4_f8((float)v[4], (float)v[5], (float)v[6], (float)v[7]);
  return (f8x8)(l | (h << 32));
}
// byte offset of 8-byte slot `slot` (8 fp8) of row `row` in a [.][D] fp8 LDS image: the XOR makes
// the 32 (row, g) pairs of a half-wave's ds_read_b64 hit 32 different slots
template <int D>
__device__ __forceinline__ int f8off(int row, int slot) {
  return row * D + ((slot ^ (D == 256 ? ((2 * row) & 31) : (row & 15))) << 3);
}

// byte offset of 16-byte chunk c16 of row `row` in a row-major bf16 LDS image with
// `row_bytes` per row, XOR-swizzled so that 16 lanes reading the same chunk of 16 different
// rows hit 16 different bank groups (cdna_hip_programming.md T2)
__device__ __forceinline__ int swz(int row, int c16, int row_bytes) {
  return row * row_bytes + (((c16 & ~15) | ((c16 ^ row) & 15)) << 4);
}

// Reductions over the 4 lane groups g (lanes l, l^16, l^32, l^48), result in every lane.
// gfx950's v_permlane32_swap / v_permlane16_swap exchange half-waves / odd-even rows in the
// VALU (no trip through the LDS crossbar as ds_bpermute / __shfl_xor would take):
//   permlane32_swap(v, v) -> ([lo, lo], [hi, hi]);  permlane16_swap(v, v) -> ([r0,r0,r2,r2],
//   [r1,r1,r3,r3]); combining the two results of each is the xor-32 / xor-16 butterfly step.
// max as ONE instruction (v_med3_f32 with +inf); fmaxf costs a canonicalising v_max x, x in front of every
// operand the compiler cannot prove quiet.  For operands that are never NaN (scores of finite inputs).
__device__ __forceinline__ float max_nn(float x, float y) { return __builtin_amdgcn_fmed3f(x, y, INFINITY); }
__device__ __forceinline__ float wave16_max_nn(float v) {
  auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = max_nn(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return max_nn(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float wave16_max(float v) {
  auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float wave16_sum(float v) {
  auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// 8 consecutive features of row `n` of a [rows][DK] activation tensor as bf16 (ABF: stored bf16,
// else fp32 rounded here); rows at or past `n_hi` read row n_hi - 1 and come back as zeros.
// UNCONDITIONAL loads on purpose: a load under a divergent `if` gets its own basic block and an
// s_waitcnt vmcnt(0) of its own, i.e. the 4-8 loads of a tile become as many serial round trips.
template <bool ABF>
__device__ __forceinline__ bf16x8 ld_x8_guard(const void* X, int64_t set_row0, int n, int n_hi,
                                              int DK, int ch) {
  const bool ok = n < n_hi;
  const int64_t row = set_row0 + (ok ? n : n_hi - 1);
  bf16x8 v;
  if (ABF) {
    v = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(X) + row * DK + ch * 8);
  } else {
    const float4* src =
        reinterpret_cast<const float4*>(reinterpret_cast<const float*>(X) + row * DK + ch * 8);
    const float4 lo = src[0], hi = src[1];
    v[0] = (__bf16)lo.x; v[1] = (__bf16)lo.y; v[2] = (__bf16)lo.z; v[3] = (__bf16)lo.w;
    v[4] = (__bf16)hi.x; v[5] = (__bf16)hi.y; v[6] = (__bf16)hi.z; v[7] = (__bf16)hi.w;
  }
  if (!ok) {
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = (__bf16)0.f;
  }
  return v;
}

// The same load WITHOUT the zeroing (ABF only): for a register prefetch the zeroing has to wait until
// the registers are consumed - applied right behind the load it makes hipcc wait for the load there,
// and the tile that was meant to arrive during the current tile's arithmetic is waited for before it.
__device__ __forceinline__ bf16x8 ld_x8_clamped(const void* X, int64_t set_row0, int n, int n_hi,
                                                int DK, int ch) {
  // (an empty range - n_hi == 0, a zero-length set - reads row 0 of the set, never row -1)
  const int last = n_hi > 0 ? n_hi - 1 : 0;
  const int64_t row = set_row0 + (n < n_hi ? n : last);
  return *reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(X) + row * DK + ch * 8);
}
__device__ __forceinline__ bf16x8 zero_unless(bool ok, bf16x8 v) {
  if (!ok) {
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = (__bf16)0.f;
  }
  return v;
}

// A chunk of cn <= 1024 points with dk (<= 4) fp32 components each, contiguous in memory, into LDS as
// [pair of points][component][2] (the operand pairs of the packed-fp32 loops of k_mab0_attn_small /
// k_mab0_bwd_small), unused components and the odd point zero - in two halves, so that the loads of
// the NEXT chunk are in flight while the current one is worked on:
//   fetch_points   the raw cn*dk floats into 16 registers per thread (coalesced dwords, all issued
//                  together: one float per loop trip, each waited for on its own, had made the
//                  staging - not the arithmetic - the cost of those kernels)
//   commit_points  registers -> LDS.  Every thread of the 256-thread workgroup must call it (two
//                  barriers inside: the first also ends the previous chunk's reads of sX)
constexpr int PCA_POINT_CHUNK = 1024;
__device__ __forceinline__ void fetch_points(const float* __restrict__ X, int cn, int dk,
                                             float (&v)[16]) {
  const int n = cn * dk;
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int e = threadIdx.x + 256 * u;
    v[u] = X[e < n ? e : n - 1];
  }
}
__device__ __forceinline__ void commit_points(const float (&v)[16], int cn, int dk, float* sX) {
  const int tid = threadIdx.x;
  const int npairs = (cn + 1) >> 1, n = cn * dk;
  __syncthreads();
  if (dk < 4 || (cn & 1))
    for (int i = tid; i < npairs * 8; i += 256) sX[i] = 0.f;
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int e = tid + 256 * u;
    if (e < n) {
      const int pt = e / dk, c = e - pt * dk;
      sX[(pt >> 1) * 8 + c * 2 + (pt & 1)] = v[u];
    }
  }
  __syncthreads();
}

}  // namespace pca
