// The 64 -> 64 (and <= 4 -> 64) Linear layers of the shipped d = 64 models over a tall activation
// (set_transformer-master/modules.py:13-16,20-21,31; Code/models.py:34-44 with dim_hidden = 64), forward
// and input gradient, for the bf16-operand GEMM chain (mab_f32.hip in PCA_MODE_BF16):
//     Y[M][64]  = X[M][64] W^T + b                       (fc_q / fc_k / fc_v / fc_o)
//     dX[M][64] (+)= dY[M][64] W                         (their input gradients)
//     Z = O W^T + b ; Y = O + relu(Z)                    (fc_o with the block's epilogue, modules.py:31)
// The generic k_gemm_bf16 stages both operands through LDS tile by tile: 33 us per call at
// M = 131 200 for 67 MB of traffic.  Here the weight matrix lives in registers as MFMA operands for the
// whole launch and a wave streams 16 rows at a time: lane (row, kg) loads X[row][16 s + 4 kg .. + 3] for
// the four k-steps s (float4 each: the four lanes of a row read 64 contiguous bytes per instruction; the
// next 16 rows are requested under the current products), rounds them to bf16 (RNE, as k_gemm_bf16 does)
// and uses them as the B operand of v_mfma_f32_16x16x16_bf16 - the transposed product Y^T = W X^T - so
// that the accumulator of tile nt is Y[row][16 nt + 4 kg .. + 3]: the same places the lane read, stores are
// float4, and the residual of the fc_o epilogue is already in its registers.  (Each lane reading its own
// 64 contiguous bytes - the contraction index permuted - measured 35 us per call: 16-byte pieces.)
#include "pca_common.h"

#include <stdint.h>
#include "mab1_bf16.hpp"
#include "mfma_common.hpp"

namespace pca {

namespace {

struct Lin64Args {
  const float* X;     // [M][64]
  const float* W;     // [64][64]: nn.Linear layout [out][in]
  const float* b;     // [64] or nullptr
  float* Y;           // [M][64]
  float* Z;           // EPI = 1: the pre-activation [M][64] (out) ; EPI = 2: dZ = dY . [Zin > 0] (out)
  const float* Zin;   // EPI = 2: the forward's pre-activation
  int64_t M;
  int wt;             // 0: Y = X W^T (forward) ; 1: Y = X W (input gradient)
  int accumulate;     // Y += (EPI = 0 only)
};

constexpr int LW = 8;              // waves per workgroup
constexpr int WS = 68;             // floats per weight row in LDS (64 + 4: the 16 rows of an operand read
                                   // spread over the banks)

// EPI 0: Y (+)= X op(W) + b ; EPI 1: Z = X W^T + b, Y = X + relu(Z) ;
// EPI 2 (the adjoint of EPI 1, X = dY): dZ = dY . [Zin > 0] stored, Y = dY + dZ W
template <int EPI>
__global__ __launch_bounds__(64 * LW) void k_lin64(const Lin64Args a) {
  // The weights reach the waves through LDS, read from memory ONCE per workgroup: every wave fetching its
  // own operands made 4096 waves ask the same 128 cache lines of the same 16 L2 channels while the
  // activation stream keeps evicting them from L1 (35 us per call; 1024 waves 14 us).
  __shared__ float sW[64 * WS + 64];     // sW[f][k] = the weight of output f, input k ; then the bias
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, kg = lane >> 4;
  for (int i = tid; i < 64 * 16; i += 64 * LW) {
    const int row = i >> 4, c4 = (i & 15) * 4;
    const float4 w = *reinterpret_cast<const float4*>(a.W + row * 64 + c4);
    if (a.wt) {                          // W[k = row][f = c4 ..]
      sW[(c4 + 0) * WS + row] = w.x; sW[(c4 + 1) * WS + row] = w.y;
      sW[(c4 + 2) * WS + row] = w.z; sW[(c4 + 3) * WS + row] = w.w;
    } else {                             // W[f = row][k = c4 ..]
      *reinterpret_cast<float4*>(sW + row * WS + c4) = w;
    }
  }
  if (tid < 64) sW[64 * WS + tid] = a.b != nullptr ? a.b[tid] : 0.f;
  __syncthreads();
  // A operands: tile nt, k-step s: row m' = lane & 15 <-> output feature 16 nt + m', k slots e <-> input
  // feature 16 s + 4 kg + e
  bf16x4 aw[4][4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float4 w = *reinterpret_cast<const float4*>(sW + (16 * nt + r) * WS + 16 * s + 4 * kg);
      const f32x4 wv = {w.x, w.y, w.z, w.w};
      aw[nt][s] = __builtin_convertvector(wv, bf16x4);
    }
  f32x4 bias[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const float4 bb = *reinterpret_cast<const float4*>(sW + 64 * WS + 16 * nt + 4 * kg);
    bias[nt] = f32x4{bb.x, bb.y, bb.z, bb.w};
  }

  // two 16-row groups per trip, the next two requested under the current products: 16 KB in flight per
  // wave
  const int64_t groups = (a.M + 15) / 16;
  const int64_t nw = (int64_t)gridDim.x * LW;
  int64_t g = (int64_t)blockIdx.x * LW + wave;
  constexpr int NZ = EPI == 2 ? 4 : 1;
  auto rows_of = [&](int64_t gg, float4 (&x)[4], float4 (&z)[NZ]) {
    const int64_t row = gg * 16 + r;
    const int64_t rc = row < a.M ? row : a.M - 1;      // clamped, unconditional; rows past the end unused
    const float* p = a.X + rc * 64 + 4 * kg;
#pragma unroll
    for (int s = 0; s < 4; ++s) x[s] = *reinterpret_cast<const float4*>(p + 16 * s);
    if constexpr (EPI == 2) {
      const float* q = a.Zin + rc * 64 + 4 * kg;
#pragma unroll
      for (int s = 0; s < 4; ++s) z[s] = *reinterpret_cast<const float4*>(q + 16 * s);
    }
  };
  float4 xn[2][4], zn[2][NZ];
  rows_of(g, xn[0], zn[0]);
  rows_of(g + nw, xn[1], zn[1]);
  for (; g < groups; g += 2 * nw) {
    float4 x[2][4], z[2][NZ];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int s = 0; s < 4; ++s) x[u][s] = xn[u][s];
#pragma unroll
      for (int s = 0; s < NZ; ++s) z[u][s] = zn[u][s];
    }
    rows_of(g + 2 * nw, xn[0], zn[0]);
    rows_of(g + 3 * nw, xn[1], zn[1]);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      f32x4 acc[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[nt] = bias[nt];
      f32x4 xm[4];                                     // the product's operand rows
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        xm[s] = f32x4{x[u][s].x, x[u][s].y, x[u][s].z, x[u][s].w};
        if constexpr (EPI == 2) {
          const float4 zz = z[u][s];
          xm[s] = f32x4{zz.x > 0.f ? xm[s][0] : 0.f, zz.y > 0.f ? xm[s][1] : 0.f,
                        zz.z > 0.f ? xm[s][2] : 0.f, zz.w > 0.f ? xm[s][3] : 0.f};
        }
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const f32x4 xv = xm[s];
        const bf16x4 xb = __builtin_convertvector(xv, bf16x4);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = mfma16(aw[nt][s], xb, acc[nt]);
      }
      const int64_t row = (g + u * nw) * 16 + r;
      if (row < a.M) {
        float* yp = a.Y + row * 64 + 4 * kg;
        if constexpr (EPI == 1) {
          float* zp = a.Z + row * 64 + 4 * kg;
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
            *reinterpret_cast<float4*>(zp + 16 * nt) =
                float4{acc[nt][0], acc[nt][1], acc[nt][2], acc[nt][3]};
            *reinterpret_cast<float4*>(yp + 16 * nt) =
                float4{x[u][nt].x + fmaxf(acc[nt][0], 0.f), x[u][nt].y + fmaxf(acc[nt][1], 0.f),
                       x[u][nt].z + fmaxf(acc[nt][2], 0.f), x[u][nt].w + fmaxf(acc[nt][3], 0.f)};
          }
        } else if constexpr (EPI == 2) {
          float* zp = a.Z + row * 64 + 4 * kg;
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
            *reinterpret_cast<float4*>(zp + 16 * nt) = float4{xm[nt][0], xm[nt][1], xm[nt][2], xm[nt][3]};
            *reinterpret_cast<float4*>(yp + 16 * nt) =
                float4{x[u][nt].x + acc[nt][0], x[u][nt].y + acc[nt][1], x[u][nt].z + acc[nt][2],
                       x[u][nt].w + acc[nt][3]};
          }
        } else {
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
            float4 o = {acc[nt][0], acc[nt][1], acc[nt][2], acc[nt][3]};
            if (a.accumulate) {
              const float4 y = *reinterpret_cast<const float4*>(yp + 16 * nt);
              o.x += y.x; o.y += y.y; o.z += y.z; o.w += y.w;
            }
            *reinterpret_cast<float4*>(yp + 16 * nt) = o;
          }
        }
      }
    }
  }
}

// Narrow input (layer 1: DQ <= 4 coordinates per point): Y[M][64] = X[M][DQ] W^T + b on the vector ALU in
// fp32; thread -> 4 output features (its 4 x DQ weights in registers) of every 16th-of-a-wavefront row
template <int DQ>
__global__ __launch_bounds__(256) void k_lin64_narrow(const float* __restrict__ X,
                                                      const float* __restrict__ W,
                                                      const float* __restrict__ b, float* __restrict__ Y,
                                                      int64_t M) {
  const int c4 = threadIdx.x & 15;
  float w[4][DQ], bb[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    bb[c] = b != nullptr ? b[4 * c4 + c] : 0.f;
#pragma unroll
    for (int q = 0; q < DQ; ++q) w[c][q] = W[(4 * c4 + c) * DQ + q];
  }
  const int64_t stride = (int64_t)gridDim.x * 16;
  for (int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4); row < M; row += stride) {
    float x[DQ];
#pragma unroll
    for (int q = 0; q < DQ; ++q) x[q] = X[row * DQ + q];
    float o[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      o[c] = bb[c];
#pragma unroll
      for (int q = 0; q < DQ; ++q) o[c] = fmaf(x[q], w[c][q], o[c]);
    }
    *reinterpret_cast<float4*>(Y + row * 64 + 4 * c4) = float4{o[0], o[1], o[2], o[3]};
  }
}

inline bool al16p(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline unsigned lin64_grid(int64_t M) {
  int64_t wgs = cdiv(cdiv(M, 16), LW);                  // a group per wave before anyone gets a second
  if (wgs > 256) wgs = 256;                             // one workgroup per CU: 512 measured 15 % slower
  return (unsigned)(wgs < 1 ? 1 : wgs);
}

}  // namespace

// shapes these kernels serve (anything else stays on the generic GEMM)
bool lin64_ok(const float* X, const float* Y, int64_t M, int din, int dout) {
  if (dout != 64 || M < 1 || !al16p(Y)) return false;
  return (din == 64 && al16p(X)) || (din >= 1 && din <= 4);
}

// Y[M][64] = X[M][din] W^T + b (din = 64, or 1 .. 4 with accumulate = 0)
int lin64_fwd(const float* X, const float* W, const float* b, float* Y, int64_t M, int din,
              hipStream_t st) {
  PCA_REQUIRE(X && W && Y && M > 0, "lin64_fwd: bad arguments");
  if (din <= 4) {
    int64_t wgs = cdiv(M, 16 * 8);
    if (wgs > 2048) wgs = 2048;
#define PCA_NARROW(Q)                                                                              \
  case Q:                                                                                          \
    hipLaunchKernelGGL((k_lin64_narrow<Q>), dim3((unsigned)wgs), dim3(256), 0, st, X, W, b, Y, M); \
    break;
    switch (din) {
      PCA_NARROW(1) PCA_NARROW(2) PCA_NARROW(3) PCA_NARROW(4)
      default: PCA_REQUIRE(false, "lin64_fwd: din = %d", din);
    }
#undef PCA_NARROW
    return check_launch("k_lin64_narrow");
  }
  PCA_REQUIRE(din == 64, "lin64_fwd: din = %d", din);
  Lin64Args a{X, W, b, Y, nullptr, nullptr, M, 0, 0};
  hipLaunchKernelGGL(k_lin64<0>, dim3(lin64_grid(M)), dim3(64 * LW), 0, st, a);
  return check_launch("k_lin64");
}

// dX[M][64] (+)= dY[M][64] W
int lin64_dx(const float* dY, const float* W, float* dX, int64_t M, int accumulate, hipStream_t st) {
  PCA_REQUIRE(dY && W && dX && M > 0, "lin64_dx: bad arguments");
  Lin64Args a{dY, W, nullptr, dX, nullptr, nullptr, M, 1, accumulate};
  hipLaunchKernelGGL(k_lin64<0>, dim3(lin64_grid(M)), dim3(64 * LW), 0, st, a);
  return check_launch("k_lin64");
}

// Z = O W^T + b ; Y = O + relu(Z)
int lin64_fc_o(const float* O, const float* W, const float* b, float* Z, float* Y, int64_t M,
               hipStream_t st) {
  PCA_REQUIRE(O && W && Z && Y && M > 0, "lin64_fc_o: bad arguments");
  Lin64Args a{O, W, b, Y, Z, nullptr, M, 0, 0};
  hipLaunchKernelGGL(k_lin64<1>, dim3(lin64_grid(M)), dim3(64 * LW), 0, st, a);
  return check_launch("k_lin64");
}

// the adjoint of lin64_fc_o up to the weight gradient: dZ = dY . [Z > 0] ; dO = dY + dZ W
int lin64_fc_o_bwd(const float* dY, const float* Z, const float* W, float* dZ, float* dO, int64_t M,
                   hipStream_t st) {
  PCA_REQUIRE(dY && Z && W && dZ && dO && M > 0, "lin64_fc_o_bwd: bad arguments");
  Lin64Args a{dY, W, nullptr, dO, dZ, Z, M, 1, 0};
  hipLaunchKernelGGL(k_lin64<2>, dim3(lin64_grid(M)), dim3(64 * LW), 0, st, a);
  return check_launch("k_lin64");
}

}  // namespace pca
