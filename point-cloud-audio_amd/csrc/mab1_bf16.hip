// Fused bf16-MFMA forward of the "many queries, few keys" MAB -- ISAB's mab1(X, H)
// (set_transformer-master/modules.py:53 with modules.py:19-33 inside): per point
//     Qp = fc_q(x) ; per head: A = softmax(Qp_h Kp_h^T / sqrt(d)) over the m inducing keys ;
//     O = Qp + A Vp ; Y = O + relu(fc_o(O))
// in ONE kernel.  Each wave owns 32 points and runs the whole chain in registers in the
// transposed layout of mfma_common.hpp: GEMM1 (Wq . X^T) -> scores (Kp_h . Qp_h^T) -> softmax
// (4 registers + 2 cross-lane steps per point) -> A.V accumulated ONTO the Qp accumulators
// (the residual is the accumulator's initial value) -> GEMM2 (Wo . O^T) -> ReLU/residual
// epilogue.  Nothing but the X tile, the weights and the set's Kp/Vp ever touches LDS, and
// the N x m attention matrix never exists in memory.  bf16 MFMA operands, fp32 accumulate,
// fp32 softmax / bias / residual.
//
// Roofline unit: per point 2*(2 d^2 + 2 m d) FLOP forward.
#include "mab1_bf16.hpp"
#include "d256_bf16.hpp"

#include <math.h>

#include <mutex>
#include <type_traits>

namespace pca {

namespace {

constexpr int TP = M1_TP;
constexpr int NB = M1_NB;

// ---------------------------------------------------------------------------------
// weight preparation (once per call; d^2 elements): fp32 nn.Linear weights -> bf16 images
//   natural : dst[n][k]            = src[n][k]
//   permK   : dst[n][32s + p]      = src[n][32s + perm32(p)]
//   transposed+permK: dst[i][32s+p]= src[32s + perm32(p)][i]
// ---------------------------------------------------------------------------------
__global__ void k_prep_weight(const float* __restrict__ src, __bf16* __restrict__ dst, int rows,
                              int cols, int mode) {
  // src is [rows][cols] fp32.  mode 0: dst[rows][cols] natural; 1: K-permuted;
  // 2: dst[cols][rows] = src^T with its K axis (the src row index) permuted; 3: src^T natural
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * cols) return;
  float v;
  if (mode <= 1) {
    const int n = idx / cols, k = idx - n * cols;
    const int kk = mode == 1 ? (k & ~31) + perm32(k & 31) : k;
    v = src[(int64_t)n * cols + kk];
  } else {
    const int n = idx / rows, k = idx - n * rows;       // dst row n = src column n
    const int kk = mode == 2 ? (k & ~31) + perm32(k & 31) : k;
    v = src[(int64_t)kk * cols + n];
  }
  dst[idx] = (__bf16)v;
}

// fp8 (e4m3) image of an nn.Linear weight [rows][cols]: dst = fp8(s * W) (mode 0 natural, 1
// K-permuted as k_prep_weight), s = the largest power of two with s * max|W| <= 448 (exact scaling);
// inv_scale[0] = 1 / s.  gridDim.x workgroups convert 4096 elements each; EVERY workgroup finds
// the maximum of the whole tensor for itself (256 KiB of L2 reads, 16-byte loads, all in flight) -
// the one-workgroup form walked the tensor twice with 64 dependent loads per thread: 20 us.
__device__ __forceinline__ void prep_weight_f8_body(const float* __restrict__ src,
                                                    uint8_t* __restrict__ dst, int rows, int cols,
                                                    int mode, float* __restrict__ inv_scale) {
  __shared__ float red[16];
  __shared__ float s_scale;
  const int tid = threadIdx.x, n = rows * cols;
  float am = 0.f;
  if ((n & 3) == 0) {
    for (int i0 = 4 * tid; i0 < n; i0 += 16 * 4096) {
      float4 v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u)
        v[u] = i0 + u * 4096 < n ? *reinterpret_cast<const float4*>(src + i0 + u * 4096)
                                 : float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 16; ++u)
        am = fmaxf(fmaxf(am, fmaxf(fabsf(v[u].x), fabsf(v[u].y))),
                   fmaxf(fabsf(v[u].z), fabsf(v[u].w)));
    }
  } else {
    for (int i = tid; i < n; i += 1024) am = fmaxf(am, fabsf(src[i]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) am = fmaxf(am, __shfl_xor(am, o, 64));
  if ((tid & 63) == 0) red[tid >> 6] = am;
  __syncthreads();
  if (tid == 0) {
    float m = 0.f;
    for (int i = 0; i < 16; ++i) m = fmaxf(m, red[i]);
    const float sc = m > 0.f ? exp2f(floorf(log2f(448.f / m))) : 1.f;
    s_scale = sc;
    if (blockIdx.x == 0) inv_scale[0] = 1.f / sc;
  }
  __syncthreads();
  const float sc = s_scale;
  for (int i0 = 4 * (blockIdx.x * 1024 + tid); i0 < n; i0 += 4096 * gridDim.x) {
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = i0 + u, r = idx / cols, k = idx - r * cols;
      const int kk = mode == 1 ? (k & ~31) + perm32(k & 31) : k;
      v[u] = idx < n ? src[(int64_t)r * cols + kk] * sc : 0.f;
    }
    *reinterpret_cast<uint32_t*>(dst + i0) = cvt4_f8(v[0], v[1], v[2], v[3]);
  }
}
__global__ __launch_bounds__(1024) void k_prep_weight_f8(const float* __restrict__ src,
                                                         uint8_t* __restrict__ dst, int rows,
                                                         int cols, int mode,
                                                         float* __restrict__ inv_scale) {
  prep_weight_f8_body(src, dst, rows, cols, mode, inv_scale);
}
// all fp8 weight images of a step in one launch (blockIdx.y = image)
__global__ __launch_bounds__(1024) void k_prep_weight_f8_jobs(const PrepF8Jobs jobs) {
  const WeightImages::F8 jb = jobs.j[blockIdx.y];
  if ((int64_t)blockIdx.x * 4096 >= (int64_t)jb.rows * jb.cols) return;
  prep_weight_f8_body(jb.src, jb.img, jb.rows, jb.cols, jb.mode, jb.inv);
}

__global__ void k_prep_jobs(const PrepJobs jobs) {
  const PrepJob jb = jobs.j[blockIdx.y];
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

__global__ void k_transpose_f32(const float* __restrict__ src, float* __restrict__ dst, int rows,
                                int cols) {
  __shared__ float t[32][33];
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + threadIdx.x;
    t[i][threadIdx.x] = (r < rows && c < cols) ? src[(int64_t)r * cols + c] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + threadIdx.x;
    if (r < rows && c < cols) dst[(int64_t)c * rows + r] = t[threadIdx.x][i];
  }
}

// ---------------------------------------------------------------------------------
// K/V projection of the m inducing-point outputs of one set (tiny): Kp = H Wk^T + bk,
// Vp = H Wv^T + bv, written in the four bf16 images the chain kernels read:
//   KpP[b][key][32j + p] = Kp[key][32j + perm32(p)]     (A operand of S^T = Kp_j . Qp_j^T)
//   VpP[b][key][32j + p] = Vp[key][32j + perm32(p)]     (A operand of dA^T = Vp_j . dO_j^T)
//   Kt [b][feat][kp]     = Kp[key(kp)][feat]            (A operand of dQp^T += Kp_j^T . dS^T)
//   Vt [b][feat][kp]     = Vp[key(kp)][feat]            (A operand of O^T  += Vp_j^T . P^T)
// with key(kp) = kp for m = 16 (16x16x16 MFMA, natural k) and perm32(kp) for m = 32.
// ---------------------------------------------------------------------------------
template <int MI>
__global__ __launch_bounds__(256) void k_kv_proj(const float* __restrict__ H,   // 64 or 256 threads
                                                 const float* __restrict__ WkT,   // [d][d] in x out
                                                 const float* __restrict__ bk,
                                                 const float* __restrict__ WvT,
                                                 const float* __restrict__ bv, int d,
                                                 __bf16* __restrict__ KpP,
                                                 __bf16* __restrict__ VpP,
                                                 __bf16* __restrict__ Kt,
                                                 __bf16* __restrict__ Vt) {
  extern __shared__ __attribute__((aligned(16))) float sH[];   // [MI][d]
  const int b = blockIdx.x;
  for (int i = threadIdx.x; i < MI * d; i += blockDim.x) sH[i] = H[(int64_t)b * MI * d + i];
  __syncthreads();
  // thread = (output feature f, K or V); d == 128 -> 256 threads cover both; wider blocks are
  // spread over blockIdx.y (a thread's MI * d MACs are the critical path)
  for (int o = blockIdx.y * blockDim.x + threadIdx.x; o < 2 * d; o += blockDim.x * gridDim.y) {
    const int f = o % d, isv = o / d;
    float acc[MI];
    const float bias = isv ? bv[f] : bk[f];
#pragma unroll
    for (int i = 0; i < MI; ++i) acc[i] = bias;
    col_gemm<MI>(sH, d, isv ? WvT : WkT, d, d, f, acc);
    // feature f sits at position pos inside its 32-block: f = 32j + perm32(pos)
    const int jb = f & ~31, fo = f & 31;
    int pos = 0;
#pragma unroll
    for (int p = 0; p < 32; ++p)
      if (perm32(p) == fo) pos = p;
    __bf16* PP = isv ? VpP : KpP;
    __bf16* TT = isv ? Vt : Kt;
#pragma unroll
    for (int i = 0; i < MI; ++i) PP[((int64_t)b * MI + i) * d + jb + pos] = (__bf16)acc[i];
#pragma unroll
    for (int kp = 0; kp < MI; ++kp) {
      const int key = (MI == 32) ? perm32(kp) : kp;
      TT[((int64_t)b * d + f) * MI + kp] = (__bf16)acc[key];
    }
  }
}

// The four images of k_kv_proj from projections that already exist in fp32 (d = 256: the
// projection itself is a [B m, d] x [d, d] product and goes through the GEMM kernel).
template <int MI>
__global__ __launch_bounds__(256) void k_kv_images(const float* __restrict__ Kf,   // [B][MI][d]
                                                   const float* __restrict__ Vf, int d,
                                                   __bf16* __restrict__ KpP,
                                                   __bf16* __restrict__ VpP,
                                                   __bf16* __restrict__ Kt,
                                                   __bf16* __restrict__ Vt) {
  const int b = blockIdx.x;
  const int o = blockIdx.y * 256 + threadIdx.x;
  if (o >= 2 * d) return;
  const int f = o % d, isv = o / d;
  const float* src = (isv ? Vf : Kf) + (int64_t)b * MI * d + f;
  float acc[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) acc[i] = src[(int64_t)i * d];
  const int jb = f & ~31, fo = f & 31;
  int pos = 0;
#pragma unroll
  for (int p = 0; p < 32; ++p)
    if (perm32(p) == fo) pos = p;
  __bf16* PP = isv ? VpP : KpP;
  __bf16* TT = isv ? Vt : Kt;
#pragma unroll
  for (int i = 0; i < MI; ++i) PP[((int64_t)b * MI + i) * d + jb + pos] = (__bf16)acc[i];
#pragma unroll
  for (int kp = 0; kp < MI; ++kp) {
    const int key = (MI == 32) ? perm32(kp) : kp;
    TT[((int64_t)b * d + f) * MI + kp] = (__bf16)acc[key];
  }
}

// ---------------------------------------------------------------------------------
// the fused forward kernel
// ---------------------------------------------------------------------------------
struct Mab1FwdArgs {
  const void* X;         // [B, N, dq]: fp32, or bf16 when ABF (dq == D)
  const __bf16* WqB;     // [D][D] natural (dq == D) ...
  const float* WqF;      // ... or fp32 [D][dq] for dq <= 4 (layer 1: exact VALU projection)
  const float* bq;
  const __bf16* KpP;     // [B][MI][D]
  const __bf16* Vt;      // [B][D][MI]
  const __bf16* WoP;     // [D][D] K-permuted
  const float* bo;
  // F8 kernels: the two images hold fp8 (e4m3) bytes of s * W, s a per-tensor power of two;
  // inv_scale[0] = 1 / s_q, inv_scale[1] = 1 / s_o (device floats written by k_prep_weight_f8)
  const float* inv_scale;
  void* Y;               // [B, N, D] fp32, or bf16 when ABF
  __bf16* QpS;           // [B*N][D] saved for backward (nullable)
  __bf16* OS;            // [B*N][D]
  uint32_t* mask;        // ReLU mask bits, see mask_index()
  int B, N, dq;
  int tiles_per_set;
  float scale_log2e;     // log2(e) / sqrt(d)
};

// ABF: activations (X when dq == D, and Y) are bf16 in memory
// NW waves per workgroup (4 or 8): the d -> d variant keeps 104 KiB of weights and staging in
// LDS, so only one workgroup fits a CU; eight waves sharing those weights give each SIMD two
// wavefronts to interleave (the chain is a long dependent sequence: ~0.3 instructions issued
// per wave-cycle at one wavefront per SIMD).  Waves 4..7 take the next 128-point tile.
//
// PHASE: d = 256 holds ONE 128 KiB weight image next to the K/V images of a set, so the d -> d
// block runs as two launches that meet at O (bf16, in the scratch block):
//   0  the whole chain (d = 128; d = 256 layer 1, which has no Wq image)
//   1  Q phase: X -> Qp -> attention -> O     (Wq + K/V images; X fragments straight from global)
//   2  O phase: Y = O + relu(O Wo^T + bo)     (Wo image)
// F8: the d x d projections (fc_q, fc_o) take fp8 e4m3 MFMA operands (weights pre-scaled per
// tensor, activations converted in registers); attention, softmax, residuals as in bf16 mode
// (F8 is a mask: bit 0 = fc_q, bit 1 = fc_o)
template <int D, int MI, bool DIN_SMALL, bool ABF, int NW, int PHASE, int F8 = 0>
__global__ __launch_bounds__(64 * NW, (NW == 8 || D > 128) ? 1 : 2) void k_mab1_fwd(
    const Mab1FwdArgs a) {
  constexpr int NT = 64 * NW, SUBS = NW / 4;
  constexpr int DT = D / 16;          // feature tiles
  constexpr int KS = D / 32;          // 32-wide K steps = heads (dh == 32)
  constexpr int ROWB = D * 2;         // bytes per row of a [.][D] bf16 image
  constexpr bool F8Q = (F8 & 1) != 0, F8O = (F8 & 2) != 0;
  constexpr int WQROWB = F8Q ? D : ROWB, WOROWB = F8O ? D : ROWB;   // ... of the weight images
  constexpr bool HAS_WO = PHASE != 1, HAS_KV = PHASE != 2;
  constexpr bool HAS_WQ = !DIN_SMALL && PHASE != 2;
  static_assert(!(DIN_SMALL && PHASE == 2), "layer 1 has no O phase of its own");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sWo = smem;                                   // D x D bf16, swizzled
  char* sKp = sWo + (HAS_WO ? D * WOROWB : 0);        // MI x D
  char* sVt = sKp + (HAS_KV ? MI * ROWB : 0);         // D x MI
  char* sWq = sVt + (HAS_KV ? D * MI * 2 : 0);        // D x D (absent when DIN_SMALL)
  char* sX = sWq + (HAS_WQ ? D * WQROWB : 0);         // TP x D (PHASE 0, d -> d only)
  // layer 1, d = 128: fc_q's [D][dq <= 4] weights (padded to 4) and bias, fp32
  constexpr bool WQ_LDS = DIN_SMALL && (D == 128 || PHASE == 1);
  float* sWqF = reinterpret_cast<float*>(sX);         // [D][4]
  float* sbq = sWqF + D * 4;                          // [D]

  const int tid = threadIdx.x, lane = tid & 63, wave8 = tid >> 6;
  const int wave = wave8 & 3, sub = wave8 >> 2;
  const int r = lane & 15, g = lane >> 4;

  // ---- weights -> LDS once per workgroup (16-byte chunks, swizzled rows) ----
  {
    // all chunks of this thread first, then the LDS stores: one round trip instead of one per
    // iteration (a store between two loads orders them)
    // (fp8 images: rows of D bytes, a 16-byte chunk = two 8-byte slots of the swizzled image)
    auto stage = [&](const void* gsrc, char* sdst, auto f8tag) {
      constexpr bool IS8 = decltype(f8tag)::value;
      constexpr int RB = IS8 ? D : ROWB, CPR = RB / 16, NC = D * CPR / NT;
      uint4 w[NC];
      const char* g8 = reinterpret_cast<const char*>(gsrc);
#pragma unroll
      for (int e = 0; e < NC; ++e) {
        const int c = tid + NT * e, row = c / CPR, c16 = c % CPR;
        w[e] = *reinterpret_cast<const uint4*>(g8 + (int64_t)row * RB + c16 * 16);
      }
#pragma unroll
      for (int e = 0; e < NC; ++e) {
        const int c = tid + NT * e, row = c / CPR, c16 = c % CPR;
        if (IS8) {
          *reinterpret_cast<uint2*>(sdst + f8off<D>(row, 2 * c16)) = uint2{w[e].x, w[e].y};
          *reinterpret_cast<uint2*>(sdst + f8off<D>(row, 2 * c16 + 1)) = uint2{w[e].z, w[e].w};
        } else {
          *reinterpret_cast<uint4*>(sdst + swz(row, c16, ROWB)) = w[e];
        }
      }
    };
    if (HAS_WO) stage(a.WoP, sWo, std::integral_constant<bool, F8O>{});
    if (HAS_WQ) stage(a.WqB, sWq, std::integral_constant<bool, F8Q>{});
  }
  const float invq = F8Q ? a.inv_scale[0] : 1.f, invo = F8O ? a.inv_scale[1] : 1.f;
  (void)invq; (void)invo;

  if (WQ_LDS) {
    for (int f = tid; f < D; f += NT) {
      float w[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) w[c] = c < a.dq ? a.WqF[f * a.dq + c] : 0.f;
      *reinterpret_cast<float4*>(sWqF + 4 * f) = float4{w[0], w[1], w[2], w[3]};
      sbq[f] = a.bq[f];
    }
  }

  const int units_per_set = (a.tiles_per_set + SUBS - 1) / SUBS;   // SUBS tiles per workgroup pass
  const int total_units = a.B * units_per_set;
  int cur_b = -1;
  for (int unit = blockIdx.x; unit < total_units; unit += gridDim.x) {
    const int b = unit / units_per_set, tile = (unit - b * units_per_set) * SUBS + sub;
    if (HAS_KV && b != cur_b) {             // this set's Kp / Vp images
      __syncthreads();
      for (int c = tid; c < MI * (D / 8); c += NT) {
        const int row = c / (D / 8), c16 = c % (D / 8);
        *reinterpret_cast<uint4*>(sKp + swz(row, c16, ROWB)) = *reinterpret_cast<const uint4*>(
            a.KpP + ((int64_t)b * MI + row) * D + c16 * 8);
      }
      for (int c = tid; c < D * MI / 8; c += NT)
        reinterpret_cast<uint4*>(sVt)[c] =
            reinterpret_cast<const uint4*>(a.Vt + (int64_t)b * D * MI)[c];
      cur_b = b;
    }
    __syncthreads();
    if (tile >= a.tiles_per_set) continue;              // odd tile count: waves 4..7 idle
                                                        // (no barrier below this point)
    const int n_base = tile * TP + wave * 32;           // first point of this wave
    f32x4 acc[DT][NB];

    if (PHASE == 2) {
      // O phase: the tile of O comes back from the scratch block (bf16, [point][feature])
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int n = n_base + 16 * nb + r;
#pragma unroll
        for (int t = 0; t < DT; ++t) {
          bf16x4 o4 = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
          if (n < a.N)
            o4 = *reinterpret_cast<const bf16x4*>(a.OS + ((int64_t)b * a.N + n) * D + 16 * t +
                                                  4 * g);
          acc[t][nb] = f32x4{(float)o4[0], (float)o4[1], (float)o4[2], (float)o4[3]};
        }
      }
    } else if (DIN_SMALL) {
      // layer 1: Qp = x Wq^T + bq with dq in 1..4, exact fp32 on the vector ALU
      float xv[NB][4];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int n = n_base + 16 * nb + r;
#pragma unroll
        for (int c = 0; c < 4; ++c)
          xv[nb][c] = (n < a.N && c < a.dq)
                          ? reinterpret_cast<const float*>(a.X)[((int64_t)b * a.N + n) * a.dq + c]
                          : 0.f;
      }
#pragma unroll
      for (int t = 0; t < DT; ++t) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int f = 16 * t + 4 * g + e;
          float w[4];
          float bias;
          if (WQ_LDS) {
            const float4 w4 = *reinterpret_cast<const float4*>(sWqF + 4 * f);
            w[0] = w4.x; w[1] = w4.y; w[2] = w4.z; w[3] = w4.w;
            bias = sbq[f];
          } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) w[c] = c < a.dq ? a.WqF[f * a.dq + c] : 0.f;
            bias = a.bq[f];
          }
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            acc[t][nb][e] = bias + w[0] * xv[nb][0] + w[1] * xv[nb][1] + w[2] * xv[nb][2] +
                            w[3] * xv[nb][3];
        }
      }
    } else if (PHASE == 1) {
      // Q phase (d = 256): no room for an X tile in LDS; a lane's B fragment is 8 consecutive
      // features of its point, i.e. 16 (bf16) or 32 (fp32) contiguous bytes of global memory
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        const float4 b4 = F8Q ? float4{0.f, 0.f, 0.f, 0.f}
                             : *reinterpret_cast<const float4*>(a.bq + 16 * t + 4 * g);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[t][nb] = f32x4{b4.x, b4.y, b4.z, b4.w};
      }
      bf16x8 bxa[KS][NB];
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const int n = n_base + 16 * nb + r;
          const int64_t xo = ((int64_t)b * a.N + n) * D + 32 * s + 8 * g;
          bf16x8 v;
          if (n < a.N && ABF) {
            v = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(a.X) + xo);
          } else if (n < a.N) {
            const float4* src =
                reinterpret_cast<const float4*>(reinterpret_cast<const float*>(a.X) + xo);
            const float4 lo = src[0], hi = src[1];
            v[0] = (__bf16)lo.x; v[1] = (__bf16)lo.y; v[2] = (__bf16)lo.z; v[3] = (__bf16)lo.w;
            v[4] = (__bf16)hi.x; v[5] = (__bf16)hi.y; v[6] = (__bf16)hi.z; v[7] = (__bf16)hi.w;
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (__bf16)0.f;
          }
          bxa[s][nb] = v;
        }
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        f8x8 bx8[NB];
        if (F8Q) {
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) bx8[nb] = bf_to_f8(bxa[s][nb]);
        }
#pragma unroll
        for (int t = 0; t < DT; ++t) {
          if (F8Q) {
            const f8x8 wa8 = *reinterpret_cast<const f8x8*>(sWq + f8off<D>(16 * t + r, 4 * s + g));
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[t][nb] = mfma32_f8(wa8, bx8[nb], acc[t][nb]);
          } else {
            const bf16x8 wa =
                *reinterpret_cast<const bf16x8*>(sWq + swz(16 * t + r, 4 * s + g, ROWB));
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[t][nb] = mfma32(wa, bxa[s][nb], acc[t][nb]);
          }
        }
      }
    } else {
      // stage this wave's 32 rows of X (fp32 -> bf16) into its private slice of sX
      char* myX = sX + wave8 * 32 * ROWB;
      for (int c = lane; c < 32 * (D / 8); c += 64) {
        const int row = c / (D / 8), c16 = c % (D / 8);
        const int n = n_base + row;
        bf16x8 v;
        if (n < a.N && ABF) {
          v = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(a.X) +
                                               ((int64_t)b * a.N + n) * D + c16 * 8);
        } else if (n < a.N) {
          const float4* src = reinterpret_cast<const float4*>(
              reinterpret_cast<const float*>(a.X) + ((int64_t)b * a.N + n) * D + c16 * 8);
          const float4 lo = src[0], hi = src[1];
          v[0] = (__bf16)lo.x; v[1] = (__bf16)lo.y; v[2] = (__bf16)lo.z; v[3] = (__bf16)lo.w;
          v[4] = (__bf16)hi.x; v[5] = (__bf16)hi.y; v[6] = (__bf16)hi.z; v[7] = (__bf16)hi.w;
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (__bf16)0.f;
        }
        *reinterpret_cast<bf16x8*>(myX + swz(row, c16, ROWB)) = v;
      }
      // (wave-private: the LDS writes above are ordered before the reads below by the
      //  compiler's s_waitcnt; no workgroup barrier needed)
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        const float4 b4 = F8Q ? float4{0.f, 0.f, 0.f, 0.f}
                             : *reinterpret_cast<const float4*>(a.bq + 16 * t + 4 * g);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[t][nb] = f32x4{b4.x, b4.y, b4.z, b4.w};
      }
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        bf16x8 bx[NB];
        f8x8 bx8[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          bx[nb] = *reinterpret_cast<const bf16x8*>(myX + swz(16 * nb + r, 4 * s + g, ROWB));
          if (F8Q) bx8[nb] = bf_to_f8(bx[nb]);
        }
#pragma unroll
        for (int t = 0; t < DT; ++t) {
          if (F8Q) {
            const f8x8 wa8 = *reinterpret_cast<const f8x8*>(sWq + f8off<D>(16 * t + r, 4 * s + g));
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[t][nb] = mfma32_f8(wa8, bx8[nb], acc[t][nb]);
          } else {
            const bf16x8 wa =
                *reinterpret_cast<const bf16x8*>(sWq + swz(16 * t + r, 4 * s + g, ROWB));
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[t][nb] = mfma32(wa, bx[nb], acc[t][nb]);
          }
        }
      }
    }
    if (F8Q && !DIN_SMALL && PHASE != 2) {
      // undo the per-tensor weight scale, then the bias (fp32)
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        const float4 b4 = *reinterpret_cast<const float4*>(a.bq + 16 * t + 4 * g);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          acc[t][nb][0] = acc[t][nb][0] * invq + b4.x;
          acc[t][nb][1] = acc[t][nb][1] * invq + b4.y;
          acc[t][nb][2] = acc[t][nb][2] * invq + b4.z;
          acc[t][nb][3] = acc[t][nb][3] * invq + b4.w;
        }
      }
    }

    if (PHASE != 2) {
    // ---- save Qp for the backward (bf16, [point][feature]) ----
    if (a.QpS != nullptr) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int n = n_base + 16 * nb + r;
        if (n < a.N) {
#pragma unroll
          for (int t = 0; t < DT; ++t)
            *reinterpret_cast<bf16x4*>(a.QpS + ((int64_t)b * a.N + n) * D + 16 * t + 4 * g) =
                pack4(acc[t][nb]);
        }
      }
    }

    // ---- attention over the MI inducing keys, head j = feature tiles 2j, 2j+1 ----
#pragma unroll
    for (int j = 0; j < KS; ++j) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const bf16x8 qb = pack8(acc[2 * j][nb], acc[2 * j + 1][nb]);
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
        s0 = mfma32(*reinterpret_cast<const bf16x8*>(sKp + swz(r, 4 * j + g, ROWB)), qb, s0);
        if (MI == 32)
          s1 = mfma32(*reinterpret_cast<const bf16x8*>(sKp + swz(16 + r, 4 * j + g, ROWB)), qb,
                      s1);
        float mx = fmaxf(fmaxf(s0[0], s0[1]), fmaxf(s0[2], s0[3]));
        if (MI == 32) mx = fmaxf(mx, fmaxf(fmaxf(s1[0], s1[1]), fmaxf(s1[2], s1[3])));
        mx = wave16_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s0[e] = exp2f((s0[e] - mx) * a.scale_log2e);
          sum += s0[e];
          if (MI == 32) {
            s1[e] = exp2f((s1[e] - mx) * a.scale_log2e);
            sum += s1[e];
          }
        }
        sum = wave16_sum(sum);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int e = 0; e < 4; ++e) { s0[e] *= inv; s1[e] *= inv; }
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          const int t = 2 * j + tt;
          const char* vrow = sVt + (16 * t + r) * (MI * 2);
          if (MI == 16) {
            acc[t][nb] = mfma16(*reinterpret_cast<const bf16x4*>(vrow + 8 * g), pack4(s0),
                                acc[t][nb]);
          } else {
            acc[t][nb] = mfma32(*reinterpret_cast<const bf16x8*>(vrow + 16 * g), pack8(s0, s1),
                                acc[t][nb]);
          }
        }
      }
    }

    // ---- save O ----
    if (a.OS != nullptr) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int n = n_base + 16 * nb + r;
        if (n < a.N) {
#pragma unroll
          for (int t = 0; t < DT; ++t)
            *reinterpret_cast<bf16x4*>(a.OS + ((int64_t)b * a.N + n) * D + 16 * t + 4 * g) =
                pack4(acc[t][nb]);
        }
      }
    }

    }   // PHASE != 2
    if (PHASE == 1) continue;               // O is in the scratch block; the O phase finishes

    // ---- GEMM2: Z^T = Wo . O^T ; Y = O + relu(Z + bo) ----
    // (d = 256: the output features in two halves of 128, so that Z needs 64 registers)
    constexpr int HT = D / 128, DTH = DT / HT;
    bf16x8 oball[(HT > 1 && !F8O) ? KS : 1][NB];      // packed once when it is used twice
    f8x8 oball8[(HT > 1 && F8O) ? KS : 1][NB];
    if (HT > 1) {
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          if (F8O) oball8[(HT > 1 && F8O) ? s : 0][nb] = pack8_f8(acc[2 * s][nb], acc[2 * s + 1][nb]);
          else oball[(HT > 1 && !F8O) ? s : 0][nb] = pack8(acc[2 * s][nb], acc[2 * s + 1][nb]);
        }
    }
#pragma unroll
    for (int hf = 0; hf < HT; ++hf) {
      f32x4 z[DTH][NB];
#pragma unroll
      for (int tt = 0; tt < DTH; ++tt) {
        const float4 b4 = F8O ? float4{0.f, 0.f, 0.f, 0.f}
                             : *reinterpret_cast<const float4*>(a.bo + 16 * (hf * DTH + tt) + 4 * g);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) z[tt][nb] = f32x4{b4.x, b4.y, b4.z, b4.w};
      }
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        if (F8O) {
          f8x8 ob8[NB];
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            ob8[nb] = HT > 1 ? oball8[(HT > 1 && F8O) ? s : 0][nb]
                             : pack8_f8(acc[2 * s][nb], acc[2 * s + 1][nb]);
#pragma unroll
          for (int tt = 0; tt < DTH; ++tt) {
            const f8x8 wa8 = *reinterpret_cast<const f8x8*>(
                sWo + f8off<D>(16 * (hf * DTH + tt) + r, 4 * s + g));
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) z[tt][nb] = mfma32_f8(wa8, ob8[nb], z[tt][nb]);
          }
        } else {
          bf16x8 ob[NB];
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            ob[nb] = HT > 1 ? oball[(HT > 1 && !F8O) ? s : 0][nb]
                            : pack8(acc[2 * s][nb], acc[2 * s + 1][nb]);
#pragma unroll
          for (int tt = 0; tt < DTH; ++tt) {
            const bf16x8 wa = *reinterpret_cast<const bf16x8*>(
                sWo + swz(16 * (hf * DTH + tt) + r, 4 * s + g, ROWB));
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) z[tt][nb] = mfma32(wa, ob[nb], z[tt][nb]);
          }
        }
      }
      if (F8O) {
#pragma unroll
        for (int tt = 0; tt < DTH; ++tt) {
          const float4 b4 =
              *reinterpret_cast<const float4*>(a.bo + 16 * (hf * DTH + tt) + 4 * g);
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            z[tt][nb][0] = z[tt][nb][0] * invo + b4.x;
            z[tt][nb][1] = z[tt][nb][1] * invo + b4.y;
            z[tt][nb][2] = z[tt][nb][2] * invo + b4.z;
            z[tt][nb][3] = z[tt][nb][3] * invo + b4.w;
          }
        }
      }
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int n = n_base + 16 * nb + r;
        uint32_t bits = 0u;                 // mask word hf covers feature tiles 8 hf .. 8 hf + 7
#pragma unroll
        for (int tt = 0; tt < DTH; ++tt) {
          const int t = hf * DTH + tt;
          f32x4 y;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float zz = z[tt][nb][e];
            y[e] = acc[t][nb][e] + fmaxf(zz, 0.f);
            if (zz > 0.f) bits |= 1u << ((t & 7) * 4 + e);
          }
          if (n < a.N) {
            const int64_t yo = ((int64_t)b * a.N + n) * D + 16 * t + 4 * g;
            if (ABF) *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(a.Y) + yo) = pack4(y);
            else *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.Y) + yo) =
                     float4{y[0], y[1], y[2], y[3]};
          }
        }
        if (a.mask != nullptr)
          a.mask[mab1_mask_index<D>(b, a.tiles_per_set, tile, wave, nb, hf, lane)] = bits;
      }
    }
  }
}

template <int D, int MI, bool DS, bool ABF, int PHASE = 0, int F8 = 0>
int launch_fwd(const Mab1FwdArgs& a, hipStream_t st) {
  // layer 1 (40 KiB of LDS): 4 waves, two workgroups per CU; d -> d (104+ KiB): 8 waves;
  // d = 256 (128 - 160 KiB): one workgroup per CU, 8 waves (layer 1: 4 waves, both the Qp and
  // the Z tiles live in registers)
  constexpr int NW = D > 128 ? ((DS && PHASE == 0) ? 4 : 8) : (DS || MI != 16) ? 4 : 8;
  const size_t wimg_o = (size_t)D * D * ((F8 & 2) ? 1 : 2), wimg_q = (size_t)D * D * ((F8 & 1) ? 1 : 2);
  const size_t kv = (size_t)MI * D * 2 + (size_t)D * MI * 2;
  const size_t lds = (PHASE != 1 ? wimg_o : 0) + (PHASE != 2 ? kv : 0) +
                     ((!DS && PHASE != 2) ? wimg_q : 0) +
                     ((!DS && PHASE == 0) ? (size_t)NW * 32 * D * 2 : 0) +
                     ((DS && (D == 128 || PHASE == 1)) ? (size_t)D * 5 * sizeof(float) : 0);
  static std::once_flag once;
  std::call_once(once, [] {
    (void)hipFuncSetAttribute(
        reinterpret_cast<const void*>(k_mab1_fwd<D, MI, DS, ABF, NW, PHASE, F8>),
        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  const int total = a.B * ((a.tiles_per_set + NW / 4 - 1) / (NW / 4));
  const int cap = (DS && D == 128) ? 512 : 256;
  const int grid = total < cap ? total : cap;
  const double pts = (double)a.B * a.N;
  // PHASE 1 / 2 split the work of one block; the pair is one profiled unit
  const double fl = 2.0 * pts * ((PHASE != 2 ? (double)a.dq * D + 2.0 * MI * D : 0.0) +
                                 (PHASE != 1 ? (double)D * D : 0.0));
  const double by = pts * ((PHASE != 2 ? (ABF && !DS ? 2.0 : 4.0) * a.dq : 0.0) +
                           (PHASE != 1 ? (ABF ? 2.0 : 4.0) * D : 0.0));
  ProfScope ps(PCA_K_MAB1_FWD, st, fl, by);
  hipLaunchKernelGGL((k_mab1_fwd<D, MI, DS, ABF, NW, PHASE, F8>), dim3(grid), dim3(64 * NW), lds,
                     st, a);
  ps.end();
  return check_launch("k_mab1_fwd");
}

}  // namespace

// ---- host side --------------------------------------------------------------------
int transpose_f32(const float* src, float* dst, int rows, int cols, hipStream_t st) {
  hipLaunchKernelGGL(k_transpose_f32, dim3((unsigned)cdiv(cols, 32), (unsigned)cdiv(rows, 32)),
                     dim3(32, 8), 0, st, src, dst, rows, cols);
  return check_launch("k_transpose_f32");
}
int prep_jobs_launch(const PrepJobs& jobs, hipStream_t st) {
  if (jobs.n == 0) return PCA_OK;
  int maxe = 0;
  for (int i = 0; i < jobs.n; ++i) {
    const int e = jobs.j[i].rows * jobs.j[i].cols;
    maxe = e > maxe ? e : maxe;
  }
  hipLaunchKernelGGL(k_prep_jobs, dim3((unsigned)cdiv(maxe, 256), (unsigned)jobs.n), dim3(256), 0,
                     st, jobs);
  return check_launch("k_prep_jobs");
}
int prep_weight_f8(const float* src, void* dst, int rows, int cols, int mode, float* inv_scale,
                   hipStream_t st) {
  hipLaunchKernelGGL(k_prep_weight_f8, dim3((unsigned)cdiv((int64_t)rows * cols, 4096)), dim3(1024), 0, st, src,
                     reinterpret_cast<uint8_t*>(dst), rows, cols, mode, inv_scale);
  return check_launch("k_prep_weight_f8");
}
// two images of equally shaped weights in one launch (blockIdx.y = image): every launch of a kernel
// this small costs ~5 us of GPU time whatever it does
__global__ void k_prep_weight2(const float* __restrict__ src0, __bf16* __restrict__ dst0, int mode0,
                               const float* __restrict__ src1, __bf16* __restrict__ dst1, int mode1,
                               int rows, int cols) {
  const float* __restrict__ src = blockIdx.y == 0 ? src0 : src1;
  __bf16* __restrict__ dst = blockIdx.y == 0 ? dst0 : dst1;
  const int mode = blockIdx.y == 0 ? mode0 : mode1;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * cols) return;
  float v;
  if (mode <= 1) {
    const int n = idx / cols, k = idx - n * cols;
    const int kk = mode == 1 ? (k & ~31) + perm32(k & 31) : k;
    v = src[(int64_t)n * cols + kk];
  } else {
    const int n = idx / rows, k = idx - n * rows;
    const int kk = mode == 2 ? (k & ~31) + perm32(k & 31) : k;
    v = src[(int64_t)kk * cols + n];
  }
  dst[idx] = (__bf16)v;
}
int prep_weight2(const float* src0, __bf16* dst0, int mode0, const float* src1, __bf16* dst1,
                 int mode1, int rows, int cols, hipStream_t st) {
  hipLaunchKernelGGL(k_prep_weight2, dim3((unsigned)cdiv((int64_t)rows * cols, 256), 2), dim3(256),
                     0, st, src0, dst0, mode0, src1, dst1, mode1, rows, cols);
  return check_launch("k_prep_weight2");
}
int prep_weight(const float* src, __bf16* dst, int rows, int cols, int mode, hipStream_t st) {
  hipLaunchKernelGGL(k_prep_weight, dim3((unsigned)cdiv((int64_t)rows * cols, 256)), dim3(256), 0,
                     st, src, dst, rows, cols, mode);
  return check_launch("k_prep_weight");
}

static thread_local const WeightImages* t_images = nullptr;
void weight_images_use(const WeightImages* t) { t_images = t; }
bool weight_images_active() { return t_images != nullptr; }
static __bf16* image_of(const float* src, int mode, int rows, int cols) {
  if (t_images == nullptr) return nullptr;
  for (int i = 0; i < t_images->n; ++i) {
    const WeightImages::E& e = t_images->e[i];
    if (e.src == src && e.mode == mode && e.rows == rows && e.cols == cols) return e.img;
  }
  return nullptr;
}
int prep_f8_jobs_launch(const PrepF8Jobs& J, hipStream_t st) {
  if (J.n == 0) return PCA_OK;
  int64_t maxe = 0;
  for (int i = 0; i < J.n; ++i) {
    const int64_t e = (int64_t)J.j[i].rows * J.j[i].cols;
    maxe = e > maxe ? e : maxe;
  }
  hipLaunchKernelGGL(k_prep_weight_f8_jobs, dim3((unsigned)cdiv(maxe, 4096), (unsigned)J.n), dim3(1024),
                     0, st, J);
  return check_launch("k_prep_weight_f8_jobs");
}
int weight_image_f8(const float* src, void** dst, int rows, int cols, int mode, float** inv,
                    hipStream_t st) {
  if (t_images != nullptr)
    for (int i = 0; i < t_images->nf8; ++i) {
      const WeightImages::F8& e = t_images->f8[i];
      if (e.src == src && e.mode == mode && e.rows == rows && e.cols == cols) {
        *dst = e.img;
        *inv = e.inv;
        return PCA_OK;
      }
    }
  return prep_weight_f8(src, *dst, rows, cols, mode, *inv, st);
}
int weight_image1(const float* src, __bf16** dst, int rows, int cols, int mode, hipStream_t st) {
  if (__bf16* im = image_of(src, mode, rows, cols)) { *dst = im; return PCA_OK; }
  return prep_weight(src, *dst, rows, cols, mode, st);
}
int weight_image2(const float* src0, __bf16** dst0, int mode0, const float* src1, __bf16** dst1,
                  int mode1, int rows, int cols, hipStream_t st) {
  __bf16* i0 = image_of(src0, mode0, rows, cols);
  __bf16* i1 = image_of(src1, mode1, rows, cols);
  if (i0 == nullptr && i1 == nullptr)
    return prep_weight2(src0, *dst0, mode0, src1, *dst1, mode1, rows, cols, st);
  if (i0 != nullptr) *dst0 = i0; else PCA_TRY(prep_weight(src0, *dst0, rows, cols, mode0, st));
  if (i1 != nullptr) *dst1 = i1; else PCA_TRY(prep_weight(src1, *dst1, rows, cols, mode1, st));
  return PCA_OK;
}

// PCA_D256_FUSED=0: the two-launch form (Q phase + row-GEMM O phase) for A/B measurements
static bool fused256_on() {
  return true;        // (the two-launch Q + O form of round 1 still serves fp32 activations)
}

// Does the training forward save the projected queries?  Not at layer 1 (two or three input
// columns): there the backward recomputes them from the points - k_mab1_bwd<.., FUSE_WQ> at d = 128,
// k_attn1_bwd3<256, SMALLQ> at d = 256 with bf16 activations (the fc_o-fused backward).
bool mab1_saves_qp(const pca_mab_shape& s) {
  if (s.dq > 4) return true;
  if (s.nk == 16 && s.dq <= 3) return false;
  if (s.d == 256 && s.nk == 32) return false;       // k_attn1_bwd3<256, SMALLQ> recomputes them
  return true;
}

int mab1_fwd_wo_mode(const pca_mab_shape& s) {
  return (s.d == 256 && s.y_dtype == PCA_BF16 && fused256_on()) ? 0 : 1;
}

bool mab1_bf16_supported(const pca_mab_shape& s, bool inference) {
  // activations: fp32 everywhere, or bf16 for Y (and for X when it is a hidden tensor)
  const bool dt_ok = s.k_dtype == PCA_F32 &&
                     (s.dq <= 4 ? s.q_dtype == PCA_F32 : s.q_dtype == s.y_dtype);
  // d = 256 / m = 32 / 8 heads (BASELINE configs[3]): forward here, backward in d256_*.hip
  (void)inference;
  const bool d_ok = s.d == 128 || (s.d == 256 && s.nk == 32);
  // fp8 projections: d = 128 / m = 16 and d = 256 with bf16 activations
  const bool f8_ok = s.mode != PCA_MODE_FP8 || (s.d == 128 && s.nk == 16) ||
                     (s.d == 256 && s.y_dtype == PCA_BF16);
  return s.q_shared == 0 && d_ok && s.h * 32 == s.d && (s.nk == 16 || s.nk == 32) &&
         s.dk == s.d && (s.dq == s.d || s.dq <= 4) && dt_ok && f8_ok;
}

size_t mab1_carve_saved(const pca_mab_shape& s, Mab1Saved* out, void* base) {
  Carver c(base);
  Mab1Saved v;
  const size_t kv = (size_t)s.B * s.nk * s.d;
  const int tiles = (int)cdiv(s.nq, TP);
  v.KpP = c.take<__bf16>(kv);
  v.VpP = c.take<__bf16>(kv);
  v.Kt = c.take<__bf16>(kv);
  v.Vt = c.take<__bf16>(kv);
  v.QpS = c.take<__bf16>((size_t)s.B * s.nq * s.d);
  v.OS = c.take<__bf16>((size_t)s.B * s.nq * s.d);
  v.mask = c.take<uint32_t>((size_t)s.B * tiles * (TP / 16) * (s.d / 128) * 64);
  if (out) *out = v;
  return c.off;
}

size_t mab1_bf16_saved_bytes(const pca_mab_shape& s) {
  return mab1_carve_saved(s, nullptr, nullptr);
}
size_t mab1_bf16_fwd_ws_bytes(const pca_mab_shape& s) {
  return 256 + 2 * align256((size_t)s.d * s.d * 2) + 2 * align256((size_t)s.d * s.d * 4) +
         (s.d > 128 ? 2 * align256((size_t)s.B * s.nk * s.d * 4) : 0) +      // fp32 Kp, Vp
         mab1_carve_saved(s, nullptr, nullptr);
}

// Q = X [B, nq, dq] fp32, K = H [B, nk, d] fp32 -> Y [B, nq, d] fp32
int mab1_bf16_fwd(const pca_mab_shape& s, const void* X, const float* H,
                  const pca_mab_params& p, void* Y, void* saved, void* ws, hipStream_t st) {
  return mab1_bf16_fwd_ex(s, X, H, p, Y, saved, ws, 0, st);
}
int mab1_bf16_fwd_ex(const pca_mab_shape& s, const void* X, const float* H,
                     const pca_mab_params& p, void* Y, void* saved, void* ws, int flags,
                     hipStream_t st, const IsabImg* img) {
  PCA_REQUIRE(mab1_bf16_supported(s, saved == nullptr), "mab1_bf16_fwd: unsupported shape");
  PCA_REQUIRE(ws != nullptr, "mab1_bf16_fwd: scratch required");
  Carver cw(ws);
  __bf16* WqB = cw.take<__bf16>((size_t)s.d * s.d);
  __bf16* WoP = cw.take<__bf16>((size_t)s.d * s.d);
  float* WkT = cw.take<float>((size_t)s.d * s.d);
  float* WvT = cw.take<float>((size_t)s.d * s.d);
  float* Kf = s.d > 128 ? cw.take<float>((size_t)s.B * s.nk * s.d) : nullptr;
  float* Vf = s.d > 128 ? cw.take<float>((size_t)s.B * s.nk * s.d) : nullptr;
  float* invs = cw.take<float>(2);      // fp8 mode: inverse scales of the two weight images
  float* inv_o = invs + 1;
  Mab1Saved v;
  const bool training = saved != nullptr;
  mab1_carve_saved(s, &v, training ? saved : (void*)(cw.base + cw.off));
  const int d = s.d;
  const bool small = s.dq <= 4;

  const bool f8 = s.mode == PCA_MODE_FP8;
  // Which projections take fp8 operands.  Measured on the reference-trained cfg1 model (10 000
  // sets, argmax agreement with the reference's fp32 logits; bf16 mode: 99.97 %):
  //   fc_o only       99.83 %   <- default: meets the 99.8 % bar of SURVEY.md 8d
  //   fc_q and fc_o   99.39 %   (PCA_FP8_PROJ=qo): the 4 significant bits of e4m3 on the projected
  //                             queries perturb the softmax logits of every head
  constexpr bool f8_q = false;      // (fc_q in fp8 as well fails the 99.8 % bar: not selectable any more)
  // d = 256, bf16 activations, fc_o in fp8 (the default split): the single-launch kernel
  const bool f8_fused = f8 && !f8_q && d == 256 && s.y_dtype == PCA_BF16 && fused256_on();
  if (f8) {
    // fp8 images of s * Wq (natural) and s * Wo (K-permuted; natural for the single-launch kernel);
    // bf16 image of Wq when it stays bf16
    if (!small && f8_q) PCA_TRY(prep_weight_f8(p.wq, WqB, d, d, 0, invs, st));
    else if (!small) PCA_TRY(weight_image1(p.wq, &WqB, d, d, 0, st));
    if (f8_fused) {          // (the single-launch kernel takes its inverse scale by pointer: the engine's
                             //  one-launch image of the step, when there is one)
      void* wo8 = WoP;
      PCA_TRY(weight_image_f8(p.wo, &wo8, d, d, 0, &inv_o, st));
      WoP = reinterpret_cast<decltype(WoP)>(wo8);
    } else {
      PCA_TRY(prep_weight_f8(p.wo, WoP, d, d, 1, invs + 1, st));
    }
  } else if (img != nullptr) {
    WqB = img->WqB;
    WoP = img->WoP;
  } else {
    // (d = 256 with bf16 activations runs k_isab1_fwd256, which takes Wo as a natural image too)
    const int wo_mode = mab1_fwd_wo_mode(s);
    if (!small) PCA_TRY(weight_image2(p.wq, &WqB, 0, p.wo, &WoP, wo_mode, d, d, st));
    else PCA_TRY(weight_image1(p.wo, &WoP, d, d, wo_mode, st));
  }
  if (d > 128 && training && mid256_kv_ready()) flags |= PCA_F_KV_READY;   // images written by mid256_fwd
  if (!(flags & PCA_F_KV_READY) && d > 128) {
    // Kp = H Wk^T + bk, Vp = H Wv^T + bv as MFMA products (fp32 accumulation; the operand
    // rounding is an order of magnitude below the bf16 rounding of the images), then the images
    pca_gemm_desc g{};
    g.M = (int64_t)s.B * s.nk; g.N = d; g.K = d;
    g.sa_m = d; g.sa_k = 1; g.sb_k = 1; g.sb_n = d; g.sc_m = d;
    g.nb1 = g.nb2 = 1; g.split_k = 1; g.alpha = 1.f;
    PCA_TRY(gemm_bf16(g, H, p.wk, p.bk, Kf, st));
    PCA_TRY(gemm_bf16(g, H, p.wv, p.bv, Vf, st));
    hipLaunchKernelGGL((k_kv_images<32>), dim3(s.B, 2 * d / 256), dim3(256), 0, st, Kf, Vf, d,
                       v.KpP, v.VpP, v.Kt, v.Vt);
    PCA_TRY(check_launch("k_kv_images"));
  } else if (!(flags & PCA_F_KV_READY)) {
  PCA_TRY(transpose_f32(p.wk, WkT, d, d, st));
  PCA_TRY(transpose_f32(p.wv, WvT, d, d, st));
  const size_t hl = (size_t)s.nk * d * sizeof(float);
  if (s.nk == 16)
    hipLaunchKernelGGL((k_kv_proj<16>), dim3(s.B), dim3(256), hl, st, H, WkT, p.bk, WvT, p.bv,
                       d, v.KpP, v.VpP, v.Kt, v.Vt);
  else
    hipLaunchKernelGGL((k_kv_proj<32>), dim3(s.B), dim3(256), hl, st, H, WkT, p.bk, WvT, p.bv,
                       d, v.KpP, v.VpP, v.Kt, v.Vt);
  PCA_TRY(check_launch("k_kv_proj"));
  }

  Mab1FwdArgs a{};
  a.X = X; a.WqB = WqB; a.WqF = p.wq; a.bq = p.bq; a.KpP = v.KpP; a.Vt = v.Vt; a.WoP = WoP;
  a.bo = p.bo; a.Y = Y;
  a.inv_scale = f8 ? invs : nullptr;
  // layer 1 (dq <= 3, m = 16): the backward recomputes Qp from the points - nothing to save
  a.QpS = (training && mab1_saves_qp(s)) ? v.QpS : nullptr;
  a.OS = training ? v.OS : nullptr;
  a.mask = training ? v.mask : nullptr;
  a.B = s.B; a.N = s.nq; a.dq = s.dq;
  a.tiles_per_set = (int)cdiv(s.nq, TP);
  a.scale_log2e = 1.4426950408889634f / sqrtf((float)d);
  const bool abf = s.y_dtype == PCA_BF16;
  if (s.nk == 16 && f8) {
    if (small) return abf ? launch_fwd<128, 16, true, true, 0, 2>(a, st)
                          : launch_fwd<128, 16, true, false, 0, 2>(a, st);
    if (f8_q) return abf ? launch_fwd<128, 16, false, true, 0, 3>(a, st)
                         : launch_fwd<128, 16, false, false, 0, 3>(a, st);
    return abf ? launch_fwd<128, 16, false, true, 0, 2>(a, st)
               : launch_fwd<128, 16, false, false, 0, 2>(a, st);
  }
  if (s.nk == 16) {
    // PCA_D128_FUSED=0: the LDS-resident-weight kernel (k_mab1_fwd) instead of the wave = head one
    constexpr bool fused128 = true;
    // (d -> d blocks only: at layer 1 - two or three input columns, no X tile to stream - the
    //  wave = head kernel measured 17.6 us against 17.0)
    if (abf && fused128 && !small)
      return isab1_fwd128_fused(X, s.dq, WqB, p.wq, p.bq, v.KpP, v.Vt, WoP, p.bo,
                                reinterpret_cast<__bf16*>(Y), a.QpS, a.OS, a.mask, s.B, s.nq, st);
    if (abf) return small ? launch_fwd<128, 16, true, true>(a, st)
                          : launch_fwd<128, 16, false, true>(a, st);
    return small ? launch_fwd<128, 16, true, false>(a, st)
                 : launch_fwd<128, 16, false, false>(a, st);
  }
  if (d == 256) {          // Q phase + O phase: O meets in the saved / scratch block
    if (small && !abf) return launch_fwd<256, 32, true, false>(a, st);
    a.OS = v.OS;
    const bool fused256 = fused256_on();
    if (f8_fused)
      return isab1_fwd256_fused(X, s.dq, WqB, p.wq, p.bq, v.KpP, v.Vt, WoP, p.bo,
                                reinterpret_cast<__bf16*>(Y), a.QpS, training ? v.OS : nullptr,
                                a.mask, s.B, s.nq, st, inv_o);
    if (abf && !f8 && fused256) {
      // one launch: wave = head, both weight slices in registers (d256_fused.hip)
      if (img != nullptr) PCA_TRY(prep_weight(p.wo, WoP, d, d, 0, st));   // natural image for this kernel
      return isab1_fwd256_fused(X, s.dq, WqB, p.wq, p.bq, v.KpP, v.Vt, WoP, p.bo,
                                reinterpret_cast<__bf16*>(Y), a.QpS, training ? v.OS : nullptr,
                                a.mask, s.B, s.nq, st);
    }
    if (abf) {
      // bf16 activations: the O phase is the row-GEMM kernel of d256_bf16.hip (full-line I/O);
      // layer 1 runs its Q phase without any weight image (8 waves, 32 KiB of LDS)
      if (small) PCA_TRY((launch_fwd<256, 32, true, true, 1>(a, st)));
      else if (f8 && f8_q) PCA_TRY((launch_fwd<256, 32, false, true, 1, 1>(a, st)));
      else PCA_TRY((launch_fwd<256, 32, false, true, 1>(a, st)));
      const double pts = (double)s.B * s.nq;
      ProfScope ps(PCA_K_MAB1_FWD, st, 2.0 * pts * d * d, pts * 4.0 * d);
      const int rc = f8 ? rowgemm256_fwd_o_f8(v.OS, WoP, invs + 1, p.bo, reinterpret_cast<__bf16*>(Y),
                                              a.mask, s.B, s.nq, st)
                        : rowgemm256_fwd_o(v.OS, WoP, p.bo, reinterpret_cast<__bf16*>(Y), a.mask,
                                           s.B, s.nq, st);
      ps.end();
      return rc;
    }
    PCA_TRY((launch_fwd<256, 32, false, false, 1>(a, st)));
    return launch_fwd<256, 32, false, false, 2>(a, st);
  }
  PCA_REQUIRE(!abf, "mab1_bf16_fwd: bf16 activations need m = 16 (d = 128)");
  return small ? launch_fwd<128, 32, true, false>(a, st) : launch_fwd<128, 32, false, false>(a, st);
}

}  // namespace pca
