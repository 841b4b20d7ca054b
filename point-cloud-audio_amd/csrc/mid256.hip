// Per-set "mid" stage of an ISAB at d = 256 / m = 32 (set_transformer-master/modules.py:51-53), the
// counterpart of mid_bf16.hip's k_mid_fwd at d = 128: everything between the few-queries block's
// attention and the many-queries block's first point-sized kernel, on the [m = 32, d = 256] tensors
// of one set, in ONE launch:
//     Z = O Wo^T + bo ; H = O + relu(Z)                (mab0 epilogue, modules.py:31)
//     Kp = H Wk^T + bk ; Vp = H Wv^T + bv              (mab1 projections, modules.py:21)
//     the four bf16 images k_isab1_fwd256_ab / k_attn1_bwd3 read (KpP, VpP, Kt, Vt)
// Round 2 ran this as k_gemm_bf16<hi+lo> + k_add_relu + 2 x k_gemm_bf16 + k_kv_images: five launches of
// ~5-11 us each on [B m] = 4096 rows (43 us per ISAB at configs[3]).
//
// One workgroup per set, 8 waves; wave j owns output features 32 j .. 32 j + 31 of each product and
// streams its [32 x 256] fp32 weight slices straight from L2 (every fragment is needed exactly once
// per set).  fc_o runs with hi + lo bf16 operand pairs (three MFMAs per fragment pair, as
// k_gemm_bf16<.., HL>: fp32-level Z, the exact ReLU mask); fc_k / fc_v with single bf16 operands
// like the GEMM they replace.  Activations cross the waves through bf16 LDS tiles ([32][256],
// 16-byte chunks XOR-swizzled by row).
#include "d256_bf16.hpp"

#include <mutex>

namespace pca {

namespace {

constexpr int D = 256, MI = 32, ROWB = D * 2, TILEB = MI * ROWB;

struct Mid256Args {
  const float* O;                 // [B][32][256]
  const float *Wo, *bo, *Wk, *bk, *Wv, *bv;
  float *Z, *H;                   // [B][32][256]
  __bf16 *KpP, *VpP, *Kt, *Vt;    // [B][32][256] x 2, [B][256][32] x 2
};

__device__ __forceinline__ void split8(const float4 lo4, const float4 hi4, bf16x8& h, bf16x8& l) {
  const float x[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    h[k] = (__bf16)x[k];
    l[k] = (__bf16)(x[k] - (float)h[k]);
  }
}

__global__ __launch_bounds__(512) void k_mid256_fwd(const Mid256Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sOh = smem;               // O, hi parts   bf16 [32][256]
  char* sOl = smem + TILEB;       // O, lo parts
  char* sH = smem + 2 * TILEB;    // H
  char* sKp = smem + 3 * TILEB;   // Kp
  char* sVp = smem + 4 * TILEB;   // Vp
  const int tid = threadIdx.x, lane = tid & 63;
  const int j = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int b = blockIdx.x;
  const float* Ob = a.O + (int64_t)b * MI * D;

  // ---- O -> hi / lo bf16 tiles (thread: row tid / 16, columns 16 (tid % 16) ..) ----
  {
    const int row = tid >> 4, cg = tid & 15;
    const float4* src = reinterpret_cast<const float4*>(Ob + row * D + 16 * cg);
    const float4 x0 = src[0], x1 = src[1], x2 = src[2], x3 = src[3];
    bf16x8 h0, l0, h1, l1;
    split8(x0, x1, h0, l0);
    split8(x2, x3, h1, l1);
    *reinterpret_cast<bf16x8*>(sOh + swz(row, 2 * cg, ROWB)) = h0;
    *reinterpret_cast<bf16x8*>(sOh + swz(row, 2 * cg + 1, ROWB)) = h1;
    *reinterpret_cast<bf16x8*>(sOl + swz(row, 2 * cg, ROWB)) = l0;
    *reinterpret_cast<bf16x8*>(sOl + swz(row, 2 * cg + 1, ROWB)) = l1;
  }
  int oB[4], oD[2];
#pragma unroll
  for (int k = 0; k < 4; ++k) oB[k] = swz(r, 4 * k + g, ROWB);
#pragma unroll
  for (int t = 0; t < 2; ++t) oD[t] = swz(r, 4 * j + 2 * t + (g >> 1), ROWB) + 8 * (g & 1);
  // the residual O of this lane's accumulator elements (row 16 nb + r, features 32 j + 16 t + 4 g ..)
  float4 ores[2][2];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int t = 0; t < 2; ++t)
      ores[nb][t] = *reinterpret_cast<const float4*>(Ob + (16 * nb + r) * D + 32 * j + 16 * t + 4 * g);
  __syncthreads();

  // ---- Z_j^T = Wo_j . O^T + bo with hi + lo operand pairs ----
  f32x4 acc[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const float4 b4 = *reinterpret_cast<const float4*>(a.bo + 32 * j + 16 * t + 4 * g);
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) acc[t][nb] = f32x4{b4.x, b4.y, b4.z, b4.w};
  }
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    float4 w[4][2][2];                       // [k-step][feature tile][8 floats]: 64 registers in flight
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const float4* p = reinterpret_cast<const float4*>(a.Wo + (int64_t)(32 * j + 16 * t + r) * D +
                                                          32 * (4 * half + s4) + 8 * g);
        w[s4][t][0] = p[0];
        w[s4][t][1] = p[1];
      }
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const int s = 4 * half + s4;
      bf16x8 bh[2], bl[2];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        bh[nb] = *reinterpret_cast<const bf16x8*>(sOh + oB[s & 3] + 256 * (s >> 2) + 8192 * nb);
        bl[nb] = *reinterpret_cast<const bf16x8*>(sOl + oB[s & 3] + 256 * (s >> 2) + 8192 * nb);
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        bf16x8 ah, al;
        split8(w[s4][t][0], w[s4][t][1], ah, al);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          acc[t][nb] = mfma32(ah, bh[nb], acc[t][nb]);
          acc[t][nb] = mfma32(ah, bl[nb], acc[t][nb]);
          acc[t][nb] = mfma32(al, bh[nb], acc[t][nb]);
        }
      }
    }
  }
  // Z, H = O + relu(Z): global fp32 (saved / block output) and the bf16 tile of H
  float* Zb = a.Z + (int64_t)b * MI * D;
  float* Hb = a.H + (int64_t)b * MI * D;
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int o = (16 * nb + r) * D + 32 * j + 16 * t + 4 * g;
      const f32x4 z = acc[t][nb];
      *reinterpret_cast<float4*>(Zb + o) = float4{z[0], z[1], z[2], z[3]};
      f32x4 hv;
      hv[0] = ores[nb][t].x + fmaxf(z[0], 0.f); hv[1] = ores[nb][t].y + fmaxf(z[1], 0.f);
      hv[2] = ores[nb][t].z + fmaxf(z[2], 0.f); hv[3] = ores[nb][t].w + fmaxf(z[3], 0.f);
      *reinterpret_cast<float4*>(Hb + o) = float4{hv[0], hv[1], hv[2], hv[3]};
      *reinterpret_cast<bf16x4*>(sH + oD[t] + 8192 * nb) = pack4(hv);
    }
  __syncthreads();

  // ---- Kp_j^T = Wk_j . H^T + bk ; Vp_j^T = Wv_j . H^T + bv (bf16 operands) ----
#pragma unroll
  for (int kv = 0; kv < 2; ++kv) {
    const float* W = kv == 0 ? a.Wk : a.Wv;
    const float* bias = kv == 0 ? a.bk : a.bv;
    char* dst = kv == 0 ? sKp : sVp;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const float4 b4 = *reinterpret_cast<const float4*>(bias + 32 * j + 16 * t + 4 * g);
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) acc[t][nb] = f32x4{b4.x, b4.y, b4.z, b4.w};
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      float4 w[4][2][2];
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const float4* p = reinterpret_cast<const float4*>(W + (int64_t)(32 * j + 16 * t + r) * D +
                                                            32 * (4 * half + s4) + 8 * g);
          w[s4][t][0] = p[0];
          w[s4][t][1] = p[1];
        }
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const int s = 4 * half + s4;
        bf16x8 bh[2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
          bh[nb] = *reinterpret_cast<const bf16x8*>(sH + oB[s & 3] + 256 * (s >> 2) + 8192 * nb);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          bf16x8 aw;
          aw[0] = (__bf16)w[s4][t][0].x; aw[1] = (__bf16)w[s4][t][0].y;
          aw[2] = (__bf16)w[s4][t][0].z; aw[3] = (__bf16)w[s4][t][0].w;
          aw[4] = (__bf16)w[s4][t][1].x; aw[5] = (__bf16)w[s4][t][1].y;
          aw[6] = (__bf16)w[s4][t][1].z; aw[7] = (__bf16)w[s4][t][1].w;
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) acc[t][nb] = mfma32(aw, bh[nb], acc[t][nb]);
        }
      }
    }
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int t = 0; t < 2; ++t)
        *reinterpret_cast<bf16x4*>(dst + oD[t] + 8192 * nb) = pack4(acc[t][nb]);
  }
  __syncthreads();

  // ---- the four images, as k_kv_images writes them (thread: feature tid % 256 of K or V) ----
  {
    const int f = tid & 255, isv = tid >> 8;
    const char* tile = isv ? sVp : sKp;
    __bf16 v[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
      v[i] = *reinterpret_cast<const __bf16*>(tile + swz(i, f >> 3, ROWB) + 2 * (f & 7));
    // feature fo = 16 t + 4 g + e of its head sits at k-slot 8 g + 4 t + e (the inverse of perm32)
    const int jb = f & ~31, fo = f & 31;
    const int pos = 8 * ((fo >> 2) & 3) + 4 * (fo >> 4) + (fo & 3);
    __bf16* PP = (isv ? a.VpP : a.KpP) + (int64_t)b * MI * D + jb + pos;
#pragma unroll
    for (int i = 0; i < MI; ++i) PP[i * D] = v[i];
    bf16x8 tt[4];
#pragma unroll
    for (int kp = 0; kp < MI; ++kp) tt[kp >> 3][kp & 7] = v[perm32(kp)];
    bf16x8* TT = reinterpret_cast<bf16x8*>((isv ? a.Vt : a.Kt) + ((int64_t)b * D + f) * MI);
#pragma unroll
    for (int q = 0; q < 4; ++q) TT[q] = tt[q];
  }
}

}  // namespace

int mid256_fwd(const float* O, const float* Wo, const float* bo, const float* Wk, const float* bk,
               const float* Wv, const float* bv, float* Z, float* H, __bf16* KpP, __bf16* VpP,
               __bf16* Kt, __bf16* Vt, int B, hipStream_t st) {
  Mid256Args a{O, Wo, bo, Wk, bk, Wv, bv, Z, H, KpP, VpP, Kt, Vt};
  static std::once_flag once;
  std::call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_mid256_fwd),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  hipLaunchKernelGGL(k_mid256_fwd, dim3(B), dim3(512), (size_t)5 * TILEB, st, a);
  return check_launch("k_mid256_fwd");
}

}  // namespace pca
