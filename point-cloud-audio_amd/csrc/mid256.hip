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
// One workgroup per set, 16 waves; wave j owns output features 16 j .. 16 j + 15 of each product and
// streams its [16 x 256] fp32 weight slices straight from L2 (every fragment is needed exactly once
// per set) in six half-K batches of 32 registers, each requested one batch ahead of its use, across
// the phase boundaries too (the loads depend on nothing the phases compute).  fc_o runs with hi + lo
// bf16 operand pairs (three MFMAs per fragment pair, as k_gemm_bf16<.., HL>: fp32-level Z, the exact
// ReLU mask); fc_k / fc_v with single bf16 operands like the GEMM they replace.  Activations cross
// the waves through bf16 LDS tiles ([32][256], 16-byte chunks XOR-swizzled by row).
// (First version: 8 waves x 32 features, a batch loaded and then consumed: 32.1 us per launch at
//  B = 128; this one: see DESIGN.md 4.5.)
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
__device__ __forceinline__ bf16x8 round8(const float4 lo4, const float4 hi4) {
  bf16x8 h;
  h[0] = (__bf16)lo4.x; h[1] = (__bf16)lo4.y; h[2] = (__bf16)lo4.z; h[3] = (__bf16)lo4.w;
  h[4] = (__bf16)hi4.x; h[5] = (__bf16)hi4.y; h[6] = (__bf16)hi4.z; h[7] = (__bf16)hi4.w;
  return h;
}

struct WBatch { float4 w[4][2]; };          // four k-steps of this lane's weight row: 32 registers

__global__ __launch_bounds__(1024) void k_mid256_fwd(const Mid256Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sOh = smem;               // O, hi parts   bf16 [32][256]
  char* sOl = smem + TILEB;       // O, lo parts
  char* sH = smem + 2 * TILEB;    // H
  char* sKp = smem + 3 * TILEB;   // Kp
  char* sVp = smem + 4 * TILEB;   // Vp
  const int tid = threadIdx.x, lane = tid & 63;
  const int j = __builtin_amdgcn_readfirstlane(tid >> 6);     // features 16 j ..
  const int r = lane & 15, g = lane >> 4;
  const int b = blockIdx.x;
  const float* Ob = a.O + (int64_t)b * MI * D;

  // weight row 16 j + r, k = 32 (4 half + s4) + 8 g .. + 7
  auto load_batch = [&](const float* W, int half) {
    WBatch wb;
    const float4* p = reinterpret_cast<const float4*>(W + (16 * j + r) * D + 128 * half + 8 * g);
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      wb.w[s4][0] = p[8 * s4];
      wb.w[s4][1] = p[8 * s4 + 1];
    }
    return wb;
  };
  WBatch w0 = load_batch(a.Wo, 0);

  // ---- O -> hi / lo bf16 tiles (thread: row tid / 32, columns 8 (tid % 32) ..) ----
  {
    const int row = tid >> 5, ch = tid & 31;
    const float4* src = reinterpret_cast<const float4*>(Ob + row * D + 8 * ch);
    bf16x8 h0, l0;
    split8(src[0], src[1], h0, l0);
    *reinterpret_cast<bf16x8*>(sOh + swz(row, ch, ROWB)) = h0;
    *reinterpret_cast<bf16x8*>(sOl + swz(row, ch, ROWB)) = l0;
  }
  int oB[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) oB[k] = swz(r, 4 * k + g, ROWB);
  const int oD = swz(r, 2 * j + (g >> 1), ROWB) + 8 * (g & 1);
  // the residual O of this lane's accumulator elements (row 16 nb + r, features 16 j + 4 g ..)
  float4 ores[2];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
    ores[nb] = *reinterpret_cast<const float4*>(Ob + (16 * nb + r) * D + 16 * j + 4 * g);
  f32x4 acc[2];
  auto init_acc = [&](const float* bias) {
    const float4 b4 = *reinterpret_cast<const float4*>(bias + 16 * j + 4 * g);
    acc[0] = acc[1] = f32x4{b4.x, b4.y, b4.z, b4.w};
  };
  // half a K range of Z_j^T = Wo_j . O^T with hi + lo operand pairs
  auto gemm_hl = [&](const WBatch& wb, int half) {
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const int s = 4 * half + s4;
      bf16x8 ah, al;
      split8(wb.w[s4][0], wb.w[s4][1], ah, al);
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const bf16x8 bh = *reinterpret_cast<const bf16x8*>(sOh + oB[s & 3] + 256 * (s >> 2) + 8192 * nb);
        const bf16x8 bl = *reinterpret_cast<const bf16x8*>(sOl + oB[s & 3] + 256 * (s >> 2) + 8192 * nb);
        acc[nb] = mfma32(ah, bh, acc[nb]);
        acc[nb] = mfma32(ah, bl, acc[nb]);
        acc[nb] = mfma32(al, bh, acc[nb]);
      }
    }
  };
  // half a K range of (W_j . H^T) with bf16 operands
  auto gemm_h = [&](const WBatch& wb, int half) {
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const int s = 4 * half + s4;
      const bf16x8 aw = round8(wb.w[s4][0], wb.w[s4][1]);
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const bf16x8 bh = *reinterpret_cast<const bf16x8*>(sH + oB[s & 3] + 256 * (s >> 2) + 8192 * nb);
        acc[nb] = mfma32(aw, bh, acc[nb]);
      }
    }
  };
  init_acc(a.bo);
  __syncthreads();

  // ---- Z_j^T = Wo_j . O^T + bo ; H = O + relu(Z) ----
  WBatch w1 = load_batch(a.Wo, 1);
  gemm_hl(w0, 0);
  w0 = load_batch(a.Wk, 0);
  gemm_hl(w1, 1);
  {
    float* Zb = a.Z + (int64_t)b * MI * D;
    float* Hb = a.H + (int64_t)b * MI * D;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const int o = (16 * nb + r) * D + 16 * j + 4 * g;
      const f32x4 z = acc[nb];
      *reinterpret_cast<float4*>(Zb + o) = float4{z[0], z[1], z[2], z[3]};
      f32x4 hv;
      hv[0] = ores[nb].x + fmaxf(z[0], 0.f); hv[1] = ores[nb].y + fmaxf(z[1], 0.f);
      hv[2] = ores[nb].z + fmaxf(z[2], 0.f); hv[3] = ores[nb].w + fmaxf(z[3], 0.f);
      *reinterpret_cast<float4*>(Hb + o) = float4{hv[0], hv[1], hv[2], hv[3]};
      *reinterpret_cast<bf16x4*>(sH + oD + 8192 * nb) = pack4(hv);
    }
  }
  init_acc(a.bk);
  __syncthreads();

  // ---- Kp_j^T = Wk_j . H^T + bk ; Vp_j^T = Wv_j . H^T + bv ----
  w1 = load_batch(a.Wk, 1);
  gemm_h(w0, 0);
  w0 = load_batch(a.Wv, 0);
  gemm_h(w1, 1);
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) *reinterpret_cast<bf16x4*>(sKp + oD + 8192 * nb) = pack4(acc[nb]);
  init_acc(a.bv);
  w1 = load_batch(a.Wv, 1);
  gemm_h(w0, 0);
  gemm_h(w1, 1);
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) *reinterpret_cast<bf16x4*>(sVp + oD + 8192 * nb) = pack4(acc[nb]);
  __syncthreads();

  // ---- the four images, as k_kv_images writes them (thread: feature tid % 256 of K or V, 16 keys) ----
  {
    const int f = tid & 255, isv = (tid >> 8) & 1, kh = tid >> 9;
    const char* tile = isv ? sVp : sKp;
    // feature fo = 16 t + 4 g + e of its head sits at k-slot 8 g + 4 t + e (the inverse of perm32)
    const int jb = f & ~31, fo = f & 31;
    const int pos = 8 * ((fo >> 2) & 3) + 4 * (fo >> 4) + (fo & 3);
    __bf16* PP = (isv ? a.VpP : a.KpP) + (int64_t)b * MI * D + jb + pos;
#pragma unroll
    for (int i = 0; i < MI / 2; ++i) {
      const int key = 16 * kh + i;
      PP[key * D] = *reinterpret_cast<const __bf16*>(tile + swz(key, f >> 3, ROWB) + 2 * (f & 7));
    }
    bf16x8* TT = reinterpret_cast<bf16x8*>((isv ? a.Vt : a.Kt) + ((int64_t)b * D + f) * MI);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      bf16x8 tt;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int key = perm32(8 * (2 * kh + q) + e);
        tt[e] = *reinterpret_cast<const __bf16*>(tile + swz(key, f >> 3, ROWB) + 2 * (f & 7));
      }
      TT[2 * kh + q] = tt;
    }
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
  hipLaunchKernelGGL(k_mid256_fwd, dim3(B), dim3(1024), (size_t)5 * TILEB, st, a);
  return check_launch("k_mid256_fwd");
}

}  // namespace pca
