// Set-resident forward of the d = 128 / 4 heads / m = 16 inducing points / k = 1 Set Transformer
// (Code/models.py:34-44: ISAB, ISAB, PMA; set_transformer-master/modules.py:19-33, 51-53, 62-63),
// BASELINE configs[1] (N = 512, din = 2, B = 128).
//
// A PAIR of 1024-thread workgroups carries ONE set through
//     mab0(I1, X) -> mid -> mab1(X, H1) -> mab0(I2, Y1) -> mid -> mab1(Y1, H2) -> PMA attention partials
// each workgroup with ITS HALF of the set's [N, 128] bf16 activation tensor resident in LDS from the
// moment layer 1 produces it until the PMA has read it (B = 128 sets -> 256 workgroups = one per CU).
// The per-block kernels this replaces (k_mab0_attn_small, k_mid_fwd, k_mab1_fwd, k_mab0_attn_h4,
// k_mid_fwd, k_isab1_fwd_t, k_mab0_attn) each ran ONE 32-point unit per wave with all 2048 waves of
// the chip in the same phase at the same time: every launch paid its weight images, the input burst,
// the dependent MFMA chain and the store drain in lock-step, plus ~5 us of launch and drain - seven
// times per forward.  Here a phase boundary is a workgroup barrier, the activations between blocks
// never leave the CU (they go to HBM only as the tensors the backward reads), and the only
// cross-workgroup traffic of a set is the (m, l, T) partial of its two few-queries attentions, handed
// to the partner workgroup through write-through (sc1) stores, one flag and sc1 loads
// (cdna_hip_programming.md Guideline 16, R1); both halves then merge the same two partials in the same
// order and run the per-set "mid" stage redundantly, so nothing else has to cross.
//
// Wave w = (quad q = w >> 2, head j = w & 3): quad q owns a quarter of the workgroup's points, wave j of
// a quad the head j (features 32 j .. 32 j + 31) - the WAVE = HEAD, WEIGHTS IN REGISTERS form of
// k_isab1_fwd_t (d128_fused.hip) and k_mab0_attn_h4 (mab0_bf16.hip), whose arithmetic and saved layouts
// this kernel keeps (same MFMA operand orders, same K-permuted images), so that the existing backward
// kernels read what it saves.
//
// LDS: sY [N/2][256 B] (rows XOR-swizzled as tr_off, so that both the row-major B-operand reads and the
// transposing ds_read_tr16_b64 reads are conflict-free), sO [N/2][256 B] (the O tile of a many-queries
// block: fc_o contracts over all heads' features; between blocks: merge buffers), 32 KiB of images.
//
// Roofline unit: MACs_fwd / set = N (3 din d + 7 d^2 + 8 m d + 2 k d) + 6 m d^2 (the
// PMA epilogue and the classifier run in k_pma_head1); algorithmic bytes 4 N din + 4 (2 N d) per set.
#include "set128.hpp"

#include <math.h>

#include <mutex>

namespace pca {

namespace {

constexpr int D = 128, MQ = 16, ROWB = 256, NT = 1024;

// byte offset of 16-byte chunk ch of row `row` of a [rows][256 B] image (the layout of the X tiles of
// k_mab0_attn_h4: the XOR term depends on row & 15 only, so any 16-aligned row block is an image)
__device__ __forceinline__ int y_off(int row, int ch) {
  return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
// B operand [k = point (k-slot order of pack8 of two score tiles)][col = feature 16 t + (lane & 15)] of
// the 32-row image `img`, through the transposing LDS read
__device__ __forceinline__ bf16x8 y_tr_frag(const char* img, int t, int lane) {
  const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
  const int a0 = y_off(4 * g + q, 2 * t + (p >> 1)) + 8 * (p & 1);
  const int a1 = y_off(16 + 4 * g + q, 2 * t + (p >> 1)) + 8 * (p & 1);
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + a0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + a1));
  const bf16x4 l4 = __builtin_bit_cast(bf16x4, lo), h4 = __builtin_bit_cast(bf16x4, hi);
  bf16x8 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) { r[e] = l4[e]; r[4 + e] = h4[e]; }
  return r;
}
__device__ __forceinline__ bf16x8 gload8(const __bf16* p) {
  return *reinterpret_cast<const bf16x8*>(p);
}
// accumulator tile (rows = features 16 t + 4 g + e, col = query r) -> [query][feature] bf16 image
__device__ __forceinline__ void put_tile(char* img, int t, int r, int g, f32x4 v) {
  *reinterpret_cast<bf16x4*>(img + swz(r, 2 * t + (g >> 1), ROWB) + 8 * (g & 1)) = pack4(v);
}
// workgroup barrier that orders LDS traffic only: global stores in flight (saved tensors) are not
// waited for, as __syncthreads()'s vmcnt(0) would
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
// max as ONE instruction (v_med3_f32 with +inf; fmaxf costs a canonicalising v_max x, x in front).
// A builtin, NOT inline asm: hipcc's hazard recognizer does not look inside asm, and a VALU
// instruction reading an MFMA result needs wait states it would not insert (the first version of this
// kernel read the accumulators of the mid stage one instruction after the MFMA - and got the old value)
__device__ __forceinline__ float maxf(float x, float y) { return __builtin_amdgcn_fmed3f(x, y, INFINITY); }
__device__ __forceinline__ float max0(float x) { return __builtin_amdgcn_fmed3f(x, 0.f, INFINITY); }
// bits = 2 bits + (z > 0)
__device__ __forceinline__ uint32_t push_gt0(uint32_t bits, float z) {
  return (bits << 1) | (z > 0.f ? 1u : 0u);
}

// ---- cross-workgroup hand-off (cdna_hip_programming.md Guideline 16, form R1): payload by sc1
// (write-through) stores, every storing wave drains, workgroup barrier, ONE lane stores the flag; the
// consumer polls the flag (bounded), then reads the payload with sc1 loads only -------------------
typedef __attribute__((address_space(1))) uint32_t gu32;
__device__ __forceinline__ void st16_sc1(float* p, float4 v4) {
  const f32x4 v = {v4.x, v4.y, v4.z, v4.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ f32x4 ld16_sc1(const float* p) {       // caller: s_waitcnt vmcnt(0) before use
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void drain_vm() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// the wait as part of the data flow of a value loaded by ld16_sc1: hipcc does not know that the asm load
// is still in flight and would otherwise be free to schedule uses of `v` above a bare s_waitcnt
__device__ __forceinline__ void landed(f32x4& v) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(v)::"memory"); }
__device__ __forceinline__ void flag_store(uint32_t* f, uint32_t v) {
  __hip_atomic_store((gu32*)(f), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one wave polls; a partner that never arrives (it would mean the pair is not co-resident) ends the
// spin after ~1 s, counts it in tmo[0] and lets the kernel finish with whatever the slot holds
__device__ __forceinline__ void flag_wait(uint32_t* f, uint32_t want, uint32_t* tmo) {
  for (unsigned spins = 0; spins < (1u << 20); ++spins) {
    if (__hip_atomic_load((gu32*)(f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == want) return;
    __builtin_amdgcn_s_sleep(4);
  }
  if ((threadIdx.x & 63) == 0) atomicAdd(tmo, 1u);
}

// -DPCA_SET_STAMPS (scripts/experiments/set_stamps.sh): wall-clock stamps (10 ns units) of workgroup 0,
// lane 0 of every wave, at the phase boundaries
#ifdef PCA_SET_STAMPS
__device__ unsigned long long g_set_stamps[16 * 32];
#define STAMP(i)                                                                         \
  do {                                                                                   \
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0)                                      \
      g_set_stamps[(threadIdx.x >> 6) * 32 + (i)] = wall_clock64();                      \
  } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

struct Ctx {
  int b, half, N, NH, tid, lane, wave, q, j, r, g;
  int UPQ;          // 32-point units per quad (N / 256)
  int qn0;          // first LOCAL row (of sY / sO) of this wave's quad
  int64_t row0;     // global row (b N + half N/2) of local row 0
};

// ---------------------------------------------------------------------------------------------
// per-set stage between the two blocks of an ISAB (k_mid_fwd, mid_bf16.hip, on 16 waves), run by BOTH
// workgroups of the pair on identical inputs (half 0 writes the saved copies):
//   A  O = Qp + T_h Wv_h^T + bv        (modules.py:29, reassociated: SURVEY 8d)     waves 0-7, tile t = wave
//   B  H = O + relu(O Wo^T + bo)       (modules.py:31)                              waves 0-7
//   C  Kp = H Wk^T + bk, Vp = H Wv^T + bv of the many-queries block (modules.py:21) all waves:
//      wave = (tile tc = wave >> 1, which = wave & 1); the two images the forward reads stay in LDS
//      (sKp: K-permuted [key][feature], sVt: [feature][key]), all four go to memory for the backward
// Every weight fragment a wave needs is requested by mid_prefetch() a phase ahead.
// ---------------------------------------------------------------------------------------------
struct MidPre {
  bf16x8 wa[4], wb[4], wc[4];
  float wvf[4][4];                 // SMALL: Wv0f[feature 16 t + 4 g + e][c]
  float4 qp, bv0, bo0, bkv;
  float bkvc;
};

template <bool SMALL>
__device__ __forceinline__ void mid_prefetch(const Set128Layer& L, const Ctx& c, int dk, MidPre& P) {
  const int r = c.r, g = c.g;
  {
    // (waves 8-15 load the fragments of tile wave - 8 and never use them: under `if (wave < 8)` hipcc
    //  spilled every value of this block right behind its load, each with an s_waitcnt vmcnt(0))
    const int t = c.wave & 7;
    P.qp = *reinterpret_cast<const float4*>(L.Qp0 + r * D + 16 * t + 4 * g);
    P.bv0 = *reinterpret_cast<const float4*>(L.bv0 + 16 * t + 4 * g);
    P.bo0 = *reinterpret_cast<const float4*>(L.bo0 + 16 * t + 4 * g);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (!SMALL) P.wa[ks] = gload8(L.Wv0 + (int64_t)(16 * t + r) * D + 32 * ks + 8 * g);
      P.wb[ks] = gload8(L.Wo0 + (int64_t)(16 * t + r) * D + 32 * ks + 8 * g);
    }
    if (SMALL) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
          P.wvf[e][cc] = cc < dk ? L.Wv0f[(16 * t + 4 * g + e) * dk + cc] : 0.f;
    }
  }
  {
    const int tc = c.wave >> 1, which = c.wave & 1;
    const __bf16* W = which ? L.Wv1 : L.Wk1;
    const float* bias = which ? L.bv1 : L.bk1;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) P.wc[ks] = gload8(W + (int64_t)(16 * tc + r) * D + 32 * ks + 8 * g);
    P.bkv = *reinterpret_cast<const float4*>(bias + 16 * tc + 4 * g);
    P.bkvc = bias[16 * tc + r];
  }
}

// sT: bf16 image [64][256 B] of the merged T (swz rows; !SMALL) / sTf: fp32 [64][4] (SMALL);
// sA, sH: [16][256 B] bf16 images of O and H.  Ends with a barrier.
template <bool SMALL>
__device__ __forceinline__ void mid_stage(const Set128Layer& L, const Ctx& c, const MidPre& P,
                                          const char* sT, const float* sTf, char* sA, char* sH,
                                          char* sKp, char* sVt) {
  const int r = c.r, g = c.g, b = c.b;
  const bool save = c.half == 0;
  f32x4 o = {0.f, 0.f, 0.f, 0.f};
  if (c.wave < 8) {                                   // ---- A
    const int t = c.wave, hh = t >> 1;
    o = f32x4{P.qp.x + P.bv0.x, P.qp.y + P.bv0.y, P.qp.z + P.bv0.z, P.qp.w + P.bv0.w};
    if (SMALL) {
      const float4 t4 = *reinterpret_cast<const float4*>(sTf + (16 * hh + r) * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        o[e] += t4.x * P.wvf[e][0] + t4.y * P.wvf[e][1] + t4.z * P.wvf[e][2] + t4.w * P.wvf[e][3];
    } else {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        o = mfma32(P.wa[ks], *reinterpret_cast<const bf16x8*>(sT + swz(16 * hh + r, 4 * ks + g, ROWB)), o);
    }
    put_tile(sA, t, r, g, o);
    if (save)
      *reinterpret_cast<float4*>(L.O0 + ((int64_t)b * MQ + r) * D + 16 * t + 4 * g) =
          float4{o[0], o[1], o[2], o[3]};
  }
  lds_barrier();
  if (c.wave < 8) {                                   // ---- B
    const int t = c.wave;
    f32x4 z = f32x4{P.bo0.x, P.bo0.y, P.bo0.z, P.bo0.w};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      z = mfma32(P.wb[ks], *reinterpret_cast<const bf16x8*>(sA + swz(r, 4 * ks + g, ROWB)), z);
    f32x4 hq;
#pragma unroll
    for (int e = 0; e < 4; ++e) hq[e] = o[e] + max0(z[e]);
    if (save) {
      const int64_t off = ((int64_t)b * MQ + r) * D + 16 * t + 4 * g;
      *reinterpret_cast<float4*>(L.Z0 + off) = float4{z[0], z[1], z[2], z[3]};
      *reinterpret_cast<float4*>(L.H + off) = float4{hq[0], hq[1], hq[2], hq[3]};
    }
    put_tile(sH, t, r, g, hq);
  }
  lds_barrier();
  {                                                   // ---- C
    const int tc = c.wave >> 1, which = c.wave & 1;
    f32x4 fr = f32x4{P.bkv.x, P.bkv.y, P.bkv.z, P.bkv.w};     // rows = features, col = key
    f32x4 kr = f32x4{P.bkvc, P.bkvc, P.bkvc, P.bkvc};         // rows = keys,     col = feature
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x8 hf = *reinterpret_cast<const bf16x8*>(sH + swz(r, 4 * ks + g, ROWB));
      fr = mfma32(P.wc[ks], hf, fr);
      kr = mfma32(hf, P.wc[ks], kr);
    }
    const bf16x4 krp = pack4(kr), frp = pack4(fr);
    // [feature][key] image: keys 4g .. 4g+3 of feature 16 tc + r (8 contiguous bytes);
    // K-permuted [key][feature] image: features 32 (tc >> 1) + perm32(8 g + 4 (tc & 1) + e) of key r -
    // half of the lane's 16 bytes (k_mid_fwd writes pack8 of the head's two tiles)
    const int o_t = (16 * tc + r) * MQ + 4 * g;
    const int o_p = r * D + 32 * (tc >> 1) + 8 * g + 4 * (tc & 1);
    if (which) *reinterpret_cast<bf16x4*>(sVt + o_t * 2) = krp;
    else *reinterpret_cast<bf16x4*>(sKp + o_p * 2) = frp;
    if (save) {
      __bf16* PP = which ? L.VpP : L.KpP;
      __bf16* TT = which ? L.Vt : L.Kt;
      *reinterpret_cast<bf16x4*>(TT + (int64_t)b * D * MQ + o_t) = krp;
      *reinterpret_cast<bf16x4*>(PP + (int64_t)b * MQ * D + o_p) = frp;
    }
  }
  lds_barrier();
}

// ---------------------------------------------------------------------------------------------
// many-queries block mab1(X, H) (modules.py:19-33) on the workgroup's rows; wave = (quad, head):
//   Qp_h = fc_q(x) ; A = softmax(Qp_h Kp_h^T / sqrt d) ; O_h = Qp_h + A Vp_h -> own slice of sO
//   barrier ; Z_h = fc_o(O) ; Y_h = O_h + relu(Z_h) -> own slice of sY ; barrier
//   coalesced stores of the saved O (from sO) and of Y (from sY)
// SMALL: layer 1 (dq = din <= 4: fc_q on the vector ALU from the points in sX, nothing in sY yet)
// ---------------------------------------------------------------------------------------------
template <bool SMALL, int DQ, int UPQ>
__device__ __forceinline__ void mab1_phase(const Set128Layer& L, const Ctx& c, char* sY, char* sO,
                                           const float* sX, const char* sKp, const char* sVt,
                                           float scale_log2e, int stamp0, bf16x8 (&wa)[4][2]) {
  (void)stamp0;
  const int r = c.r, g = c.g, j = c.j;
  constexpr int KS = 4;
  // this head's weight slices as A operands [row = feature 32 j + 16 t + r][k-slots 32 s + 8 g ..]:
  // wa arrives holding fc_q's (requested by the caller one stage ahead: mab1_load_wq)
  float wqs[2][4][SMALL ? DQ : 1];
  if (SMALL) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int cc = 0; cc < DQ; ++cc)
          wqs[t][e][SMALL ? cc : 0] = L.WqF[(32 * j + 16 * t + 4 * g + e) * DQ + cc];
  }
  f32x4 bqv[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const float4 q4 = *reinterpret_cast<const float4*>(L.bq1 + 32 * j + 16 * t + 4 * g);
    bqv[t] = f32x4{q4.x, q4.y, q4.z, q4.w};
  }
  // the head's 16 keys: A operand [key r][k-slots 8 g .. of the head's 32 features] and
  // V^T: A operand [feature 16 t + r][keys 4 g ..]
  const bf16x8 kpa = *reinterpret_cast<const bf16x8*>(sKp + (r * D + 32 * j + 8 * g) * 2);
  bf16x4 vta[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
    vta[t] = *reinterpret_cast<const bf16x4*>(sVt + ((32 * j + 16 * t + r) * MQ + 4 * g) * 2);

  // per-lane byte offsets inside a 16-row block of sY / sO (y_off depends on row & 15 only); each
  // set is computed where its phase starts - held from the top they cost 14 registers the kernel does
  // not have (a spilled index reloaded next to a prefetch makes the reload's vmcnt(0) wait for it)
  int oB[KS], oD[2];
#pragma unroll
  for (int s = 0; s < KS; ++s) oB[s] = y_off(r, 4 * s + g);
#pragma unroll
  for (int t = 0; t < 2; ++t) oD[t] = y_off(r, 4 * j + 2 * t + (g >> 1)) + 8 * (g & 1);

#pragma unroll
  for (int u = 0; u < UPQ; ++u) {     // (a compile-time bound: both units of a wave in ONE basic block, so
                                      //  that hipcc interleaves their dependent MFMA -> softmax -> MFMA chains)
    const int n0 = c.qn0 + 32 * u;
    f32x4 acc[2][2];
    if (SMALL) {
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const float4 x4 = *reinterpret_cast<const float4*>(sX + (n0 + 16 * nb + r) * 4);
        const float xc[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v = bqv[t][e];
#pragma unroll
            for (int cc = 0; cc < DQ; ++cc) v += wqs[t][e][SMALL ? cc : 0] * xc[cc];
            acc[t][nb][e] = v;
          }
      }
    } else {
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        acc[0][nb] = bqv[0];
        acc[1][nb] = bqv[1];
      }
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          const bf16x8 bx = *reinterpret_cast<const bf16x8*>(sY + (n0 + 16 * nb) * ROWB + oB[s]);
          acc[0][nb] = mfma32(wa[s][0], bx, acc[0][nb]);
          acc[1][nb] = mfma32(wa[s][1], bx, acc[1][nb]);
        }
      // saved for the backward: straight from the accumulators (8-byte pieces of 16 rows)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int t = 0; t < 2; ++t)
          *reinterpret_cast<bf16x4*>(L.QpS + (c.row0 + n0 + 16 * nb + r) * D + 32 * j + 16 * t + 4 * g) =
              pack4(acc[t][nb]);
    }
    // attention over the 16 inducing keys, all inside the wave
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const bf16x8 qb = pack8(acc[0][nb], acc[1][nb]);
      const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
      f32x4 s0 = mfma32(kpa, qb, z4);            // [key 4 g + e][point r]
      float mx = maxf(maxf(s0[0], s0[1]), maxf(s0[2], s0[3]));
      mx = wave16_max(mx);
      float sum = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s0[e] = __builtin_amdgcn_exp2f((s0[e] - mx) * scale_log2e);
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
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int t = 0; t < 2; ++t)
        *reinterpret_cast<bf16x4*>(sO + (n0 + 16 * nb) * ROWB + oD[t]) = pack4(acc[t][nb]);
  }
  // fc_o's weight slice replaces fc_q's (K-permuted image: the O tiles come back from sO in
  // accumulator order); requested before the barrier
#pragma unroll
  for (int s = 0; s < KS; ++s)
#pragma unroll
    for (int t = 0; t < 2; ++t)
      wa[s][t] = gload8(L.WoP + (int64_t)(32 * j + 16 * t + r) * D + 32 * s + 8 * g);
  f32x4 bov[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const float4 o4 = *reinterpret_cast<const float4*>(L.bo1 + 32 * j + 16 * t + 4 * g);
    bov[t] = f32x4{o4.x, o4.y, o4.z, o4.w};
  }
  int oP[KS][2];
#pragma unroll
  for (int s = 0; s < KS; ++s)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) oP[s][hh] = y_off(r, 4 * s + 2 * hh + (g >> 1)) + 8 * (g & 1);
  STAMP(stamp0);
  lds_barrier();                        // O rows complete; every head has read the unit's input rows
  STAMP(stamp0 + 1);
#pragma unroll
  for (int u = 0; u < UPQ; ++u) {     // (a compile-time bound: both units of a wave in ONE basic block, so
                                      //  that hipcc interleaves their dependent MFMA -> softmax -> MFMA chains)
    const int n0 = c.qn0 + 32 * u;
    f32x4 acc[2][2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      acc[0][nb] = bov[0];
      acc[1][nb] = bov[1];
    }
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const char* rowp = sO + (n0 + 16 * nb) * ROWB;
        const bf16x4 lo = *reinterpret_cast<const bf16x4*>(rowp + oP[s][0]);
        const bf16x4 hi = *reinterpret_cast<const bf16x4*>(rowp + oP[s][1]);
        bf16x8 ob;
#pragma unroll
        for (int e = 0; e < 4; ++e) { ob[e] = lo[e]; ob[4 + e] = hi[e]; }
        acc[0][nb] = mfma32(wa[s][0], ob, acc[0][nb]);
        acc[1][nb] = mfma32(wa[s][1], ob, acc[1][nb]);
      }
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      // Y = O + relu(Z) = max(O + Z, O); byte j of the lane's mask word: bit 4 t + e <-> Z > 0
      uint32_t bits = 0u;
      f32x4 y[2];
#pragma unroll
      for (int t = 1; t >= 0; --t) {
        const bf16x4 o4 = *reinterpret_cast<const bf16x4*>(sO + (n0 + 16 * nb) * ROWB + oD[t]);   // the residual
#pragma unroll
        for (int e = 3; e >= 0; --e) {
          const float zz = acc[t][nb][e], of = (float)o4[e];
          bits = push_gt0(bits, zz);
          y[t][e] = maxf(of + zz, of);
        }
      }
#pragma unroll
      for (int t = 0; t < 2; ++t)
        *reinterpret_cast<bf16x4*>(sY + (n0 + 16 * nb) * ROWB + oD[t]) = pack4(y[t]);
      const int64_t blk = (c.row0 + n0) / 16 + nb;
      reinterpret_cast<uint8_t*>(L.mask)[(blk * 64 + c.lane) * 4 + j] = (uint8_t)bits;
    }
  }
  STAMP(stamp0 + 2);
  lds_barrier();                        // Y rows complete (and every head has read the O rows)
  STAMP(stamp0 + 3);
  // saved O and the block's output: coalesced 16-byte pieces of full rows
  for (int p = c.tid; p < c.NH * 16; p += NT) {
    const int row = p >> 4, ch = p & 15;
    *reinterpret_cast<uint4*>(L.OS + (c.row0 + row) * D + ch * 8) =
        *reinterpret_cast<const uint4*>(sO + y_off(row, ch));
    *reinterpret_cast<uint4*>(L.Y + (c.row0 + row) * D + ch * 8) =
        *reinterpret_cast<const uint4*>(sY + y_off(row, ch));
  }
  STAMP(stamp0 + 4);
}

// ---------------------------------------------------------------------------------------------
// The per-set stages between the PMA's attention forward and its attention backward (k_pma_head1,
// mab0_bwd_bf16.hip, on 1024 threads): PMA epilogue O = Qp + T_h Wv_h^T + bv, P = O + relu(O Wo^T + bo)
// (modules.py:29-31), classifier + mean cross-entropy forward and backward (Code/models.py:40,
// Code/settransformer.py:104), and the adjoint of the epilogue: dZ, dO = dP + dZ Wo, dT_h = dO_h Wv_h,
// Delta = rowdot(dT, T).  Products over the INPUT index take 8 (16) adjacent lanes per output and a
// shuffle reduction; products over the OUTPUT index take thread = (column, eighth of the rows), coalesced
// weight rows, and an LDS reduction.  sPm: this workgroup's PMA partial, theirs: the partner's (sc1).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void head_stages(const PmaHeadArgs& a, const Ctx& c, const float* sPm,
                                            const float* theirs, float* sh) {
  constexpr int DH = 32, R = 4;
  float* sT = sh;                 // [4][128] merged, normalised T
  float* sOv = sT + 512;          // O, P, Z, dP, dZ, dO: [128] each
  float* sP = sOv + 128;
  float* sZ = sP + 128;
  float* sdP = sZ + 128;
  float* sdZ = sdP + 128;
  float* sdO = sdZ + 128;
  float* sL = sdO + 128;          // logits / dlogits [64]
  float* part = sL + 64;          // [8][128] partial sums of the output-index products
  float* sDl = part + 1024;       // [4][2]
  float* red = sDl + 8;           // m, sum
  int* ramax = reinterpret_cast<int*>(red + 2);
  float* sLSE = red + 4;          // [4]
  const int tid = c.tid, b = c.b, C = a.C, B = a.B;
  const int fA = tid >> 3, pA = tid & 7;            // mapping A: 8 adjacent lanes per output f
  const int fB = tid & 127, pB = tid >> 7;          // mapping B: column f, eighth pB of the rows

  // everything the first three stages read from memory, requested together
  float pth = 0.f;
  if (tid < 520)
    pth = __hip_atomic_load((__attribute__((address_space(1))) const float*)(theirs + tid), __ATOMIC_RELAXED,
                            __HIP_MEMORY_SCOPE_AGENT);
  float4 wv4[4], wo4[4], wc4[2];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    wv4[u] = *reinterpret_cast<const float4*>(a.Wv + (int64_t)fA * D + 16 * pA + 4 * u);
    wo4[u] = *reinterpret_cast<const float4*>(a.Wo + (int64_t)fA * D + 16 * pA + 4 * u);
  }
  const int c3 = tid >> 4, p3 = tid & 15;
#pragma unroll
  for (int u = 0; u < 2; ++u)
    wc4[u] = *reinterpret_cast<const float4*>(a.Wc + (int64_t)(c3 < C ? c3 : 0) * D + 8 * p3 + 4 * u);
  const float qb = a.Qp[fA] + a.bv[fA], bo_f = a.bo[fA], bc3 = a.bc[c3 < C ? c3 : 0];
  const int64_t y = a.labels[b];
  if (a.zero_ptr != nullptr)
    for (int i = b * NT + tid; i < a.zero_n; i += B * NT) a.zero_ptr[i] = 0.f;

  // ---- merge the two halves' partials (ordered by half: this is half 0) ----
  float* sEx = part;                                  // the partner's 520 values
  if (tid < 520) sEx[tid] = pth;
  lds_barrier();
  if (tid < 512) {
    const int rr = tid >> 7;
    const float m0 = sPm[512 + rr], m1 = sEx[512 + rr];
    const float M = fmaxf(m0, m1);
    const float f0 = __builtin_amdgcn_exp2f(m0 - M), f1 = __builtin_amdgcn_exp2f(m1 - M);
    const float Lt = f0 * sPm[516 + rr] + f1 * sEx[516 + rr];
    const float v = (f0 * sPm[tid] + f1 * sEx[tid]) / Lt;
    sT[tid] = v;
    a.T[(int64_t)b * R * D + tid] = v;
    if ((tid & 127) == 0) {
      const float lse = M + log2f(Lt);
      a.LSE[(int64_t)b * R + rr] = lse;
      sLSE[rr] = lse;
    }
  }
  lds_barrier();
  auto dot16 = [&](const float4 (&w)[4], const float* v) {
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float4 x = *reinterpret_cast<const float4*>(v + 4 * u);
      acc += w[u].x * x.x + w[u].y * x.y + w[u].z * x.z + w[u].w * x.w;
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64);
    return acc;
  };
  // O = Qp + T_h Wv_h^T + bv
  {
    const float o1 = qb + dot16(wv4, sT + (fA / DH) * D + 16 * pA);
    if (pA == 0) sOv[fA] = o1;
  }
  lds_barrier();
  // Z = O Wo^T + bo ; P = O + relu(Z)
  {
    const float z1 = bo_f + dot16(wo4, sOv + 16 * pA);
    if (pA == 0) {
      const float o1 = sOv[fA], hv = o1 + fmaxf(z1, 0.f);
      const int64_t o = (int64_t)b * D + fA;
      a.H[o] = hv;
      a.Osave[o] = o1;
      a.Zsave[o] = z1;
      sP[fA] = hv;
      sZ[fA] = z1;
    }
  }
  lds_barrier();
  // the weights of the adjoint stages, requested under the classifier
  float wcB[8], woB[16];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int cc = pB + 8 * u;
    wcB[u] = a.Wc[(int64_t)(cc < C ? cc : 0) * D + fB];
  }
#pragma unroll
  for (int u = 0; u < 16; ++u) woB[u] = a.Wo[(int64_t)(pB + 8 * u) * D + fB];
  // logits
  {
    float acc = 0.f;
    const float* x = sP + 8 * p3;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const float4 x4 = *reinterpret_cast<const float4*>(x + 4 * u);
      acc += wc4[u].x * x4.x + wc4[u].y * x4.y + wc4[u].z * x4.z + wc4[u].w * x4.w;
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64);
    acc += __shfl_xor(acc, 8, 64);
    if (c3 < C && p3 == 0) {
      acc += bc3;
      sL[c3] = acc;
      a.logits[(int64_t)b * C + c3] = acc;
    }
  }
  lds_barrier();
  if (tid < 64) {
    float m = -INFINITY;
    int am = 0x7fffffff;
    for (int j = tid; j < C; j += 64)
      if (sL[j] > m) { m = sL[j]; am = j; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float om = __shfl_xor(m, o, 64);
      const int oa = __shfl_xor(am, o, 64);
      if (om > m || (om == m && oa < am)) { m = om; am = oa; }
    }
    float sm = 0.f;
    for (int j = tid; j < C; j += 64) sm += expf(sL[j] - m);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 64);
    if (tid == 0) { red[0] = m; red[1] = sm; *ramax = am; }
  }
  lds_barrier();
  {
    const float m = red[0], sm = red[1];
    const float gs = a.grad_scale / (float)B;
    float g = 0.f;
    if (tid < C) g = (expf(sL[tid] - m) / sm - (tid == y ? 1.f : 0.f)) * gs;
    if (tid == 0) {
      a.lossv[b] = m + logf(sm) - sL[y];
      a.corrv[b] = *ramax == (int)y ? 1.f : 0.f;
    }
    lds_barrier();
    if (tid < C) {
      sL[tid] = g;
      a.dlogits[(int64_t)b * C + tid] = g;
    }
    lds_barrier();
  }
  // dP = dlogits Wc
  {
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int cc = pB + 8 * u;
      if (cc < C) acc = fmaf(sL[cc], wcB[u], acc);
    }
    part[pB * D + fB] = acc;
  }
  lds_barrier();
  if (tid < D) {
    float dp = 0.f;
#pragma unroll
    for (int p = 0; p < 8; ++p) dp += part[p * D + tid];
    a.dP[(int64_t)b * D + tid] = dp;
    sdP[tid] = dp;
    const float v = sZ[tid] > 0.f ? dp : 0.f;          // dZ = dP . [Z > 0]
    sdZ[tid] = v;
    a.dZ[(int64_t)b * D + tid] = v;
  }
  lds_barrier();
  // dO = dP + dZ Wo
  float wvB[16];
  {
    const int j = (tid >> 7) & 3, hlf = tid >> 9;      // stage after this one: (column, head, half of the head)
#pragma unroll
    for (int u = 0; u < 16; ++u) wvB[u] = a.Wv[(int64_t)(j * DH + 16 * hlf + u) * D + fB];
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) acc = fmaf(sdZ[pB + 8 * u], woB[u], acc);
    part[pB * D + fB] = acc;
  }
  lds_barrier();
  if (tid < D) {
    float v = sdP[tid];
#pragma unroll
    for (int p = 0; p < 8; ++p) v += part[p * D + tid];
    sdO[tid] = v;
    a.dO[(int64_t)b * D + tid] = v;
  }
  lds_barrier();
  // dT_h = dO_h Wv_h ; Delta = rowdot(dT, T) ; the images k_mab0_bwd reads
  {
    const int j = (tid >> 7) & 3, hlf = tid >> 9;
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) acc = fmaf(sdO[j * DH + 16 * hlf + u], wvB[u], acc);
    part[(hlf * 4 + j) * D + fB] = acc;
  }
  lds_barrier();
  if (tid < 512) {
    const int j = tid >> 7, f = tid & 127;
    const float acc = part[j * D + f] + part[(4 + j) * D + f];
    const float tv = sT[j * D + f];
    a.Th[((int64_t)j * B + b) * D + f] = tv;
    a.dTb[((int64_t)b * a.Rp + j) * D + f] = (__bf16)acc;
    int pos = 0;
#pragma unroll
    for (int p = 0; p < 32; ++p)
      if (perm32(p) == j) pos = p;
    a.dTt[((int64_t)b * D + f) * a.Rp + pos] = (__bf16)acc;
    float dl = acc * tv;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dl += __shfl_xor(dl, o, 64);
    if ((tid & 63) == 0) sDl[2 * j + ((tid >> 6) & 1)] = dl;
  }
  lds_barrier();
  for (int r = tid; r < a.Rp; r += NT) {
    a.Delta[(int64_t)b * a.Rp + r] = r < R ? sDl[2 * r] + sDl[2 * r + 1] : 0.f;
    a.LSEp[(int64_t)b * a.Rp + r] = r < R ? sLSE[r] : 1.0e30f;
  }
  for (int o = tid; o < (a.Rp - R) * D; o += NT) {          // padding rows / columns of the images
    const int r = R + o / D, cc = o % D;
    a.dTb[((int64_t)b * a.Rp + r) * D + cc] = (__bf16)0.f;
    const int rb32 = r & ~31, ro = r & 31;
    int pos = 0;
#pragma unroll
    for (int p = 0; p < 32; ++p)
      if (perm32(p) == ro) pos = p;
    a.dTt[((int64_t)b * D + cc) * a.Rp + rb32 + pos] = (__bf16)0.f;
  }
}

__device__ __forceinline__ void mab1_load_wq(const Set128Layer& L, const Ctx& c, bf16x8 (&wa)[4][2]) {
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int t = 0; t < 2; ++t)
      wa[s][t] = gload8(L.WqB + (int64_t)(32 * c.j + 16 * t + c.r) * D + 32 * s + 8 * c.g);
}

typedef __attribute__((address_space(4))) const Set128FwdArgs karg_t;
// The kernel's arguments are ~75 pointers: left to itself hipcc loads them all at entry and then spills
// SGPRs to VGPR lanes (930 v_readlane in the first version).  Reading them through a pointer the
// optimiser cannot see through keeps every s_load at its phase.
__device__ __forceinline__ karg_t* launder(karg_t* p) {
  asm volatile("" : "+s"(p));
  return p;
}
// The same for the thread index at a phase boundary: everything derived from it (lane, r, g, the swizzled
// LDS offsets ...) is recomputed in the next phase - a handful of VALU instructions - instead of being kept
// alive across phases by common-subexpression elimination, which cost ~20 spilled registers whose scratch
// reloads (vmcnt!) then sat next to the prefetches
__device__ __forceinline__ void rederive(Ctx& c) {
  int t = c.tid;
  asm volatile("" : "+v"(t));
  c.tid = t; c.lane = t & 63; c.r = t & 15; c.g = (t >> 4) & 3;
}

// (explicit address-space cast: the host pass of hipcc rejects the implicit one; the optimiser infers the
//  constant address space back, the loads stay scalar)
#define LAYER(i) (*(const Set128Layer*)(&ap->L[i]))

template <int DIN, int UPQ>
__global__ __launch_bounds__(NT) void k_set128_fwd(const Set128FwdArgs a_by_value) {
  (void)a_by_value;
  karg_t* ap = launder((karg_t*)__builtin_amdgcn_kernarg_segment_ptr());
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sY = smem;                               // [256][256 B]
  char* sO = smem + 256 * ROWB;                  // [256][256 B]; between blocks: merge buffers
  char* sS = smem + 512 * ROWB;                  // 32 KiB of images
  float* sX = reinterpret_cast<float*>(sS);      // [N][4] the set's points (layer 1)           0 ..  8 K
  char* sA = sS + 8192;                          // O image of the mid stage                    8 .. 12 K
  char* sH = sS + 12288;                         // H image                                    12 .. 16 K
  char* sKp = sS + 16384;                        // K image of the many-queries block          16 .. 20 K
  char* sVt = sS + 20480;                        // V^T image                                  20 .. 24 K
  float* sTf = reinterpret_cast<float*>(sS + 24576);   // layer 1: merged T fp32 [64][4]       24 .. 25 K
  float* sAl = reinterpret_cast<float*>(sS + 25600);   // 32 floats per wave                   25 .. 27 K
  float* sML = reinterpret_cast<float*>(sS + 27648);   // [16 waves][16 columns] float2        27 .. 29 K
  char* sT = sO;                                 // bf16 image of the merged T (layer 2)

  Ctx c;
  c.tid = threadIdx.x; c.lane = c.tid & 63;
  c.wave = __builtin_amdgcn_readfirstlane(c.tid >> 6);
  c.q = c.wave >> 2; c.j = c.wave & 3; c.r = c.lane & 15; c.g = c.lane >> 4;
  {
    // pair = workgroups w and w + 8: the same XCD when ids are dealt round-robin over the 8 XCDs (a
    // placement bonus - same L2 - never a correctness condition)
    const int w = blockIdx.x, xcd = w & 7, k = w >> 3;
    c.half = k & 1;
    c.b = (k >> 1) * 8 + xcd;
  }
  if (c.b >= ap->B) return;
  c.N = ap->N; c.NH = c.N >> 1;
  c.UPQ = c.N >> 8; c.qn0 = c.q * (c.NH >> 2);
  c.row0 = (int64_t)c.b * c.N + c.half * c.NH;
  constexpr int dk = DIN;
  const int b = c.b, NH = c.NH;
  uint32_t* const flags = ap->flags;
  STAMP(0);

  // ================= layer 1, few-queries block: scores G x, softmax over the points, T = A x ====
  // (k_mab0_attn_small's arithmetic: exact fp32 on the vector ALU.)  Its cost is 2 din + 6 operations
  // per (score row, point): BOTH workgroups of the pair run it over the WHOLE set instead of handing
  // partials to each other.  lane = score row (head * 16 + query), wave w walks the points
  // [w N/16, (w+1) N/16) - their coordinates are wave-uniform LDS broadcasts; wave 0 merges.
  MidPre pre;
  {
    const Set128Layer& L = LAYER(0);
    const int N = c.N, tid = c.tid, lane = c.lane;
    float xs[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {                 // N * 4 <= 2048 staged words
      const int i = tid + NT * it;
      const int pt = i >> 2, cc = i & 3;
      const int ptc = pt < N ? pt : N - 1, ccc = cc < dk ? cc : 0;      // unconditional, clamped
      const float v = ap->X[((int64_t)b * N + ptc) * dk + ccc];
      xs[it] = cc < dk ? v : 0.f;
    }
    float gk[DIN];
#pragma unroll
    for (int cc = 0; cc < DIN; ++cc) gk[cc] = L.Gf[lane * dk + cc];
    // two images of the points: [point][4] for the layer-1 projection of the many-queries block, and
    // [pair of points][component][2] - the operand pairs of the packed-fp32 loop below (v_pk_fma_f32 /
    // v_pk_mul_f32 work on two points per issue slot, as in k_mab0_attn_small)
    float* sXp = reinterpret_cast<float*>(sO) + 8192;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int i = tid + NT * it;
      if (i < N * 4) {
        const int pt = i >> 2, cc = i & 3;
        sX[i] = xs[it];
        sXp[(pt >> 1) * 8 + cc * 2 + (pt & 1)] = xs[it];
      }
    }
    lds_barrier();
    const int npw = N >> 5, q0 = c.wave * npw;       // 8 or 16 pairs of points per wave
    float m = -INFINITY;
#pragma unroll 8
    for (int i = 0; i < npw; ++i) {
      const float* px = sXp + (q0 + i) * 8;
      f32x2 sv = gk[0] * *reinterpret_cast<const f32x2*>(px);
#pragma unroll
      for (int cc = 1; cc < DIN; ++cc) sv += gk[cc] * *reinterpret_cast<const f32x2*>(px + 2 * cc);
      m = maxf(m, maxf(sv[0], sv[1]));
    }
    f32x2 l2 = {0.f, 0.f}, t2[DIN];
#pragma unroll
    for (int cc = 0; cc < DIN; ++cc) t2[cc] = f32x2{0.f, 0.f};
#pragma unroll 8
    for (int i = 0; i < npw; ++i) {
      const float* px = sXp + (q0 + i) * 8;
      f32x2 xv[DIN];
#pragma unroll
      for (int cc = 0; cc < DIN; ++cc) xv[cc] = *reinterpret_cast<const f32x2*>(px + 2 * cc);
      f32x2 sv = gk[0] * xv[0];
#pragma unroll
      for (int cc = 1; cc < DIN; ++cc) sv += gk[cc] * xv[cc];
      const f32x2 p = {__builtin_amdgcn_exp2f(sv[0] - m), __builtin_amdgcn_exp2f(sv[1] - m)};
      l2 += p;
#pragma unroll
      for (int cc = 0; cc < DIN; ++cc) t2[cc] += p * xv[cc];
    }
    const float l = l2[0] + l2[1];
    float t4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int cc = 0; cc < DIN; ++cc) t4[cc] = t2[cc][0] + t2[cc][1];
    float* sM = reinterpret_cast<float*>(sO);              // [16][64]
    float* sL = sM + 16 * 64;                              // [16][64]
    float* sTp = sL + 16 * 64;                             // [16][64][4]
    sM[c.wave * 64 + lane] = m;
    sL[c.wave * 64 + lane] = l;
    *reinterpret_cast<float4*>(sTp + (c.wave * 64 + lane) * 4) = float4{t4[0], t4[1], t4[2], t4[3]};
    // the mid stage's weight fragments, one barrier ahead.  (Requested at kernel entry they were live
    // across the whole attention loop and hipcc spilled every one of them right behind its load: seven
    // serialised round trips at the start of the kernel.)
    mid_prefetch<true>(L, c, dk, pre);
    STAMP(1);
    lds_barrier();
    if (c.wave == 0) {
      float M = -INFINITY;
#pragma unroll 4
      for (int p = 0; p < 16; ++p) M = maxf(M, sM[p * 64 + lane]);
      float Ls = 0.f, tt[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
      for (int p = 0; p < 16; ++p) {
        const float f = __builtin_amdgcn_exp2f(sM[p * 64 + lane] - M);
        const float4 tp = *reinterpret_cast<const float4*>(sTp + (p * 64 + lane) * 4);
        Ls += sL[p * 64 + lane] * f;
        tt[0] += tp.x * f; tt[1] += tp.y * f; tt[2] += tp.z * f; tt[3] += tp.w * f;
      }
      const float inv = 1.f / Ls;
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) tt[cc] *= inv;
      *reinterpret_cast<float4*>(sTf + lane * 4) = float4{tt[0], tt[1], tt[2], tt[3]};
      if (c.half == 0) {
        for (int cc = 0; cc < dk; ++cc) L.T[((int64_t)b * 64 + lane) * dk + cc] = tt[cc];
        L.LSE[(int64_t)b * 64 + lane] = M + log2f(Ls);
      }
    }
    lds_barrier();
  }
  STAMP(2);
  ap = launder(ap);
  rederive(c);
  mid_stage<true>(LAYER(0), c, pre, sT, sTf, sA, sH, sKp, sVt);
  STAMP(3);

  // ================= layer 1, many-queries block ================================================
  ap = launder(ap);
  rederive(c);
  bf16x8 wa[4][2];
  mab1_phase<true, DIN, UPQ>(LAYER(0), c, sY, sO, sX + c.half * NH * 4, sKp, sVt, ap->scale_log2e, 4, wa);

  // ================= layer 2, few-queries block over the rows in sY (k_mab0_attn_h4) ==============
  ap = launder(ap);
  rederive(c);
  {
    const Set128Layer& L = LAYER(1);
    const int tid = c.tid, lane = c.lane, r = c.r, g = c.g;
    bf16x8 gf[4];                     // this head's G rows: B operand of the score MFMAs
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) gf[ks] = gload8(L.Gb + (int64_t)(16 * c.j + r) * D + 32 * ks + 8 * g);
    float* myAl = sAl + c.wave * 32;
    float mrow = -INFINITY, lrow = 0.f;
    f32x4 T[8];
#pragma unroll
    for (int ft = 0; ft < 8; ++ft) T[ft] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < UPQ; ++u) {
      const char* img = sY + (c.qn0 + 32 * u) * ROWB;
      f32x4 s[2];
#pragma unroll
      for (int pb = 0; pb < 2; ++pb) {
        s[pb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          s[pb] = mfma32(*reinterpret_cast<const bf16x8*>(img + y_off(16 * pb + r, 4 * ks + g)), gf[ks],
                         s[pb]);
      }
      // rows of s = points 16 pb + 4 g + e ; column = query row r of this head
      float mt = -INFINITY;
#pragma unroll
      for (int pb = 0; pb < 2; ++pb)
#pragma unroll
        for (int e = 0; e < 4; ++e) mt = maxf(mt, s[pb][e]);
      mt = wave16_max(mt);
      const float mnew = maxf(mrow, mt);
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
      if (g == 0) myAl[r] = alpha;                    // T rows are query rows 4g+e
      const float4 a4 = *reinterpret_cast<const float4*>(&myAl[4 * g]);
      const bf16x8 pa = pack8(s[0], s[1]);
#pragma unroll
      for (int ft = 0; ft < 8; ++ft) {
        T[ft][0] *= a4.x; T[ft][1] *= a4.y; T[ft][2] *= a4.z; T[ft][3] *= a4.w;
        T[ft] = mfma32(pa, y_tr_frag(img, ft, lane), T[ft]);
      }
    }
    STAMP(9);
    // ---- the four quads' partials of a head -> quad 0, in two rounds through sO (8 KiB per wave) ----
    // (m, l) of query row r live in every lane of column r; the T accumulators hold query rows
    // 4 g + e: their factors come back from LDS as a float4
    auto put_partial = [&](int slot) {
      float4* dst = reinterpret_cast<float4*>(sO) + slot * 512;
#pragma unroll
      for (int ft = 0; ft < 8; ++ft) dst[ft * 64 + lane] = float4{T[ft][0], T[ft][1], T[ft][2], T[ft][3]};
      if (g == 0) *reinterpret_cast<float2*>(sML + (c.wave * 16 + r) * 2) = float2{mrow, lrow};
    };
    auto take_partial = [&](int slot, int from_wave) {
      const float2 ml = *reinterpret_cast<const float2*>(sML + (from_wave * 16 + r) * 2);
      const float mn = maxf(mrow, ml.x);
      const float f1 = __builtin_amdgcn_exp2f(mrow - mn), f2 = __builtin_amdgcn_exp2f(ml.x - mn);
      lrow = lrow * f1 + ml.y * f2;
      mrow = mn;
      if (g == 0) {
        myAl[r] = f1;
        myAl[16 + r] = f2;
      }
      const float4 a1 = *reinterpret_cast<const float4*>(&myAl[4 * g]);
      const float4 a2 = *reinterpret_cast<const float4*>(&myAl[16 + 4 * g]);
      const float4* src = reinterpret_cast<const float4*>(sO) + slot * 512;
#pragma unroll
      for (int ft = 0; ft < 8; ++ft) {
        const float4 o = src[ft * 64 + lane];
        T[ft][0] = T[ft][0] * a1.x + o.x * a2.x;
        T[ft][1] = T[ft][1] * a1.y + o.y * a2.y;
        T[ft][2] = T[ft][2] * a1.z + o.z * a2.z;
        T[ft][3] = T[ft][3] * a1.w + o.w * a2.w;
      }
    };
    lds_barrier();                       // all score reads of sY done; sO (saved O rows) is stored
    STAMP(20);
    if (c.q & 1) put_partial((c.q >> 1) * 4 + c.j);
    lds_barrier();
    if (!(c.q & 1)) take_partial((c.q >> 1) * 4 + c.j, c.wave + 4);
    lds_barrier();
    STAMP(21);
    if (c.q == 2) put_partial(c.j);
    lds_barrier();
    STAMP(22);
    float* mine = ap->ex2 + (int64_t)(b * 2 + c.half) * 9216;
    float* theirs = ap->ex2 + (int64_t)(b * 2 + (c.half ^ 1)) * 9216;
    if (c.q == 0) {
      take_partial(c.j, c.wave + 8);
      // hand-off 2: the workgroup's partial of head j (unnormalised T, m, l)
#pragma unroll
      for (int ft = 0; ft < 8; ++ft)
        st16_sc1(mine + ((ft * 4 + c.j) * 64 + lane) * 4, float4{T[ft][0], T[ft][1], T[ft][2], T[ft][3]});
      st16_sc1(mine + 8192 + (c.j * 64 + lane) * 4, float4{mrow, lrow, 0.f, 0.f});
      drain_vm();
    }
    STAMP(23);
    lds_barrier();                       // every storing wave has drained
    if (tid == 0) flag_store(flags + 4 + (b * 2 + 1) * 2 + c.half, 2u);
    if (c.wave == 0) flag_wait(flags + 4 + (b * 2 + 1) * 2 + (c.half ^ 1), 2u, flags);
    lds_barrier();
    STAMP(24);
    float* sTs = reinterpret_cast<float*>(sO) + 8192;       // staging [64][128] fp32 behind the slots
    if (c.q == 0) {
      f32x4 pt[8];
      f32x4 pml = ld16_sc1(theirs + 8192 + (c.j * 64 + lane) * 4);
#pragma unroll
      for (int ft = 0; ft < 8; ++ft) pt[ft] = ld16_sc1(theirs + ((ft * 4 + c.j) * 64 + lane) * 4);
#pragma unroll
      for (int ft = 0; ft < 8; ++ft) landed(pt[ft]);
      landed(pml);
      STAMP(26);
      // the SAME expression in both halves - f_1 x_1 + (f_0 x_0), the half-0 product rounded first, whoever
      // "own" is - so that the two workgroups of a set merge to bit-identical T (hipcc contracts
      // __fadd_rn(__fmul_rn(..), __fmul_rn(..)) into v_fmac: "own product first" differed between them)
      const bool h0 = c.half == 0;
      auto comb = [&](float fo_, float xo_, float fp_, float xp_) {
        return h0 ? __builtin_fmaf(fp_, xp_, fo_ * xo_) : __builtin_fmaf(fo_, xo_, fp_ * xp_);
      };
      const float Mx = fmaxf(mrow, pml.x);
      const float fo = __builtin_amdgcn_exp2f(mrow - Mx), fp = __builtin_amdgcn_exp2f(pml.x - Mx);
      const float Lt = comb(fo, lrow, fp, pml.y);
      const float inv = 1.f / Lt;
      if (g == 0) {
        myAl[r] = fo;
        myAl[16 + r] = fp;
      }
      if (g == 0) *reinterpret_cast<float2*>(sML + (c.wave * 16 + r) * 2) = float2{inv, Mx + log2f(Lt)};
      const float4 a1 = *reinterpret_cast<const float4*>(&myAl[4 * g]);
      const float4 a2 = *reinterpret_cast<const float4*>(&myAl[16 + 4 * g]);
      const float a1v[4] = {a1.x, a1.y, a1.z, a1.w}, a2v[4] = {a2.x, a2.y, a2.z, a2.w};
#pragma unroll
      for (int ft = 0; ft < 8; ++ft) {
        const float pv[4] = {pt[ft].x, pt[ft].y, pt[ft].z, pt[ft].w};
#pragma unroll
        for (int e = 0; e < 4; ++e)       // unnormalised; row 16 j + 4 g + e, feature 16 ft + r
          sTs[(16 * c.j + 4 * g + e) * D + 16 * ft + r] = comb(a1v[e], T[ft][e], a2v[e], pv[e]);
      }
    }
    mid_prefetch<false>(L, c, D, pre);   // one barrier ahead of the mid stage (see layer 1)
    STAMP(25);
    lds_barrier();
    {                                  // T (global fp32, saved) and its bf16 image; (inv, lse) of row: lane `row & 15` of wave j
      const int row = tid >> 4, ch = tid & 15;
      const float2 il = *reinterpret_cast<const float2*>(sML + ((row >> 4) * 16 + (row & 15)) * 2);
      const float4 lo = *reinterpret_cast<const float4*>(sTs + row * D + ch * 8);
      const float4 hi = *reinterpret_cast<const float4*>(sTs + row * D + ch * 8 + 4);
      float t[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
      bf16x8 v;
#pragma unroll
      for (int k = 0; k < 8; ++k) { t[k] *= il.x; v[k] = (__bf16)t[k]; }
      if (c.half == 0) {
        float4* tg = reinterpret_cast<float4*>(L.T + ((int64_t)b * 64 + row) * D + ch * 8);
        tg[0] = float4{t[0], t[1], t[2], t[3]};
        tg[1] = float4{t[4], t[5], t[6], t[7]};
        if (ch == 0) L.LSE[(int64_t)b * 64 + row] = il.y;
      }
      *reinterpret_cast<bf16x8*>(sT + swz(row, ch, ROWB)) = v;
    }
    lds_barrier();
    STAMP(10);
    mab1_load_wq(L, c, wa);            // fc_q's slice of the many-queries block, a stage ahead
    mid_stage<false>(L, c, pre, sT, sTf, sA, sH, sKp, sVt);
    STAMP(11);
  }

  // ================= layer 2, many-queries block ================================================
  ap = launder(ap);
  rederive(c);
  mab1_phase<false, D, UPQ>(LAYER(1), c, sY, sO, sX, sKp, sVt, ap->scale_log2e, 12, wa);

  // ================= PMA attention partials over Y2 (k_mab0_attn<1>): wave w < NH / 32 = unit w ====
  ap = launder(ap);
  rederive(c);
  {
    const int tid = c.tid, lane = c.lane, r = c.r, g = c.g;
    float mrow = -INFINITY, lrow = 0.f;
    f32x4 T[8];
#pragma unroll
    for (int ft = 0; ft < 8; ++ft) T[ft] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nun = NH >> 5;                    // 4 or 8 units
    if (c.wave < nun) {
      const char* img = sY + (32 * c.wave) * ROWB;
      f32x4 s[2];
#pragma unroll
      for (int pb = 0; pb < 2; ++pb) {
        s[pb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          s[pb] = mfma32(*reinterpret_cast<const bf16x8*>(img + y_off(16 * pb + r, 4 * ks + g)),
                         gload8(ap->Gpma + (int64_t)r * D + 32 * ks + 8 * g), s[pb]);
      }
      float mt = -INFINITY;
#pragma unroll
      for (int pb = 0; pb < 2; ++pb)
#pragma unroll
        for (int e = 0; e < 4; ++e) mt = maxf(mt, s[pb][e]);
      mrow = wave16_max(mt);
      float ls = 0.f;
#pragma unroll
      for (int pb = 0; pb < 2; ++pb)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s[pb][e] = __builtin_amdgcn_exp2f(s[pb][e] - mrow);
          ls += s[pb][e];
        }
      lrow = wave16_sum(ls);
      const bf16x8 pa = pack8(s[0], s[1]);
#pragma unroll
      for (int ft = 0; ft < 8; ++ft) T[ft] = mfma32(pa, y_tr_frag(img, ft, lane), T[ft]);
    }
    STAMP(17);
    lds_barrier();                     // sO is dead (its rows are on their way to memory): the slabs alias it
    STAMP(18);
    float* slab = reinterpret_cast<float*>(sO) + c.wave * 520;     // [4 rows][128] + m[4] + l[4]
    if (g == 0 && c.wave < nun) {
#pragma unroll
      for (int ft = 0; ft < 8; ++ft)
#pragma unroll
        for (int e = 0; e < 4; ++e) slab[e * D + 16 * ft + r] = T[ft][e];
      if (r < 4) {
        slab[512 + r] = mrow;
        slab[516 + r] = lrow;
      }
    }
    lds_barrier();
    if (!ap->fuse_head) {
      const int Spw = ap->Sp >> 1, upp = nun / Spw;         // partials of this workgroup, units per partial
      if (tid < Spw * 512) {
        const int p = tid >> 9, rr = (tid >> 7) & 3, f = tid & 127;
        const float* s0 = reinterpret_cast<const float*>(sO);
        float M = -INFINITY;
        for (int w = p * upp; w < (p + 1) * upp; ++w) M = fmaxf(M, s0[w * 520 + 512 + rr]);
        float Ls = 0.f, t = 0.f;
        for (int w = p * upp; w < (p + 1) * upp; ++w) {
          const float fs = __builtin_amdgcn_exp2f(s0[w * 520 + 512 + rr] - M);
          Ls += fs * s0[w * 520 + 516 + rr];
          t += fs * s0[w * 520 + rr * D + f];
        }
        const int64_t o = ((int64_t)b * ap->Sp + c.half * Spw + p) * 4 + rr;
        ap->TpP[o * D + f] = t;
        if (f == 0) {
          ap->MpP[o] = M;
          ap->LpP[o] = Ls;
        }
      }
      STAMP(19);
      return;
    }
    // ---- the workgroup's ONE partial over all its units: (t [4][128], m [4], l [4]) -> sPm --------
    float* sPm = reinterpret_cast<float*>(sS);              // 528 floats (the images in sS are dead)
    if (tid < 512) {
      const int rr = tid >> 7, f = tid & 127;
      const float* s0 = reinterpret_cast<const float*>(sO);
      float M = -INFINITY;
      for (int w = 0; w < nun; ++w) M = fmaxf(M, s0[w * 520 + 512 + rr]);
      float Ls = 0.f, t = 0.f;
      for (int w = 0; w < nun; ++w) {
        const float fs = __builtin_amdgcn_exp2f(s0[w * 520 + 512 + rr] - M);
        Ls += fs * s0[w * 520 + 516 + rr];
        t += fs * s0[w * 520 + rr * D + f];
      }
      sPm[tid] = t;
      if (f == 0) {
        sPm[512 + rr] = M;
        sPm[516 + rr] = Ls;
      }
    }
    lds_barrier();
    float* exMine = ap->exP + (int64_t)(b * 2 + c.half) * 528;
    float* exTheirs = ap->exP + (int64_t)(b * 2 + (c.half ^ 1)) * 528;
    if (c.half == 1) {
      // hand-off 3 (R1 again: 4-byte write-through stores, drained, barrier, one flag): half 1 is done
      if (tid < 520)
        __hip_atomic_store((__attribute__((address_space(1))) float*)(exMine + tid), sPm[tid],
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      drain_vm();
      lds_barrier();
      if (tid == 0) flag_store(flags + 4 + (b * 2 + 0) * 2 + 1, 3u);
      return;
    }
    if (c.wave == 0) flag_wait(flags + 4 + (b * 2 + 0) * 2 + 1, 3u, flags);
    lds_barrier();
    head_stages(*(const PmaHeadArgs*)(&ap->head), c, sPm, exTheirs, reinterpret_cast<float*>(sS) + 1024);
    STAMP(19);
  }
}

}  // namespace

#ifdef PCA_SET_STAMPS
extern "C" int pca_debug_set_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_set_stamps), sizeof(g_set_stamps));
}
#endif

size_t set128_flag_bytes(int B) { return align256((size_t)(4 + B * 4) * sizeof(uint32_t)); }

size_t set128_fwd_ws_bytes(int B) {
  return set128_flag_bytes(B) + align256((size_t)B * 2 * 9216 * sizeof(float)) +
         align256((size_t)B * 2 * 528 * sizeof(float));
}

bool set128_shape_ok(int B, int N, int din, int d, int h, int m, int k) {
  if (!(d == 128 && h == 4 && m == 16 && k == 1 && din >= 1 && din <= 4 && (N == 256 || N == 512)))
    return false;
  // the two workgroups of a set wait for each other: every pair must be resident at once - one
  // workgroup per CU (160 KiB of LDS), workgroup ids dealt in order
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    return n;
  }();
  return 16 * (int)cdiv(B, 8) <= cus;
}

int set128_fwd_launch(const Set128FwdArgs& a, hipStream_t st) {
  PCA_REQUIRE(set128_shape_ok(a.B, a.N, a.din, 128, 4, 16, 1), "set128_fwd: B=%d N=%d din=%d not built",
              a.B, a.N, a.din);
  PCA_REQUIRE(a.Sp == 2 || a.Sp == 4, "set128_fwd: %d PMA partials", a.Sp);
  PCA_REQUIRE(a.Sp <= a.N / 128, "set128_fwd: %d PMA partials of %d points", a.Sp, a.N);
  static std::once_flag once;
  std::call_once(once, [] {
    const void* ks[8] = {reinterpret_cast<const void*>(k_set128_fwd<1, 1>), reinterpret_cast<const void*>(k_set128_fwd<2, 1>),
                         reinterpret_cast<const void*>(k_set128_fwd<3, 1>), reinterpret_cast<const void*>(k_set128_fwd<4, 1>),
                         reinterpret_cast<const void*>(k_set128_fwd<1, 2>), reinterpret_cast<const void*>(k_set128_fwd<2, 2>),
                         reinterpret_cast<const void*>(k_set128_fwd<3, 2>), reinterpret_cast<const void*>(k_set128_fwd<4, 2>)};
    for (const void* k : ks) (void)hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  const size_t lds = (size_t)512 * ROWB + 32768;
  // reference-formulation FLOPs of the three blocks this launch covers (the PMA's
  // epilogue and the classifier run in k_pma_head1), algorithmic bytes: X in, two [N, d] bf16 tensors
  // written and read once each
  const double Nn = a.N, dd = 128, mm = 16;
  const double macs = Nn * (3.0 * a.din * dd + 7.0 * dd * dd + 8.0 * mm * dd + 2.0 * dd) + 6.0 * mm * dd * dd;
  ProfScope ps(PCA_K_SET_FWD, st, 2.0 * macs * a.B, (double)a.B * (4.0 * Nn * a.din + 8.0 * Nn * dd));
  const dim3 grid(16 * (unsigned)cdiv(a.B, 8));
  const int key = a.din * 2 + (a.N == 512 ? 1 : 0);       // units per quad of waves: N / 256
  switch (key) {
    case 2: hipLaunchKernelGGL((k_set128_fwd<1, 1>), grid, dim3(NT), lds, st, a); break;
    case 3: hipLaunchKernelGGL((k_set128_fwd<1, 2>), grid, dim3(NT), lds, st, a); break;
    case 4: hipLaunchKernelGGL((k_set128_fwd<2, 1>), grid, dim3(NT), lds, st, a); break;
    case 5: hipLaunchKernelGGL((k_set128_fwd<2, 2>), grid, dim3(NT), lds, st, a); break;
    case 6: hipLaunchKernelGGL((k_set128_fwd<3, 1>), grid, dim3(NT), lds, st, a); break;
    case 7: hipLaunchKernelGGL((k_set128_fwd<3, 2>), grid, dim3(NT), lds, st, a); break;
    case 8: hipLaunchKernelGGL((k_set128_fwd<4, 1>), grid, dim3(NT), lds, st, a); break;
    default: hipLaunchKernelGGL((k_set128_fwd<4, 2>), grid, dim3(NT), lds, st, a); break;
  }
  ps.end();
  return check_launch("k_set128_fwd");
}

}  // namespace pca
