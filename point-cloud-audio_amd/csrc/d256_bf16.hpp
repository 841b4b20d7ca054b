// Launchers of the d = 256 training kernels (d256_bf16.hip); all activations bf16 [rows][256].
#pragma once
#include "mab1_bf16.hpp"

namespace pca {

// Y = X WP^T + bias   (WP: [256][256] K-permuted image of the nn.Linear weight, prep mode 1)
int rowgemm256_proj(const __bf16* X, const __bf16* WP, const float* bias, __bf16* Y, int B, int N,
                    hipStream_t st);
// Y = O + relu(O Wo^T + bo)  (WoP: prep mode 1 of Wo); mask (nullable): ReLU bits for the backward
int rowgemm256_fwd_o(const __bf16* O, const __bf16* WoP, const float* bo, __bf16* Y, uint32_t* mask,
                     int B, int N, hipStream_t st);
// the same two with fp8 (e4m3) MFMA operands (PCA_MODE_FP8): W8 from prep_weight_f8 (mode 1)
int rowgemm256_proj_f8(const __bf16* X, const void* W8, const float* inv_scale, const float* bias,
                       __bf16* Y, int B, int N, hipStream_t st);
int rowgemm256_fwd_o_f8(const __bf16* O, const void* W8, const float* inv_scale, const float* bo,
                        __bf16* Y, uint32_t* mask, int B, int N, hipStream_t st);
// dZ = dY . [Z > 0] (mask bits of k_mab1_fwd) ; dO = dY + dZ Wo   (WoTP: prep mode 2 of Wo)
int rowgemm256_bwd_o(const __bf16* dY, const uint32_t* mask, const __bf16* WoTP, __bf16* dZ,
                     __bf16* dO, int B, int N, hipStream_t st);
// dX (+)= G W   (WTP: prep mode 2 of W)
int rowgemm256_dx(const __bf16* G, const __bf16* WTP, __bf16* dX, int B, int N, int accumulate,
                  hipStream_t st);
// attention adjoint of the many-queries block (m = 32 keys): dQp, and the set's dKp / dVp (fp32,
// [B][32][256]; the per-range partials are summed into dKp / dVp)
int attn1_bwd256_parts(int B, int N);
int attn1_bwd256(const __bf16* dO, const __bf16* QpS, const __bf16* KpP, const __bf16* VpP,
                 const __bf16* Kt, __bf16* dQp, float* dKpPart, float* dVpPart, float* dKp,
                 float* dVp, int B, int N, hipStream_t st);

// the whole many-queries block in ONE launch (d256_fused.hip): wave = head, both weight slices in
// registers.  WqB / WoB: natural bf16 images (prep mode 0); X bf16 [B*N][256] (or fp32 [B*N][dq],
// dq <= 4, with WqF); QpS / OS / mask nullable (saved for the backward)
int isab1_fwd256_fused(const void* X, int dq, const __bf16* WqB, const float* WqF, const float* bq,
                       const __bf16* KpP, const __bf16* Vt, const __bf16* WoB, const float* bo,
                       __bf16* Y, __bf16* QpS, __bf16* OS, uint32_t* mask, int B, int N,
                       hipStream_t st, const float* inv_o = nullptr);

// dW[256 x 256] += G^T A, db[256] += colsum(G) (nullable); G, A bf16 [M][256]
struct Wgrad256Job {
  const void* G;      // bf16, or fp32 with wgrad256_launch_t(..., f32_operands = true)
  const void* A;
  float* dW;
  float* db;
  int64_t M;
  // optional (bf16 LDS-DMA kernel only, M % 32 == 0): ReLU mask words of the many-queries forward
  // over the same rows (mab1_mask_index<256> with N % 128 == 0, i.e. 128 words per 16 rows); G is
  // then used as G . [mask] - the job reads dY and the mask instead of a materialised dZ
  const uint32_t* mask;
};
struct Wgrad256Jobs {
  Wgrad256Job j[8];
  int n;
};
bool wgrad256_masked_ok(int64_t rows_per_set);
// mid256.hip: the per-set stage between the two blocks of a d = 256 ISAB in one launch
int mid256_fwd(const float* O, const float* Wo, const float* bo, const float* Wk, const float* bk,
               const float* Wv, const float* bv, float* Z, float* H, __bf16* KpP, __bf16* VpP,
               __bf16* Kt, __bf16* Vt, int B, hipStream_t st);
// Armed by the engine around an ISAB's two forward calls (training): the few-queries block then
// ends in mid256_fwd, which also writes the K / V images of the many-queries block described by
// (s1, p1, saved1), and that block's forward skips its own projection of H (mid256_kv_ready()).
void mid256_arm(const pca_mab_shape* s1, const pca_mab_params* p1, void* saved1);
bool mid256_kv_ready();
int mab0_d256_prep_all(int n, const pca_mab_shape* const* shapes, const float* const* I,
                       const pca_mab_params* params, void* const* saved, hipStream_t st);
// ... or only collected into `out`, for a launch the caller makes together with other preparation jobs
struct Mab0PrepJobs;
void mab0_d256_prep_collect(int n, const pca_mab_shape* const* shapes, const float* const* I,
                            const pca_mab_params* params, void* const* saved, Mab0PrepJobs* out);
void mab0_d256_prep_done(bool on);
// image modes the d = 256 backward asks for (fc_o / fc_q of the many-queries block, fc_k / fc_v of
// the few-queries block): they follow the A/B switches of d256_host.hip
int d256_bwd_wo_mode();
int d256_bwd_wq_mode();
int d256_bwd_kv_mode();
void wgrad256_handoff_arm(bool on);
bool wgrad256_handoff_pending();
struct DxHandoff {            // mab1's dX = dQp Wq, deferred into the few-queries block's DX launch
  const __bf16 *dQp, *WqT;
  __bf16* dX;
  int B, N;
};
int wgrad256_handoff_flush(void* ws, hipStream_t st);
size_t wgrad256_ws_bytes(int njobs, int64_t maxM);
int wgrad256_launch(const Wgrad256Jobs& jobs, void* ws, hipStream_t st);
int wgrad256_launch_t(const Wgrad256Jobs& jobs, void* ws, bool f32_operands, hipStream_t st);

int cvt_f32_bf16(const float* s, __bf16* d, int64_t n, hipStream_t st);            // n % 4 == 0
int cvt_bf16_f32(const __bf16* s, float* d, int64_t n, int accumulate, hipStream_t st);
int kv_proj_small256(const float* X, int64_t M, int dk, const float* Wk, const float* bk,
                     const float* Wv, const float* bv, __bf16* Kp, __bf16* Vp, hipStream_t st);

// layer 1 (inputs of dq <= 4 columns)
// d256_stream.hip: streaming row-GEMMs with the weights in registers
int rowstream256_proj2(const __bf16* X, const __bf16* WkB, const __bf16* WvB, const float* bk,
                       const float* bv, __bf16* Kp, __bf16* Vp, int B, int N, hipStream_t st);
int rowstream256_proj2_f8(const __bf16* X, const void* Wk8, const void* Wv8, const float* inv_scale,
                          const float* bk, const float* bv, __bf16* Kp, __bf16* Vp, int B, int N,
                          hipStream_t st);
int rowstream256_dx2(const __bf16* dKp, const __bf16* dVp, const __bf16* WkT, const __bf16* WvT,
                     __bf16* dX, int B, int N, int accumulate, hipStream_t st);
int fq_proj_attn_fwd256(const __bf16* X, const __bf16* WkB, const __bf16* WvB, const float* bk,
                        const float* bv, const float* Qp, int B, int N, int m,
                        const int32_t* lengths, __bf16* Kp, __bf16* Vp, float* Op, float* Mp,
                        float* Lp, float* O, float* LSE, hipStream_t st,
                        const float* inv_scale = nullptr);   // non-null: WkB / WvB are fp8 images
int rowstream256_dx3(const __bf16* dQp, const __bf16* dKp, const __bf16* dVp, const __bf16* WqT,
                     const __bf16* WkT, const __bf16* WvT, __bf16* dX, int B, int N, hipStream_t st);
int rowstream256_dx1(const __bf16* dQp, const __bf16* WqT, __bf16* dX, int B, int N,
                     hipStream_t st);
int attn1_bwd256_fused(const __bf16* dY, const uint32_t* mask, const __bf16* WoT, const __bf16* QpS,
                       const __bf16* KpP, const __bf16* VpP, const __bf16* Kt, __bf16* dZ,
                       __bf16* dQp, float* dKpPart, float* dVpPart, float* dKp, float* dVp, int B,
                       int N, hipStream_t st, const float* Xs = nullptr,
                       const float* WqF = nullptr, const float* bq = nullptr, int dq = 0);
// d128_fused.hip: the same single-launch forward at d = 128 / 4 heads / m = 16
int isab1_fwd128_fused(const void* X, int dq, const __bf16* WqB, const float* WqF, const float* bq,
                       const __bf16* KpP, const __bf16* Vt, const __bf16* WoP, const float* bo,
                       __bf16* Y, __bf16* QpS, __bf16* OS, uint32_t* mask, int B, int N,
                       hipStream_t st);
size_t wgrad_small256_ws_bytes(int64_t M);
int wgrad_small256(const __bf16* G, const float* X, int64_t M, int dq, float* dW, float* db,
                   void* ws, hipStream_t st);
int epi_small_fwd256(const float* T, const float* Qp, const float* Wv, const float* bv, int B, int m,
                     int dk, float* O, hipStream_t st);
size_t epi_small_bwd256_ws_bytes(int B, int m);
int epi_small_bwd256(const float* dO, const float* T, const float* Wv, int B, int m, int dk,
                     float* dT, float* Delta, float* dWv, float* dbv, void* ws, hipStream_t st);

// PMA (R = h*m <= 16 score rows) at dk = 256, reassociated: X read once, keys never projected
int pma_splits256(int B, int N);
// Gb [>=16][256] bf16 (sl2e folded, rows >= R zero); Tp [B][S][16][256], Mp / Lp [B][S][16] scratch;
// T [B][R][256] = A X, LSE [B][R] (log2 domain)
int pma_attn_fwd256(const __bf16* X, const __bf16* Gb, int B, int N, int R, const int32_t* lengths,
                    float* Tp, float* Mp, float* Lp, float* T, float* LSE, hipStream_t st);
// per set: dT = dO_h Wv_h and its images (dTb [B][16][256], TG [B][256][32]), Delta, LSEp [B][16];
// dWv += sum over sets of dO^T T
int pma_epi_bwd256(const float* dO, const float* T, const float* LSE, const float* Wv,
                   const float* Gf, int B, int m, int R, __bf16* dTb, __bf16* TG, float* Delta,
                   float* LSEp, float* dWv, hipStream_t st);
// dX (+)= P^T dT + dS^T G' ; DG [16][256] += dS X (ln2 units, as k_mab0_bwd; caller zeroes it)
int pma_attn_bwd256(const __bf16* X, const __bf16* Gb, const __bf16* dTb, const __bf16* TG,
                    const float* LSEp, const float* Delta, int B, int N, int R,
                    const int32_t* lengths, __bf16* dX, int accumulate_dx, float* DG, float* DGslabs,
                    hipStream_t st);
size_t pma_bwd256_slab_bytes(int B);
int slab_sum(const float* slabs, int S, int n, float* out, int accumulate, hipStream_t st);

// few shared queries (m <= 32) over projected keys, head dim 32
int fq_splits256(int B, int N);
// Op [B][S][m][256], Mp / Lp [B][S][8][MQ] scratch; O = Qp + A Vp [B][m][256]; LSE [B][8][MQ]
int fq_attn_fwd256(const __bf16* Kp, const __bf16* Vp, const float* Qp, int B, int N, int m,
                   const int32_t* lengths, float* Op, float* Mp, float* Lp, float* O, float* LSE,
                   hipStream_t st);
// dO [B][m][256] = gradient w.r.t. O; writes dKp, dVp (bf16 [B*N][256]) and
// dOt = dO + (attention gradient w.r.t. Qp) per set [B][m][256]
int fq_attn_bwd256(const __bf16* Kp, const __bf16* Vp, const float* Qp, const float* dO,
                   const float* O, const float* LSE, float* Delta, int B, int N, int m,
                   const int32_t* lengths, __bf16* dKp, __bf16* dVp, float* dQpPart, float* dOt,
                   hipStream_t st);

}  // namespace pca
