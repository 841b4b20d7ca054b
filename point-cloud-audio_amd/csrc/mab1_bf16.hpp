// Declarations shared by the fused bf16 mab1 forward / backward translation units.
#pragma once
#include "pca_common.h"
#include "mfma_common.hpp"

namespace pca {

constexpr int M1_TP = 128;     // points per workgroup tile (4 waves x 32)
constexpr int M1_NB = 2;       // 16-point blocks per wave

struct Mab1Saved {
  __bf16 *KpP, *VpP, *Kt, *Vt, *QpS, *OS;
  uint32_t* mask;
};
size_t mab1_carve_saved(const pca_mab_shape& s, Mab1Saved* out, void* base);

// one 32-bit word per lane covers 8 feature tiles (4 bits each)
template <int D>
__host__ __device__ __forceinline__ int64_t mab1_mask_index(int b, int tiles_per_set, int tile,
                                                            int wave, int nb, int w, int lane) {
  const int64_t pb = ((int64_t)b * tiles_per_set + tile) * (M1_TP / 16) + wave * M1_NB + nb;
  return (pb * (D / 128) + w) * 64 + lane;
}

// saved-for-backward block of the fused mab0 (few queries, many keys)
struct Mab0Saved {
  float* Qp;      // [m][d]     fc_q(I)
  float* Gf;      // [Rpad][dk] scale*log2e * Qp_h Wk_h   (fp32)
  __bf16* Gb;     // same, bf16 (MFMA operand)
  __bf16* GtP;    // [dk][Rp] K-permuted transpose of G (backward)
  float* T;       // [B][R][dk] A X
  float* LSE;     // [B][R]     log2-domain
  float *O, *Z;   // [B][m][d]
  float *WvT, *WoT;   // transposed fp32 weights ([in][out]) for the per-set epilogue
  float *Tp, *Mp, *Lp;   // per point-range partials of the attention (merged by the epilogue)
};
int mab0_splits(const pca_mab_shape& s);
struct Mab0PrepJob {
  const float *I, *Wq, *bq, *Wk;
  int m, d, dq, dk, h, Rp;
  float sl2e;
  float *Qp, *Gf;
  __bf16 *Gb, *GtP;
  // epilogue weights transposed to [in][out] fp32 by spare workgroups of the same launch
  // (null when the epilogue runs elsewhere, e.g. inside k_mid_fwd)
  const float *Wv, *Wo;
  float *WvT, *WoT;
};
struct Mab0PrepJobs {
  Mab0PrepJob j[3];
  int n;
};
void mab0_collect_prep(const pca_mab_shape& s, const float* I, const pca_mab_params& p,
                       const Mab0Saved& v, bool training, bool epilogue_images,
                       Mab0PrepJobs* J);
int mab0_prep_launch(const Mab0PrepJobs& J, hipStream_t st);
size_t mab0_carve_saved(const pca_mab_shape& s, Mab0Saved* out, void* base);

// dst[c][r] = src[r][c]  (fp32): gives the per-set row-GEMM kernels coalesced weight reads
int transpose_f32(const float* src, float* dst, int rows, int cols, hipStream_t st);

// acc[q] += sum_c sX[q*ldx + c] * WT[c*ldw + f]   for q < MQ : thread-owned output column f,
// activations broadcast from LDS, weights read coalesced (consecutive threads = consecutive f)
template <int MQ>
__device__ __forceinline__ void col_gemm(const float* sX, int ldx, const float* __restrict__ WT,
                                         int ldw, int K, int f, float (&acc)[MQ]) {
  // 16 independent weight loads in flight per thread: these per-set kernels run at one
  // workgroup per set and are otherwise bound by L2 latency, not bandwidth
  int c = 0;
  for (; c + 16 <= K; c += 16) {
    float w[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) w[u] = WT[(int64_t)(c + u) * ldw + f];
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int q = 0; q < MQ; ++q) acc[q] = fmaf(sX[q * ldx + c + u], w[u], acc[q]);
  }
  for (; c < K; ++c) {
    const float w = WT[(int64_t)c * ldw + f];
#pragma unroll
    for (int q = 0; q < MQ; ++q) acc[q] = fmaf(sX[q * ldx + c], w, acc[q]);
  }
}

// ---- batched weight-gradient reduction on the MFMA (mab1_bwd_bf16.hip) ------------------
// job: dW[128 x 128] += G[M x 128]^T . A[M x 128]  (only output rows [g_lo, g_hi) are written:
// block-diagonal per-head products), db[128] += column sums of G (nullable).
struct WgradJob {
  const void* G;
  const void* A;
  float* dW;
  float* db;
  int64_t M;
  int g_lo, g_hi;
  float* slab;            // set by the launcher: [workgroups][(g_hi - g_lo) * 128 (+ 128 with db)]
                          // partials, summed in a fixed order afterwards (null: fp32 atomics)
  // optional (bf16 G): ReLU mask words of the many-queries forward over the same rows
  // (mab1_mask_index<128> with N % 128 == 0: 64 words per 16 rows); G is then used as G . [mask] -
  // the fc_o job reads dY and the mask instead of a materialised dZ
  const uint32_t* mask;
};
struct WgradJobs {
  WgradJob j[16];
  int n;
};
// g_bf16 / a_bf16: element type of every job's G / A (bf16 or fp32)
struct SlabSumJobs;
struct WgradSlabs {          // optional slab mode of wgrad128_launch
  float* ws;                 // partials go here (cap bytes) ...
  size_t cap;
  SlabSumJobs* sums_out;     // ... and their sum jobs are appended here (run them afterwards)
  const SlabSumJobs* riders; // sums that are due now: extra workgroup rows of this launch
  size_t used;               // out: bytes of ws taken
};
int wgrad128_launch(const WgradJobs& jobs, bool g_bf16, bool a_bf16, int rows_per_wg,
                    hipStream_t st, WgradSlabs* slabs = nullptr);
// dW[128 x dq] += G[M x 128]^T . X_h[M x dq] (dq <= 4; X_h = X + head(f)*x_head_stride), db += colsum
struct BwdDefer;
int wgrad_small_f32_launch(const float* G, const float* X, int64_t M, int dq,
                           int64_t x_head_stride, float* dW, float* db, hipStream_t st,
                           BwdDefer* defer = nullptr);
// dH[q][c] (+)= dKp[q][:] . Wk[:][c] + dVp[q][:] . Wv[:][c]   per set (m = 16 rows, d = 128)
int kv_dh_launch(const float* dKp, const float* dVp, const float* Wk, const float* Wv, float* dH,
                 int B, int m, int d, int accumulate, hipStream_t st);

// ---- batched weight-image preparation (one launch per step) -----------------------------
// bf16 images of one ISAB's weights, owned by the caller (the ST engine) for a whole step
struct IsabImg {
  __bf16 *Wv0, *Wo0, *Wk1, *Wv1;                     // natural: k_mid_fwd
  __bf16 *WqB, *WoP;                                 // k_mab1_fwd (natural Wq, K-permuted Wo)
  __bf16 *Wk1T, *Wv1T, *Wo0TP, *Wv0TP, *Wv0T;        // k_mid_bwd
  __bf16 *WoTP, *WqTP;                               // k_mab1_bwd
};
struct PrepJob {
  const float* src;
  __bf16* dst;
  int rows, cols, mode;      // modes of prep_weight; 4: dst[0 .. rows * cols) = 0 (src unused)
};
struct PrepJobs {
  PrepJob j[32];
  int n;
};
int prep_jobs_launch(const PrepJobs& jobs, hipStream_t st);
// Weight images prepared ahead by the caller (the ST engine: every bf16 image the d = 256 blocks of a
// training step will ask for, in ONE launch at the start of the step instead of one ~5 us launch per
// block and direction).  While a table is in use on the calling thread, weight_image1 / 2 redirect
// *dst to a registered image of (src, mode, rows, cols) instead of converting into *dst; anything
// not registered is converted as before, so a table can only save launches, never change results.
struct WeightImages {
  struct E { const float* src; int mode, rows, cols; __bf16* img; } e[24];
  int n;
  // fp8 mode: e4m3 images of s * W (prep_weight_f8) with their inverse scales, one launch for all
  struct F8 { const float* src; int mode, rows, cols; uint8_t* img; float* inv; } f8[8];
  int nf8;
};
struct PrepF8Jobs {
  WeightImages::F8 j[8];
  int n;
};
int prep_f8_jobs_launch(const PrepF8Jobs& J, hipStream_t st);
// the registered fp8 image of (src, mode) - *dst and *inv are redirected to it - or a conversion into
// *dst / *inv on the spot
int weight_image_f8(const float* src, void** dst, int rows, int cols, int mode, float** inv,
                    hipStream_t st);
void weight_images_use(const WeightImages* t);       // thread-local; nullptr ends the scope
int weight_image1(const float* src, __bf16** dst, int rows, int cols, int mode, hipStream_t st);
int weight_image2(const float* src0, __bf16** dst0, int mode0, const float* src1, __bf16** dst1,
                  int mode1, int rows, int cols, hipStream_t st);
int mab1_fwd_wo_mode(const pca_mab_shape& s);        // image mode of fc_o the mab1 forward asks for
// the weight images AND the query-side tensors of a step in ONE launch (both depend on the
// parameters only; blockIdx.y selects the job, the two kinds share the grid)
int prep_all_launch(const PrepJobs& W, const Mab0PrepJobs& Q, hipStream_t st);

// scratch layouts of the two backward passes (shared with the ISAB-level orchestration)
struct Mab1BwdWs {
  __bf16 *WoTP, *WqTP, *dZ, *dQp, *dOs, *dS, *P;
  float *dKp, *dVp;            // [B][MI][D]
  float *dKpPart, *dVpPart;    // [B][nparts][MI][D] per-workgroup partials (fused mode)
};
size_t mab1_carve_bwd_ws(const pca_mab_shape& s, Mab1BwdWs* out, void* base);
struct Mab0BwdWs {
  float *dZ, *dO, *Th, *dTf, *Delta, *LSEp, *DG, *dQs, *dQp;
  __bf16 *dTb, *dTt, *GtP;
  float* slabs;           // [workgroups][R][dk] partial dG of k_mab0_bwd
};
int mab0_bwd_splits(const pca_mab_shape& s);
size_t mab0_carve_bwd_ws(const pca_mab_shape& s, Mab0BwdWs* out, void* base);

// flags of the *_ex host entry points used by the fused ISAB path
enum {
  PCA_F_SKIP_EPILOGUE = 1,   // mab0 fwd: stop after the attention partials (mid_fwd follows)
  PCA_F_KV_READY = 2,        // mab1 fwd: Kp/Vp images already written (by mid_fwd)
  PCA_F_SKIP_KV_TAIL = 4,    // mab1 bwd: stop after dKp/dVp (mid_bwd + batched wgrad follow)
  PCA_F_SKIP_HEAD = 8,       // mab0 bwd: dT/Delta images, dZ, dO, dQs already produced
  PCA_F_IMAGES_READY = 16,   // weight images were prepared by the caller (IsabImg)
  PCA_F_PREP_DONE = 32,      // mab0 fwd: Qp / G images were prepared by the caller
  PCA_F_SKIP_WGRAD = 64      // mab0 bwd: dWo / dWv reductions are done by the caller; DG is clear
};

// ---- per-set mid kernels of a fused ISAB (mid_bf16.hip); m = 16, d = 128, h = 4 ----------
struct MidFwdLaunch {
  int B, dk, S;
  const float *Tp, *Mp, *Lp;
  float *T, *LSE;
  const float* Qp;
  const __bf16* Wv0;
  const float* Wv0f;
  const float *bv0, *bo0;
  const __bf16 *Wo0, *Wk1, *Wv1;
  const float *bk1, *bv1;
  float *O, *Z, *H;
  __bf16 *KpP, *VpP, *Kt, *Vt;
};
int mid_fwd_launch(const MidFwdLaunch& L, hipStream_t st);
struct MidBwdLaunch {
  int B, dk;
  const float *dKpPart, *dVpPart;   // [B][nparts][16][128] partials from k_mab1_bwd
  int nparts;
  float *dKp, *dVp;                 // [B][16][128] sums (written; read by the wgrad jobs)
  float* zero_ptr;                  // optional accumulator cleared by this launch (DG)
  int zero_n;
  const float *Z, *T, *LSE;
  const __bf16 *Wk1T, *Wv1T, *Wo0TP, *Wv0TP, *Wv0T;
  const float* Wv0f;
  float *dZ, *dO, *Th, *dQs, *dTf;
  __bf16 *dTb, *dTt;
  float *Delta, *LSEp;
};
int mid_bwd_launch(const MidBwdLaunch& L, hipStream_t st);

// fp32 weight [rows][cols] -> bf16 image; mode 0 natural, 1 K-permuted, 2 transposed +
// K-permuted ([cols][rows]), 3 transposed natural
bool mab1_saves_qp(const pca_mab_shape& s);
int prep_weight2(const float* src0, __bf16* dst0, int mode0, const float* src1, __bf16* dst1,
                 int mode1, int rows, int cols, hipStream_t st);
int prep_weight(const float* src, __bf16* dst, int rows, int cols, int mode, hipStream_t st);
// fp8 e4m3 image of s * W (mode 0 natural / 1 K-permuted), s a power of two; inv_scale[0] = 1 / s
int prep_weight_f8(const float* src, void* dst, int rows, int cols, int mode, float* inv_scale,
                   hipStream_t st);

// ---- host entry points of the fused blocks (single source of truth for every TU) --------
// fused attention core of the bf16-operand GEMM chain for head dims <= 16 (attn_core.hip): O = Q_ + A V_
// and its adjoint without the [B h, nq, nk] matrix A (LSE [B][h][nq] is what is saved instead)
bool attn_core_ok(const pca_mab_shape& s);
// floats of the forward's saved statistics area / of the backward's Delta scratch
size_t attn_core_fwd_elems(const pca_mab_shape& s);
size_t attn_core_bwd_elems(const pca_mab_shape& s);
int attn_core_fwd(const pca_mab_shape& s, const float* Qp, const float* Kp, const float* Vp, float* O,
                  float* LSE, hipStream_t st);
int attn_core_bwd(const pca_mab_shape& s, const float* Qp, const float* Kp, const float* Vp,
                  const float* O, const float* LSE, const float* dO, float* dQp, float* dKp, float* dVp,
                  float* Delta, hipStream_t st);
// weight + bias gradient of a 64 -> 64 (or <= 4 -> 64) Linear over a tall activation in one launch
// (wgrad64.hip)
bool wgrad64_ok(const float* dY, const float* X, int64_t M, int din, int dout);
int wgrad64(const float* dY, const float* X, float* dW, float* db, int64_t M, int din,
            hipStream_t st);
// the 64 -> 64 / <= 4 -> 64 Linear layers over a tall activation with the weights in registers
// (linear64.hip): forward, input gradient, fc_o with the block's epilogue Y = O + relu(Z)
bool lin64_ok(const float* X, const float* Y, int64_t M, int din, int dout);
int lin64_fwd(const float* X, const float* W, const float* b, float* Y, int64_t M, int din,
              hipStream_t st);
int lin64_dx(const float* dY, const float* W, float* dX, int64_t M, int accumulate, hipStream_t st);
int lin64_fc_o(const float* O, const float* W, const float* b, float* Z, float* Y, int64_t M,
               hipStream_t st);
int lin64_fc_o_bwd(const float* dY, const float* Z, const float* W, float* dZ, float* dO, int64_t M,
                   hipStream_t st);
// exact fp32 path (mab_f32.hip)
int validate_shape(const pca_mab_shape* s);
size_t mab_f32_saved_bytes(const pca_mab_shape& s);
size_t mab_f32_bwd_ws_bytes(const pca_mab_shape& s);
int mab_f32_fwd(const pca_mab_shape& s, const float* Q, const float* K,
                const pca_mab_params& p, float* Y, void* saved, hipStream_t st);
int mab_f32_bwd(const pca_mab_shape& s, const float* Q, const float* K,
                const pca_mab_params& p, const void* saved, const float* dY, float* dQ,
                float* dK, int dk_accumulate, const pca_mab_grads& g, void* ws,
                hipStream_t st);
// while alive: linear_fwd_f32 / linear_bwd_f32 / linear_dx_acc_f32 use k_gemm_bf16 instead of the
// exact fp32 GEMM: mode 1 = bf16 MFMA operands, mode 2 = hi + lo bf16 pairs (fp32-level results
// at the same launch cost); fp32 accumulation and I/O
struct Bf16OperandScope {
  int prev;
  explicit Bf16OperandScope(int mode);
  ~Bf16OperandScope();
};
int linear_fwd_f32(const float* X, const float* W, const float* b, float* Y, int64_t M,
                   int din, int dout, hipStream_t st);
int linear_bwd_f32(const float* X, const float* W, const float* dY, float* dX, float* dW,
                   float* db, int64_t M, int din, int dout, hipStream_t st);
int linear_dx_acc_f32(const float* dY, const float* W, float* dX, int64_t M, int din, int dout,
                      int accumulate, hipStream_t st);
// fused mab1 (many queries X, few keys H).  X / Y / dY / dX are fp32 or bf16 per the shape's
// q_dtype / y_dtype; H and dH are fp32
// the shipped d = 64 / 8 heads / <= 64 inducing points shape: fused fp32 forward (sd64_fwd.hip);
// kind 1 = many queries, 2 = few shared queries, 0 = another shape
int sd64_kind(const pca_mab_shape& s);
size_t sd64_fwd_ws_bytes(const pca_mab_shape& s);
int sd64_fwd(const pca_mab_shape& s, const float* Q, const float* K, const pca_mab_params& p,
             float* Y, void* ws, hipStream_t st);
bool mab1_bf16_supported(const pca_mab_shape& s, bool inference = false);
size_t mab1_bf16_saved_bytes(const pca_mab_shape& s);
size_t mab1_bf16_fwd_ws_bytes(const pca_mab_shape& s);
size_t mab1_bf16_bwd_ws_bytes(const pca_mab_shape& s);
int mab1_bf16_fwd(const pca_mab_shape& s, const void* X, const float* H,
                  const pca_mab_params& p, void* Y, void* saved, void* ws, hipStream_t st);
int mab1_bf16_fwd_ex(const pca_mab_shape& s, const void* X, const float* H,
                     const pca_mab_params& p, void* Y, void* saved, void* ws, int flags,
                     hipStream_t st, const IsabImg* img = nullptr);
int mab1_bf16_bwd(const pca_mab_shape& s, const void* X, const float* H,
                  const pca_mab_params& p, const void* saved, const void* dY, void* dX,
                  float* dH, int dk_accumulate, const pca_mab_grads& gr, void* ws,
                  hipStream_t st);
struct BwdDefer;
int mab1_bf16_bwd_ex(const pca_mab_shape& s, const void* X, const float* H,
                     const pca_mab_params& p, const void* saved, const void* dY, void* dX,
                     float* dH, int dk_accumulate, const pca_mab_grads& gr, void* ws, int flags,
                     hipStream_t st, const IsabImg* img = nullptr, float* zero_ptr = nullptr,
                     int zero_n = 0, int* nparts_out = nullptr, BwdDefer* defer = nullptr);
// fused mab0 / PMA (few shared queries I, many keys X).  X / dX fp32 or bf16 per k_dtype
bool mab0_bf16_supported(const pca_mab_shape& s);
size_t mab0_bf16_saved_bytes(const pca_mab_shape& s);
size_t mab0_bf16_fwd_ws_bytes(const pca_mab_shape& s);
size_t mab0_bf16_bwd_ws_bytes(const pca_mab_shape& s);
int mab0_bf16_fwd(const pca_mab_shape& s, const float* I, const void* X,
                  const pca_mab_params& p, float* H, void* saved, void* ws, hipStream_t st);
int mab0_bf16_fwd_ex(const pca_mab_shape& s, const float* I, const void* X,
                     const pca_mab_params& p, float* H, void* saved, void* ws, int flags,
                     hipStream_t st);
int mab0_bf16_bwd(const pca_mab_shape& s, const float* I, const void* X,
                  const pca_mab_params& p, const void* saved, const float* dH, float* dI,
                  void* dX, int dk_accumulate, const pca_mab_grads& gr, void* ws,
                  hipStream_t st);
struct ClsWgradArgs {
  const float *dlogits, *P, *lossv, *corrv;
  int B, d, C;
  float *dWc, *dbc, *loss_out, *stats;
};
struct SmallWgradArgs {
  const float *G, *X;
  int64_t M;
  int dq, rows_per_wg;
  int64_t x_head_stride;
  float *dW, *db;
  float* slab;            // per-workgroup partials [wg][128*dq + 128] instead of atomics (null: atomics)
};
// shared-query parameter gradients (dWk, dWq, dbq, dI) of up to 3 MABs: tiny, latency-bound
// kernels, so callers may collect them and run ONE pair of launches at the end of a phase
struct Mab0PostJob {
  const float *dQs, *DG, *Qp, *Wk, *I, *Wq;
  float *dWk, *dQp, *dWq, *dbq, *dI;
  int m, d, dk, dq, h;
  float sl2e;
  // dQs == null: the sum over sets of dO [B][m][d] is taken inside the post kernel (cheaper
  // than B workgroups adding atomically into the same m*d addresses)
  const float* dO;
  int B;
};
struct Mab0PostJobs {
  Mab0PostJob j[3];
  int n;
};
int mab0_post_launch(const Mab0PostJobs& J, hipStream_t st);
// Terminal reductions of a backward pass (only the optimiser / the all-reduce reads their
// results): a caller that runs several blocks collects them and flushes ONCE at the end of the
// phase - one bf16 and one fp32 weight-gradient launch (job tables) and one pair of post
// launches instead of one set per block.  Their operands live in per-block workspaces that
// stay untouched until then.
// fixed-order sums of per-workgroup partial tensors ([S][n] fp32 -> out[n]), several per launch
struct SlabSumJob {
  const float* slabs;
  float* out;
  int S, n, accumulate;
  int stride;             // floats between consecutive slabs (0: n)
};
struct SlabSumJobs {
  SlabSumJob j[40];
  int n;
};
inline bool slab_sum_job_ok(const SlabSumJob& j) {      // 16-byte accesses throughout
  return j.n % 4 == 0 && j.stride % 4 == 0 && ((uintptr_t)j.out & 15) == 0 &&
         ((uintptr_t)j.slabs & 15) == 0;
}
int slab_sum_jobs(const SlabSumJobs& J, hipStream_t st);
struct BwdDefer {
  SlabSumJobs sums;       // partial sums the post stages read: run before them
  SlabSumJobs late;       // partial sums only the optimizer reads (ride in the last launches)
  // d = 256: the [B*m]-row weight-gradient jobs of all blocks (fp32 operands, k_wgrad256<float>):
  // one launch + one sum at the end instead of one pair per block.  Collected only when wg256_ws is
  // set (room for wgrad256_ws_bytes(8, rows)); the operands stay in the blocks' workspaces
  struct Wg256 { const void *G, *A; float *dW, *db; int64_t M; } wg256[8];
  int wg256_n;
  void* wg256_ws;
  float* slab_ws;         // room for the weight-gradient partials of the two deferred lists
  size_t slab_cap;        // (bytes; null / 0: those reductions use fp32 atomics)
  Mab0PostJobs posts;
  WgradJobs wg_bf16;      // G, A bf16, M = B*N rows   (512 rows per workgroup)
  WgradJobs wg_f32;       // G, A fp32, M = B*m rows   (64 rows per workgroup)
  // classifier weight gradient + loss counters, layer-1 fc_v gradient: they ride in the first
  // post launch (k_terminal1) as extra job rows
  ClsWgradArgs cls;
  SmallWgradArgs sw;
  int has_cls, has_sw;
};
int bwd_defer_flush(BwdDefer& D, hipStream_t st);
int wgrad256_flush_deferred(BwdDefer& D, hipStream_t st);       // d256_host.hip
bool wgrad_slabs_on();       // reductions of the fused d = 128 path as slabs + fixed-order sums
                             // (PCA_WGRAD_SLABS=0: fp32 atomics)
// PMA epilogue + classifier + cross-entropy (forward and backward) + PMA backward epilogue of the
// train step in ONE launch per set (after mab0_bf16_fwd_ex(..., PCA_F_SKIP_EPILOGUE); followed
// by mab0_bf16_bwd_ex(..., PCA_F_SKIP_HEAD)).  P [B, d] receives the pooled features; the
// classifier's weight gradient is queued in `defer`.  ws_bwd is the PMA's backward workspace.
int pma_head_launch(const pca_mab_shape& s, const pca_mab_params& p, void* saved, void* ws_bwd,
                    float* P, const float* Wc, const float* bc, const int64_t* labels, int C,
                    float grad_scale, float* logits, float* dlogits, float* dP, float* dWc,
                    float* dbc, float* loss_out, float* stats, float* cls_ws, BwdDefer* defer,
                    hipStream_t st);
// arguments of that launch (k_pma_head / k_pma_head1; the set-resident forward runs the same stages in its
// own tail: set128_fwd.hip)
struct PmaHeadArgs {
  // forward epilogue
  const float *Tp, *Mp, *Lp;
  int S;
  float *T, *LSE;
  const float *Qp, *WvT, *bv, *WoT, *bo;
  int m, d, dk, h;
  float *H, *Osave, *Zsave;
  // classifier + loss
  const float *Wc, *bc;
  const int64_t* labels;
  int B, C;
  float grad_scale;
  float *logits, *dlogits, *dP, *lossv, *corrv;
  // backward epilogue
  const float *Wo, *Wv;
  int Rp;
  float *dZ, *dO, *Th, *dTf;
  __bf16 *dTb, *dTt;
  float *Delta, *LSEp, *zero_ptr;
  int zero_n;
};
int pma_head_args(const pca_mab_shape& s, const pca_mab_params& p, void* saved, void* ws_bwd,
                  float* P, const float* Wc, const float* bc, const int64_t* labels, int C,
                  float grad_scale, float* logits, float* dlogits, float* dP, float* dWc,
                  float* dbc, float* loss_out, float* stats, float* cls_ws, BwdDefer* defer,
                  PmaHeadArgs* out);
// post stages + riders (`late`: sums nobody reads before the optimizer, e.g. weight gradients)
int terminal_launch(const BwdDefer& D, hipStream_t st, const SlabSumJobs* late = nullptr);
// launch `jobs` now, or append them to the matching list of `defer`
int wgrad128_defer(BwdDefer* defer, const WgradJobs& jobs, bool bf16, int rows_per_wg,
                   hipStream_t st);
// `defer` non-null: the post job is appended there instead of being launched
int mab0_bf16_bwd_ex(const pca_mab_shape& s, const float* I, const void* X,
                     const pca_mab_params& p, const void* saved, const float* dH, float* dI,
                     void* dX, int dk_accumulate, const pca_mab_grads& gr, void* ws, int flags,
                     hipStream_t st, BwdDefer* defer = nullptr);
// d = 256 / 8 heads (d256_host.hip): the many-queries backward and the few-queries block
size_t mab1_d256_bwd_ws_bytes(const pca_mab_shape& s);
int mab1_d256_bwd(const pca_mab_shape& s, const void* X, const float* H, const pca_mab_params& p,
                  const void* saved, const void* dY, void* dX, float* dH, int dk_accumulate,
                  const pca_mab_grads& gr, void* ws, hipStream_t st, BwdDefer* defer = nullptr);
bool mab0_d256_supported(const pca_mab_shape& s);
size_t mab0_d256_saved_bytes(const pca_mab_shape& s);
size_t mab0_d256_fwd_ws_bytes(const pca_mab_shape& s);
size_t mab0_d256_bwd_ws_bytes(const pca_mab_shape& s);
int mab0_d256_fwd(const pca_mab_shape& s, const float* I, const void* X, const pca_mab_params& p,
                  float* H, void* saved, void* ws, hipStream_t st);
int mab0_d256_bwd(const pca_mab_shape& s, const float* I, const void* X, const pca_mab_params& p,
                  const void* saved, const float* dH, float* dI, void* dX, int dk_accumulate,
                  const pca_mab_grads& gr, void* ws, hipStream_t st, BwdDefer* defer);
int small_row_split(int B, int R);
int mab0_attn_small_launch(const float* X, const float* Gf, int B, int N, int R, int dk, float* T,
                           float* LSE, const int32_t* lengths, hipStream_t st);
int mab0_bwd_small_launch(const float* X, const float* Gf, const float* dTf, const float* LSE,
                          const float* Delta, int B, int N, int R, int Rp, int dk, float* DG,
                          const int32_t* lengths, hipStream_t st, float* slabs = nullptr);
// per-block dispatch (api_mab.hip): kind 0 exact fp32, 1 fused mab1, 2 fused mab0
int mab_kind(const pca_mab_shape& s, bool inference = false);
size_t mab_saved_bytes_any(const pca_mab_shape& s);
size_t mab_fwd_ws_bytes_any(const pca_mab_shape& s);
size_t mab_bwd_ws_bytes_any(const pca_mab_shape& s);
int mab_fwd_any(const pca_mab_shape& s, const void* Q, const void* K, const pca_mab_params& p,
                void* Y, void* saved, void* ws, hipStream_t st);
int mab_bwd_any(const pca_mab_shape& s, const void* Q, const void* K, const pca_mab_params& p,
                const void* saved, const void* dY, void* dQ, void* dK, int dk_accumulate,
                const pca_mab_grads& g, void* ws, hipStream_t st);
// fused ISAB (isab_bf16.hip)
bool isab_bf16_supported(const pca_mab_shape& s0, const pca_mab_shape& s1);
size_t isab_bf16_fwd_ws_bytes(const pca_mab_shape& s0, const pca_mab_shape& s1);
size_t isab_bf16_bwd_ws_bytes(const pca_mab_shape& s0, const pca_mab_shape& s1);
size_t isab_img_bytes();
void isab_img_carve(void* base, IsabImg* im);
void isab_collect_prep(const pca_mab_shape& s0, const pca_mab_params& p0,
                       const pca_mab_params& p1, const IsabImg& im, bool training,
                       bool need_dx, PrepJobs* J);
int isab_bf16_fwd(const pca_mab_shape& s0, const pca_mab_shape& s1, const float* I,
                  const void* X, const pca_mab_params& p0, const pca_mab_params& p1, float* H,
                  void* Y, void* saved0, void* saved1, void* ws, const IsabImg& im,
                  hipStream_t st);
int isab_bf16_bwd(const pca_mab_shape& s0, const pca_mab_shape& s1, const float* I,
                  const void* X, const float* H, const pca_mab_params& p0,
                  const pca_mab_params& p1, const void* saved0, const void* saved1,
                  const void* dY, float* dI, void* dX, const pca_mab_grads& g0,
                  const pca_mab_grads& g1, void* ws, const IsabImg& im, hipStream_t st,
                  BwdDefer* defer = nullptr);
// classifier head (train_ops.hip)
int cls_train_head(const float* P, const float* Wc, const float* bc, const int64_t* labels,
                   int B, int d, int C, float grad_scale, float* logits, float* dlogits,
                   float* dP, float* dWc, float* dbc, float* loss_out, float* stats, float* ws,
                   hipStream_t st, BwdDefer* defer = nullptr);

}  // namespace pca
