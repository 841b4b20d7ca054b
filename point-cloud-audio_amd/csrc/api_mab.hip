// extern "C" entry points for the multihead attention block: argument validation and
// dispatch on the arithmetic mode.  (set_transformer-master/modules.py:19-33)
#include "mab1_bf16.hpp"

namespace pca {



static int check_f32(const pca_mab_shape* s, bool inference) {
  // the exact path exchanges fp32 only; the fused kernels validate their own dtypes
  PCA_REQUIRE(mab_kind(*s, inference) != 0 || (s->q_dtype == PCA_F32 && s->k_dtype == PCA_F32 &&
                                    s->y_dtype == PCA_F32),
              "mab: the exact fp32 path needs fp32 Q, K and Y");
  return PCA_OK;
}

// ---- mode resolution + dispatch, shared by the C entry points and the ST engine ----
// kind: 0 = exact fp32 chain of GEMMs, 1 = fused bf16 mab1 (many queries, few keys),
//       2 = fused bf16 mab0 (few shared queries, many keys),
//       3 = the shipped d = 64 / 8-head shape, fused fp32 forward (inference only: sd64_fwd.hip)
int mab_kind(const pca_mab_shape& s, bool inference) {
  // (a fused mab1 has the m inducing-point outputs as keys: always all of them; a caller that
  // masks keys of such a shape gets the exact path, whose softmax honours k_lengths)
  if (s.ln) return 0;          // LayerNorm variants: exact chain only
  const bool fused_mode = s.mode == PCA_MODE_BF16 || s.mode == PCA_MODE_FP8;
  if (fused_mode && s.k_lengths == nullptr && mab1_bf16_supported(s, inference)) return 1;
  if (fused_mode && mab0_bf16_supported(s)) return 2;
  if (fused_mode && inference && sd64_kind(s) != 0) return 3;
  return 0;
}
size_t mab_saved_bytes_any(const pca_mab_shape& s) {
  const int k = mab_kind(s);
  return k == 1 ? mab1_bf16_saved_bytes(s) : k == 2 ? mab0_bf16_saved_bytes(s)
                                                    : mab_f32_saved_bytes(s);
}
size_t mab_fwd_ws_bytes_any(const pca_mab_shape& s) {
  // inference (saved == NULL) keeps the intermediates in the scratch block instead; a training
  // forward of the same shape never needs more (the exact chain's scratch is its saved block)
  const int k = mab_kind(s, true);
  if (k != mab_kind(s, false)) {
    const size_t a = k == 3 ? sd64_fwd_ws_bytes(s)
                            : k == 1 ? mab1_bf16_fwd_ws_bytes(s) : mab0_bf16_fwd_ws_bytes(s);
    const size_t b = mab_f32_saved_bytes(s);
    return a > b ? a : b;
  }
  return k == 1 ? mab1_bf16_fwd_ws_bytes(s) : k == 2 ? mab0_bf16_fwd_ws_bytes(s)
                                                     : mab_f32_saved_bytes(s);
}
size_t mab_bwd_ws_bytes_any(const pca_mab_shape& s) {
  const int k = mab_kind(s);
  return k == 1 ? mab1_bf16_bwd_ws_bytes(s) : k == 2 ? mab0_bf16_bwd_ws_bytes(s)
                                                     : mab_f32_bwd_ws_bytes(s);
}
int mab_fwd_any(const pca_mab_shape& s, const void* Q, const void* K, const pca_mab_params& p,
                void* Y, void* saved, void* ws, hipStream_t st) {
  const int k = mab_kind(s, saved == nullptr);
  if (k == 3) return sd64_fwd(s, (const float*)Q, (const float*)K, p, (float*)Y, ws, st);
  if (k == 1) return mab1_bf16_fwd(s, Q, (const float*)K, p, Y, saved, ws, st);
  if (k == 2) return mab0_bf16_fwd(s, (const float*)Q, K, p, (float*)Y, saved, ws, st);
  return mab_f32_fwd(s, (const float*)Q, (const float*)K, p, (float*)Y, saved ? saved : ws, st);
}
int mab_bwd_any(const pca_mab_shape& s, const void* Q, const void* K, const pca_mab_params& p,
                const void* saved, const void* dY, void* dQ, void* dK, int dk_accumulate,
                const pca_mab_grads& g, void* ws, hipStream_t st) {
  const int k = mab_kind(s);
  if (k == 2)
    return mab0_bf16_bwd(s, (const float*)Q, K, p, saved, (const float*)dY, (float*)dQ, dK,
                         dk_accumulate, g, ws, st);
  if (k == 1)
    return mab1_bf16_bwd(s, Q, (const float*)K, p, saved, dY, dQ, (float*)dK, dk_accumulate, g,
                         ws, st);
  return mab_f32_bwd(s, (const float*)Q, (const float*)K, p, saved, (const float*)dY,
                     (float*)dQ, (float*)dK, dk_accumulate, g, ws, st);
}
}  // namespace pca

namespace pca {
int handoffs_empty(const char* where, bool pack_allowed) {
  PCA_REQUIRE(!mid256_pending(), "%s: a d = 256 mid-stage hand-off is pending on this thread", where);
  PCA_REQUIRE(!mab0_d256_prep_pending(), "%s: a query-side preparation flag is pending on this thread", where);
  PCA_REQUIRE(!weight_images_active(), "%s: a weight-image table is still registered on this thread", where);
  PCA_REQUIRE(!wgrad256_handoff_pending(), "%s: a weight-gradient job hand-over is pending on this thread",
              where);
  PCA_REQUIRE(pack_allowed || !pack_pending(),
              "%s: a deferred pack is pending on this thread (only pca_st_forward / pca_st_train_fwd_bwd "
              "consume it)", where);
  return PCA_OK;
}
}  // namespace pca

extern "C" {

// An explicit PCA_MODE_BF16 request must be served by a fused kernel (no silent change of
// arithmetic at this level); callers that want "bf16 where available" query
// pca_mab_saved_bytes() first, which returns 0 for unsupported bf16 shapes.
static int bf16_demand(const pca_mab_shape* s, bool inference = false) {
  if ((s->mode == PCA_MODE_BF16 || s->mode == PCA_MODE_FP8) && pca::mab_kind(*s, inference) == 0) {
    pca::set_error("mab: no bf16 / fp8 kernel for B=%d nq=%d nk=%d dq=%d dk=%d d=%d h=%d q_shared=%d",
                   s->B, s->nq, s->nk, s->dq, s->dk, s->d, s->h, s->q_shared);
    return PCA_EUNSUPPORTED;
  }
  if (s->mode != PCA_MODE_BF16 && s->mode != PCA_MODE_F32 && s->mode != PCA_MODE_FP8) {
    pca::set_error("mab: unknown mode %d", s->mode);
    return PCA_EINVAL;
  }
  return PCA_OK;
}

size_t pca_mab_saved_bytes(const pca_mab_shape* s) {
  if (pca::validate_shape(s) != PCA_OK || bf16_demand(s) != PCA_OK) return 0;
  return pca::mab_saved_bytes_any(*s);
}
size_t pca_mab_fwd_ws_bytes(const pca_mab_shape* s) {
  if (pca::validate_shape(s) != PCA_OK || bf16_demand(s, true) != PCA_OK) return 0;
  return pca::mab_fwd_ws_bytes_any(*s);
}
size_t pca_mab_bwd_ws_bytes(const pca_mab_shape* s) {
  if (pca::validate_shape(s) != PCA_OK || bf16_demand(s) != PCA_OK) return 0;
  return pca::mab_bwd_ws_bytes_any(*s);
}


// runs the body, then re-checks the hand-offs: nothing may be left behind by this call either
#define PCA_WITH_HANDOFF_CHECK(where, pack_allowed, call)            \
  do {                                                               \
    PCA_TRY(pca::handoffs_empty(where, pack_allowed));              \
    const int rc_ = (call);                                          \
    if (rc_ != PCA_OK) return rc_;                                   \
    return pca::handoffs_empty(where " (exit)", false);             \
  } while (0)

int pca_mab_fwd(const pca_mab_shape* s, const void* Q, const void* K,
                const pca_mab_params* p, void* Y, void* saved, void* ws, void* stream) {
  PCA_TRY(pca::validate_shape(s));
  PCA_REQUIRE(Q && K && p && Y, "mab_fwd: null pointer");
  PCA_REQUIRE(p->wq && p->bq && p->wk && p->bk && p->wv && p->bv && p->wo && p->bo,
              "mab_fwd: null parameter");
  PCA_TRY(bf16_demand(s, saved == nullptr));
  PCA_TRY(pca::check_f32(s, saved == nullptr));
  PCA_REQUIRE(ws != nullptr || (saved != nullptr && pca::mab_kind(*s, false) == 0),
              "mab_fwd: scratch block required");
  PCA_WITH_HANDOFF_CHECK("pca_mab_fwd", false,
                         pca::mab_fwd_any(*s, Q, K, *p, Y, saved, ws, pca::as_stream(stream)));
}

int pca_mab_bwd(const pca_mab_shape* s, const void* Q, const void* K,
                const pca_mab_params* p, const void* saved, const void* dY, void* dQ,
                void* dK, int dk_accumulate, const pca_mab_grads* g, void* ws, void* stream) {
  PCA_TRY(pca::validate_shape(s));
  PCA_REQUIRE(Q && K && p && saved && dY && g && ws, "mab_bwd: null pointer");
  PCA_REQUIRE(g->wq && g->bq && g->wk && g->bk && g->wv && g->bv && g->wo && g->bo,
              "mab_bwd: null gradient buffer");
  PCA_TRY(bf16_demand(s));
  PCA_TRY(pca::check_f32(s, false));
  PCA_WITH_HANDOFF_CHECK("pca_mab_bwd", false,
                         pca::mab_bwd_any(*s, Q, K, *p, saved, dY, dQ, dK, dk_accumulate, *g, ws,
                                          pca::as_stream(stream)));
}

int pca_linear_fwd(const float* X, const float* W, const float* b, float* Y, int64_t M,
                   int din, int dout, void* stream) {
  PCA_REQUIRE(X && W && b && Y && M > 0 && din > 0 && dout > 0, "linear_fwd: bad arguments");
  return pca::linear_fwd_f32(X, W, b, Y, M, din, dout, pca::as_stream(stream));
}
size_t pca_linear_bwd_ws_bytes(int64_t, int, int) { return 256; }
int pca_linear_bwd(const float* X, const float* W, const float* dY, float* dX, float* dW,
                   float* db, int64_t M, int din, int dout, void* ws, void* stream) {
  (void)ws;
  PCA_REQUIRE(X && W && dY && M > 0 && din > 0 && dout > 0, "linear_bwd: bad arguments");
  return pca::linear_bwd_f32(X, W, dY, dX, dW, db, M, din, dout, pca::as_stream(stream));
}
}
