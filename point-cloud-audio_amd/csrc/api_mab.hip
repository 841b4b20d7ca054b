// extern "C" entry points for the multihead attention block: argument validation and
// dispatch on the arithmetic mode.  (set_transformer-master/modules.py:19-33)
#include "pca_common.h"

namespace pca {
int validate_shape(const pca_mab_shape* s);
size_t mab_f32_saved_bytes(const pca_mab_shape& s);
size_t mab_f32_bwd_ws_bytes(const pca_mab_shape& s);
int mab_f32_fwd(const pca_mab_shape& s, const float* Q, const float* K,
                const pca_mab_params& p, float* Y, void* saved, hipStream_t st);
int mab_f32_bwd(const pca_mab_shape& s, const float* Q, const float* K,
                const pca_mab_params& p, const void* saved, const float* dY, float* dQ,
                float* dK, int dk_accumulate, const pca_mab_grads& g, void* ws,
                hipStream_t st);
int linear_fwd_f32(const float* X, const float* W, const float* b, float* Y, int64_t M,
                   int din, int dout, hipStream_t st);
int linear_bwd_f32(const float* X, const float* W, const float* dY, float* dX, float* dW,
                   float* db, int64_t M, int din, int dout, hipStream_t st);

static int check_f32(const pca_mab_shape* s) {
  PCA_REQUIRE(s->q_dtype == PCA_F32 && s->k_dtype == PCA_F32 && s->y_dtype == PCA_F32,
              "mab: PCA_MODE_F32 needs fp32 Q, K and Y");
  return PCA_OK;
}
}  // namespace pca

extern "C" {

size_t pca_mab_saved_bytes(const pca_mab_shape* s) {
  if (pca::validate_shape(s) != PCA_OK) return 0;
  if (s->mode == PCA_MODE_F32) return pca::mab_f32_saved_bytes(*s);
  return 0;
}
size_t pca_mab_fwd_ws_bytes(const pca_mab_shape* s) {
  // inference (saved == NULL) keeps the intermediates in the scratch block instead
  return pca_mab_saved_bytes(s);
}
size_t pca_mab_bwd_ws_bytes(const pca_mab_shape* s) {
  if (pca::validate_shape(s) != PCA_OK) return 0;
  if (s->mode == PCA_MODE_F32) return pca::mab_f32_bwd_ws_bytes(*s);
  return 0;
}

int pca_mab_fwd(const pca_mab_shape* s, const void* Q, const void* K,
                const pca_mab_params* p, void* Y, void* saved, void* ws, void* stream) {
  PCA_TRY(pca::validate_shape(s));
  PCA_REQUIRE(Q && K && p && Y, "mab_fwd: null pointer");
  PCA_REQUIRE(p->wq && p->bq && p->wk && p->bk && p->wv && p->bv && p->wo && p->bo,
              "mab_fwd: null parameter");
  PCA_REQUIRE(saved || ws, "mab_fwd: need a saved block or a scratch block");
  if (s->mode == PCA_MODE_F32) {
    PCA_TRY(pca::check_f32(s));
    return pca::mab_f32_fwd(*s, (const float*)Q, (const float*)K, *p, (float*)Y,
                            saved ? saved : ws, pca::as_stream(stream));
  }
  pca::set_error("mab_fwd: mode %d not built", s->mode);
  return PCA_EUNSUPPORTED;
}

int pca_mab_bwd(const pca_mab_shape* s, const void* Q, const void* K,
                const pca_mab_params* p, const void* saved, const void* dY, void* dQ,
                void* dK, int dk_accumulate, const pca_mab_grads* g, void* ws, void* stream) {
  PCA_TRY(pca::validate_shape(s));
  PCA_REQUIRE(Q && K && p && saved && dY && g && ws, "mab_bwd: null pointer");
  PCA_REQUIRE(g->wq && g->bq && g->wk && g->bk && g->wv && g->bv && g->wo && g->bo,
              "mab_bwd: null gradient buffer");
  if (s->mode == PCA_MODE_F32) {
    PCA_TRY(pca::check_f32(s));
    return pca::mab_f32_bwd(*s, (const float*)Q, (const float*)K, *p, saved,
                            (const float*)dY, (float*)dQ, (float*)dK, dk_accumulate, *g, ws,
                            pca::as_stream(stream));
  }
  pca::set_error("mab_bwd: mode %d not built", s->mode);
  return PCA_EUNSUPPORTED;
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
