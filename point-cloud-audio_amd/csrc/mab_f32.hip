// Exact-fp32 MAB forward / backward (PCA_MODE_F32): the parity path.
//
// Restates set_transformer-master/modules.py:19-33 as a chain of strided GEMM launches
// (gemm_f32.hip) with the row softmax in between.  The reference's head split
// (`torch.cat(X.split(dh, 2), 0)`, modules.py:24-26) is pure index arithmetic here: head j
// of set b is the batch element (b, j) of a two-level strided batch over the [B, n, d]
// projection, so no head-major copy is ever made.  The learned query of ISAB / PMA
// (modules.py:52,63 `I.repeat(B,1,1)`) is projected once (q_shared) and broadcast through
// a zero batch stride.
#include "mab1_bf16.hpp"

#include <math.h>

namespace pca {

namespace {

// PCA_MODE_BF16 on a shape without fused kernels: the same chain with bf16 MFMA operands
thread_local int t_bf16_operands = 0;    // 0 exact fp32, 1 bf16 MFMA operands, 2 hi + lo bf16 pairs
struct OperandMode {
  int prev;
  explicit OperandMode(const pca_mab_shape& s) : prev(t_bf16_operands) {
    t_bf16_operands = s.mode != PCA_MODE_F32 ? 1 : 0;
  }
  ~OperandMode() { t_bf16_operands = prev; }
};
inline int gemm_sel(const pca_gemm_desc& g, const float* A, const float* B, const float* bias,
                    float* C, hipStream_t st) {
  if (t_bf16_operands == 2) return gemm_bf16_hl(g, A, B, bias, C, st);
  return t_bf16_operands ? gemm_bf16(g, A, B, bias, C, st) : gemm_f32(g, A, B, bias, C, st);
}

struct SavedF32 {
  float *Qp, *Kp, *Vp, *A, *O, *Z;
  // ln = 1 only: O1 = ln0(O), Ypre = O1 + relu(Z) (the input of ln1), row statistics
  float *O1, *Ypre, *mean0, *rstd0, *mean1, *rstd1;
};

inline size_t saved_elems(const pca_mab_shape& s, SavedF32* out, void* base) {
  const int64_t Bq = s.q_shared ? 1 : s.B;
  Carver c(base);
  SavedF32 v;
  v.Qp = c.take<float>((size_t)Bq * s.nq * s.d);
  v.Kp = c.take<float>((size_t)s.B * s.nk * s.d);
  v.Vp = c.take<float>((size_t)s.B * s.nk * s.d);
  // (the fused attention core keeps the log-sum-exp [B][h][nq] here instead of the matrix A)
  v.A = c.take<float>(attn_core_ok(s) ? attn_core_fwd_elems(s) : (size_t)s.B * s.h * s.nq * s.nk);
  v.O = c.take<float>((size_t)s.B * s.nq * s.d);
  v.Z = c.take<float>((size_t)s.B * s.nq * s.d);
  v.O1 = v.Ypre = v.mean0 = v.rstd0 = v.mean1 = v.rstd1 = nullptr;
  if (s.ln) {
    const size_t Mq = (size_t)s.B * s.nq;
    v.O1 = c.take<float>(Mq * s.d);
    v.Ypre = c.take<float>(Mq * s.d);
    v.mean0 = c.take<float>(Mq);
    v.rstd0 = c.take<float>(Mq);
    v.mean1 = c.take<float>(Mq);
    v.rstd1 = c.take<float>(Mq);
  }
  if (out) *out = v;
  return c.off;
}

struct BwdWsF32 {
  float *dZ, *dO, *dQp, *dA, *dKp, *dVp, *dQps;
  float* dYp;       // ln = 1: gradient w.r.t. the input of ln1
  float* Wt;        // [d][d]: a transposed weight (linear_dx of large bf16-mode problems)
};

inline size_t bwd_ws_elems(const pca_mab_shape& s, BwdWsF32* out, void* base) {
  Carver c(base);
  BwdWsF32 v;
  v.dZ = c.take<float>((size_t)s.B * s.nq * s.d);
  v.dO = c.take<float>((size_t)s.B * s.nq * s.d);
  v.dQp = c.take<float>((size_t)s.B * s.nq * s.d);
  v.dA = c.take<float>(attn_core_ok(s) ? attn_core_bwd_elems(s) : (size_t)s.B * s.h * s.nq * s.nk);
  v.dKp = c.take<float>((size_t)s.B * s.nk * s.d);
  v.dVp = c.take<float>((size_t)s.B * s.nk * s.d);
  v.dQps = c.take<float>((size_t)s.nq * s.d);
  v.dYp = s.ln ? c.take<float>((size_t)s.B * s.nq * s.d) : nullptr;
  const int wmax = s.dq > s.dk ? (s.dq > s.d ? s.dq : s.d) : (s.dk > s.d ? s.dk : s.d);
  v.Wt = c.take<float>((size_t)s.d * wmax);
  if (out) *out = v;
  return c.off;
}

inline pca_gemm_desc gd(int64_t M, int64_t N, int64_t K, int64_t sa_m, int64_t sa_k,
                        int64_t sb_k, int64_t sb_n, int64_t sc_m, int accumulate) {
  pca_gemm_desc g{};
  g.M = M; g.N = N; g.K = K;
  g.sa_m = sa_m; g.sa_k = sa_k; g.sb_k = sb_k; g.sb_n = sb_n; g.sc_m = sc_m;
  g.nb1 = 1; g.nb2 = 1;
  g.accumulate = accumulate;
  g.split_k = 0;
  g.alpha = 1.f;
  return g;
}

// Y[M, dout] (+)= X[M, din] W^T (+ b)
inline int linear(const float* X, const float* W, const float* b, float* Y, int64_t M,
                  int din, int dout, int accumulate, hipStream_t st) {
  // the shipped d = 64 layers over a tall activation: the weights stay in registers (linear64.hip)
  if (t_bf16_operands == 1 && !accumulate && lin64_ok(X, Y, M, din, dout))
    return lin64_fwd(X, W, b, Y, M, din, st);
  pca_gemm_desc g = gd(M, dout, din, din, 1, 1, din, dout, accumulate);
  g.split_k = 1;
  return gemm_sel(g, X, W, b, Y, st);
}
// dX[M, din] (+)= dY[M, dout] W
// Wt (optional scratch, [din][dout]): large bf16-mode problems transpose W first (one tiny
// launch), so that the weight operand is staged along k like the forward's (measured 360 ->
// 263 us at [524288, 256] x [256, 256]: the row-vector staging of B is the slower one)
inline int linear_dx(const float* dY, const float* W, float* dX, int64_t M, int din,
                     int dout, int accumulate, hipStream_t st, float* Wt = nullptr) {
  if (t_bf16_operands == 1 && din == 64 && dout == 64 && lin64_ok(dY, dX, M, 64, 64))
    return lin64_dx(dY, W, dX, M, accumulate, st);
  if (Wt != nullptr && t_bf16_operands == 1 && M * din >= (int64_t)1 << 24 && din % 4 == 0 &&
      dout % 4 == 0) {
    PCA_TRY(transpose_f32(W, Wt, dout, din, st));                 // Wt[n][k] = W[k][n]
    pca_gemm_desc g = gd(M, din, dout, dout, 1, 1, dout, din, accumulate);
    g.split_k = 1;
    return gemm_sel(g, dY, Wt, nullptr, dX, st);
  }
  pca_gemm_desc g = gd(M, din, dout, dout, 1, din, 1, din, accumulate);
  g.split_k = 1;
  return gemm_sel(g, dY, W, nullptr, dX, st);
}
// dW[dout, din] += dY[M, dout]^T X[M, din]   (split-K over the M rows)
inline int linear_dw(const float* dY, const float* X, float* dW, int64_t M, int din,
                     int dout, hipStream_t st) {
  pca_gemm_desc g = gd(dout, din, M, 1, dout, din, 1, din, 1);
  return gemm_sel(g, dY, X, nullptr, dW, st);
}

// dW += dY^T X and db += colsum(dY): one launch that reads each operand once where the shape has
// one (wgrad64.hip: 64 -> 64 layers in the bf16-operand chain), the GEMM and the column sum otherwise
inline int linear_dw_db(const float* dY, const float* X, float* dW, float* db, int64_t M, int din,
                        int dout, hipStream_t st) {
  if (t_bf16_operands == 1 && wgrad64_ok(dY, X, M, din, dout)) return wgrad64(dY, X, dW, db, M, din, st);
  PCA_TRY(linear_dw(dY, X, dW, M, din, dout, st));
  return colsum(dY, M, dout, db, 1, st);
}

inline void set_heads(pca_gemm_desc& g, const pca_mab_shape& s, int64_t a_b, int64_t a_h,
                      int64_t b_b, int64_t b_h, int64_t c_b, int64_t c_h) {
  g.nb1 = s.B; g.nb2 = s.h;
  g.sa_b1 = a_b; g.sa_b2 = a_h;
  g.sb_b1 = b_b; g.sb_b2 = b_h;
  g.sc_b1 = c_b; g.sc_b2 = c_h;
}

}  // namespace

int validate_shape(const pca_mab_shape* s) {
  PCA_REQUIRE(s != nullptr, "mab: null shape");
  PCA_REQUIRE(s->B > 0 && s->nq > 0 && s->nk > 0 && s->dq > 0 && s->dk > 0 && s->d > 0 &&
                  s->h > 0,
              "mab: non-positive extent (B=%d nq=%d nk=%d dq=%d dk=%d d=%d h=%d)", s->B,
              s->nq, s->nk, s->dq, s->dk, s->d, s->h);
  PCA_REQUIRE(s->d % s->h == 0, "mab: d=%d not divisible by h=%d", s->d, s->h);
  return PCA_OK;
}

size_t mab_f32_saved_bytes(const pca_mab_shape& s) { return saved_elems(s, nullptr, nullptr); }
size_t mab_f32_bwd_ws_bytes(const pca_mab_shape& s) { return bwd_ws_elems(s, nullptr, nullptr); }

int mab_f32_fwd(const pca_mab_shape& s, const float* Q, const float* K,
                const pca_mab_params& p, float* Y, void* saved, hipStream_t st) {
  OperandMode om(s);
  SavedF32 v;
  saved_elems(s, &v, saved);
  const int64_t Bq = s.q_shared ? 1 : s.B;
  const int d = s.d, h = s.h, dh = d / h, nq = s.nq, nk = s.nk;
  const int64_t qb = s.q_shared ? 0 : (int64_t)nq * d;   // batch stride of Qp
  const float scale = 1.0f / sqrtf((float)d);            // modules.py:28: sqrt(dim_V)

  PCA_TRY(linear(Q, p.wq, p.bq, v.Qp, Bq * nq, s.dq, d, 0, st));           // :20
  PCA_TRY(linear(K, p.wk, p.bk, v.Kp, (int64_t)s.B * nk, s.dk, d, 0, st)); // :21
  PCA_TRY(linear(K, p.wv, p.bv, v.Vp, (int64_t)s.B * nk, s.dk, d, 0, st)); // :21

  if (attn_core_ok(s)) {
    // head dim <= 16 in the bf16-operand mode: scores, softmax and A V in one launch, A never exists
    PCA_TRY(attn_core_fwd(s, v.Qp, v.Kp, v.Vp, v.O, v.A, st));             // :28-29
  } else {
  {  // S[b,j] = Qp_j Kp_j^T  -> A buffer                                       :28
    pca_gemm_desc g = gd(nq, nk, dh, d, 1, 1, d, nk, 0);
    g.split_k = 1;
    set_heads(g, s, qb, dh, (int64_t)nk * d, dh, (int64_t)h * nq * nk, (int64_t)nq * nk);
    PCA_TRY(gemm_sel(g, v.Qp, v.Kp, nullptr, v.A, st));
  }
  PCA_TRY(softmax_rows(v.A, (int64_t)s.B * h * nq, nk, scale, st, s.k_lengths,
                       (int64_t)h * nq));                                   // :28
  PCA_TRY(copy_rows(v.Qp, Bq * nq, v.O, (int64_t)s.B * nq, d, st));         // O = Q_
  {  // O_j += A_j Vp_j                                                          :29
    pca_gemm_desc g = gd(nq, dh, nk, nk, 1, d, 1, d, 1);
    g.split_k = 1;
    set_heads(g, s, (int64_t)h * nq * nk, (int64_t)nq * nk, (int64_t)nk * d, dh,
              (int64_t)nq * d, dh);
    PCA_TRY(gemm_sel(g, v.A, v.Vp, nullptr, v.O, st));
  }
  }
  const int64_t Mq = (int64_t)s.B * nq;
  const float* Oe = v.O;
  if (s.ln) {                                                               // :30
    PCA_REQUIRE(p.ln0_w && p.ln0_b && p.ln1_w && p.ln1_b, "mab: ln = 1 without LayerNorm parameters");
    PCA_TRY(layernorm_fwd(v.O, p.ln0_w, p.ln0_b, v.O1, v.mean0, v.rstd0, Mq, d, st));
    Oe = v.O1;
  }
  if (!s.ln && t_bf16_operands == 1 && d == 64 && lin64_ok(Oe, Y, Mq, d, d) && lin64_ok(Oe, v.Z, Mq, d, d))
    return lin64_fc_o(Oe, p.wo, p.bo, v.Z, Y, Mq, st);                      // :31, one launch
  PCA_TRY(linear(Oe, p.wo, p.bo, v.Z, Mq, d, d, 0, st));                    // :31
  if (s.ln) {
    PCA_TRY(add_relu(Oe, v.Z, v.Ypre, Mq * d, st));                         // :31
    PCA_TRY(layernorm_fwd(v.Ypre, p.ln1_w, p.ln1_b, Y, v.mean1, v.rstd1, Mq, d, st));   // :32
  } else {
    PCA_TRY(add_relu(Oe, v.Z, Y, Mq * d, st));                              // :31
  }
  return PCA_OK;
}

int mab_f32_bwd(const pca_mab_shape& s, const float* Q, const float* K,
                const pca_mab_params& p, const void* saved, const float* dY, float* dQ,
                float* dK, int dk_accumulate, const pca_mab_grads& g, void* ws,
                hipStream_t st) {
  OperandMode om(s);
  SavedF32 v;
  saved_elems(s, &v, const_cast<void*>(saved));
  BwdWsF32 w;
  bwd_ws_elems(s, &w, ws);
  const int d = s.d, h = s.h, dh = d / h, nq = s.nq, nk = s.nk;
  const int64_t Mq = (int64_t)s.B * nq, Mk = (int64_t)s.B * nk;
  const int64_t qb = s.q_shared ? 0 : (int64_t)nq * d;
  const float scale = 1.0f / sqrtf((float)d);

  // ln1, then O + relu(fc_o(O)), then ln0 (modules.py:30-32 backwards)
  const float* dYe = dY;
  const float* Oe = v.O;
  if (s.ln) {
    PCA_REQUIRE(g.ln0_w && g.ln0_b && g.ln1_w && g.ln1_b, "mab: ln = 1 without LayerNorm gradients");
    PCA_TRY(layernorm_bwd(dY, v.Ypre, v.mean1, v.rstd1, p.ln1_w, w.dYp, g.ln1_w, g.ln1_b, Mq, d, st));
    dYe = w.dYp;
    Oe = v.O1;
  }
  if (!s.ln && t_bf16_operands == 1 && d == 64 && lin64_ok(dYe, w.dO, Mq, d, d) &&
      lin64_ok(v.Z, w.dZ, Mq, d, d)) {
    // dZ = dY . [Z > 0] and dO = dY + dZ Wo in one launch (linear64.hip)
    PCA_TRY(lin64_fc_o_bwd(dYe, v.Z, p.wo, w.dZ, w.dO, Mq, st));
    PCA_TRY(linear_dw_db(w.dZ, Oe, g.wo, g.bo, Mq, d, d, st));
  } else {
    PCA_TRY(relu_bwd(dYe, v.Z, w.dZ, Mq * d, st));
    PCA_TRY(linear_dw_db(w.dZ, Oe, g.wo, g.bo, Mq, d, d, st));
    PCA_TRY(copy_rows(dYe, Mq, w.dO, Mq, d, st));
    PCA_TRY(linear_dx(w.dZ, p.wo, w.dO, Mq, d, d, 1, st, w.Wt));
  }
  if (s.ln)     // in place: every element is read before it is rewritten
    PCA_TRY(layernorm_bwd(w.dO, v.O, v.mean0, v.rstd0, p.ln0_w, w.dO, g.ln0_w, g.ln0_b, Mq, d, st));

  if (attn_core_ok(s)) {
    // the adjoint of the fused core: P recomputed from the saved log-sum-exp, dQp (incl. the residual),
    // dKp and dVp written once each (w.dA holds delta = rowdot(dO, A V))
    PCA_TRY(attn_core_bwd(s, v.Qp, v.Kp, v.Vp, v.O, v.A, w.dO, w.dQp, w.dKp, w.dVp, w.dA, st));
  } else {
  // attention: dV_j = A_j^T dO_j
  PCA_TRY(fill_zero(w.dVp, Mk * d, st));
  PCA_TRY(fill_zero(w.dKp, Mk * d, st));
  {
    pca_gemm_desc gg = gd(nk, dh, nq, 1, nk, d, 1, d, 1);
    set_heads(gg, s, (int64_t)h * nq * nk, (int64_t)nq * nk, (int64_t)nq * d, dh,
              (int64_t)nk * d, dh);
    PCA_TRY(gemm_sel(gg, v.A, w.dO, nullptr, w.dVp, st));
  }
  {  // dA = dO_j Vp_j^T
    pca_gemm_desc gg = gd(nq, nk, dh, d, 1, 1, d, nk, 0);
    gg.split_k = 1;
    set_heads(gg, s, (int64_t)nq * d, dh, (int64_t)nk * d, dh, (int64_t)h * nq * nk,
              (int64_t)nq * nk);
    PCA_TRY(gemm_sel(gg, w.dO, v.Vp, nullptr, w.dA, st));
  }
  PCA_TRY(softmax_bwd_rows(v.A, w.dA, (int64_t)s.B * h * nq, nk, scale, st));  // dS*scale
  PCA_TRY(copy_rows(w.dO, Mq, w.dQp, Mq, d, st));                              // residual Q_
  {  // dQp_j += dS Kp_j
    pca_gemm_desc gg = gd(nq, dh, nk, nk, 1, d, 1, d, 1);
    gg.split_k = 1;
    set_heads(gg, s, (int64_t)h * nq * nk, (int64_t)nq * nk, (int64_t)nk * d, dh,
              (int64_t)nq * d, dh);
    PCA_TRY(gemm_sel(gg, w.dA, v.Kp, nullptr, w.dQp, st));
  }
  {  // dKp_j = dS^T Qp_j
    pca_gemm_desc gg = gd(nk, dh, nq, 1, nk, d, 1, d, 1);
    set_heads(gg, s, (int64_t)h * nq * nk, (int64_t)nq * nk, qb, dh, (int64_t)nk * d, dh);
    PCA_TRY(gemm_sel(gg, w.dA, v.Qp, nullptr, w.dKp, st));
  }
  }

  // fc_k / fc_v
  PCA_TRY(linear_dw_db(w.dKp, K, g.wk, g.bk, Mk, s.dk, d, st));
  PCA_TRY(linear_dw_db(w.dVp, K, g.wv, g.bv, Mk, s.dk, d, st));
  if (dK != nullptr) {
    PCA_TRY(linear_dx(w.dKp, p.wk, dK, Mk, s.dk, d, dk_accumulate ? 1 : 0, st, w.Wt));
    PCA_TRY(linear_dx(w.dVp, p.wv, dK, Mk, s.dk, d, 1, st, w.Wt));
  }

  // fc_q
  if (s.q_shared) {
    PCA_TRY(colsum(w.dQp, s.B, nq * d, w.dQps, 0, st));   // sum over sets
    PCA_TRY(linear_dw_db(w.dQps, Q, g.wq, g.bq, nq, s.dq, d, st));
    if (dQ != nullptr) PCA_TRY(linear_dx(w.dQps, p.wq, dQ, nq, s.dq, d, 1, st));
  } else {
    PCA_TRY(linear_dw_db(w.dQp, Q, g.wq, g.bq, Mq, s.dq, d, st));
    if (dQ != nullptr) PCA_TRY(linear_dx(w.dQp, p.wq, dQ, Mq, s.dq, d, 0, st, w.Wt));
  }
  return PCA_OK;
}

// Scope guard for callers outside this file (the d = 256 blocks): the linear_* helpers below run
// their GEMMs with bf16 MFMA operands while one is alive on the calling thread.
Bf16OperandScope::Bf16OperandScope(int mode) : prev(t_bf16_operands) { t_bf16_operands = mode; }
Bf16OperandScope::~Bf16OperandScope() { t_bf16_operands = prev; }

// ---- classifier head ----------------------------------------------------------
int linear_fwd_f32(const float* X, const float* W, const float* b, float* Y, int64_t M,
                   int din, int dout, hipStream_t st) {
  return linear(X, W, b, Y, M, din, dout, 0, st);
}
int linear_dx_acc_f32(const float* dY, const float* W, float* dX, int64_t M, int din, int dout,
                      int accumulate, hipStream_t st) {
  return linear_dx(dY, W, dX, M, din, dout, accumulate, st);
}
int linear_bwd_f32(const float* X, const float* W, const float* dY, float* dX, float* dW,
                   float* db, int64_t M, int din, int dout, hipStream_t st) {
  if (dW && db) {
    PCA_TRY(linear_dw_db(dY, X, dW, db, M, din, dout, st));
  } else {
    if (dW) PCA_TRY(linear_dw(dY, X, dW, M, din, dout, st));
    if (db) PCA_TRY(colsum(dY, M, dout, db, 1, st));
  }
  if (dX) PCA_TRY(linear_dx(dY, W, dX, M, din, dout, 0, st));
  return PCA_OK;
}

}  // namespace pca
