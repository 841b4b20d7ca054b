// Host orchestration of the d = 256 / 8-head training blocks (kernels: d256_bf16.hip).
//   mab1_d256_bwd     adjoint of ISAB's mab1(X, H) (modules.py:53 / 19-33): three row-GEMM /
//                     attention launches + the 256-wide weight-gradient reduction
//   mab0_d256_*       the few-shared-queries block (ISAB mab0, PMA; modules.py:52,63):
//                       dk = 256, R = h*m <= 16 (PMA): reassociated, X read once (k_pma_*256)
//                       dk = 256, more queries: keys projected (reference formulation), flash
//                                  attention per head
//                       dk <= 4  : reassociated layer-1 kernels (k_mab0_attn_small / _bwd_small)
//                     with the per-set [B*m]-row epilogues on the GEMM kernel.
// Activations cross the ABI in fp32 or bf16 (shape.*_dtype); inside, every [B*N, 256] tensor is
// bf16 - fp32 callers pay one conversion pass per tensor.
#include "d256_bf16.hpp"

#include <math.h>

namespace pca {

namespace {
constexpr int D = 256, H8 = 8;

inline size_t nelem(const pca_mab_shape& s, bool keys) {
  return (size_t)s.B * (keys ? s.nk : s.nq) * D;
}
}  // namespace

// =====================================================================================
// many-queries block, backward
// =====================================================================================
struct Mab1D256Ws {
  __bf16 *WoTP, *WqTP;
  __bf16 *dZ, *dO, *dQp;        // [B*N][D]
  __bf16 *dYb, *Xb, *dXb;       // conversions of fp32 tensors at the ABI (null when bf16)
  float *dKp, *dVp, *dKpPart, *dVpPart;
  void* wg;
};
// Hand-over of mab1's fc_q weight-gradient job ({dQp, X}) to the few-queries block of the same ISAB,
// whose two jobs ({dKp, X}, {dVp, X}) read the same X: launched together, the three share each X tile
// through the XCD's L2 (k_wgrad256's shared-operand order).  Armed by the engine only, around the
// pair of calls, with separate workspaces for the two blocks (dQp must outlive mab1's call).
namespace {
struct WgHandoff {
  bool armed = false, has = false, has_dx = false;
  Wgrad256Job job{};
  DxHandoff dx{};
};
thread_local WgHandoff g_handoff;
}  // namespace
namespace {
struct Mid256Handoff {
  bool armed = false, ready = false;
  pca_mab_shape s1{};
  pca_mab_params p1{};
  void* saved1 = nullptr;
};
thread_local Mid256Handoff g_mid256;
}  // namespace
void mid256_arm(const pca_mab_shape* s1, const pca_mab_params* p1, void* saved1) {
  g_mid256.ready = false;
  g_mid256.armed = s1 != nullptr && p1 != nullptr && saved1 != nullptr;
  if (g_mid256.armed) { g_mid256.s1 = *s1; g_mid256.p1 = *p1; g_mid256.saved1 = saved1; }
}
bool mid256_pending() { return g_mid256.armed || g_mid256.ready; }
bool mid256_kv_ready() {
  const bool r = g_mid256.ready;
  g_mid256.ready = false;
  return r;
}
// PCA_D256_MID=0: the five-launch form of round 2 (A/B measurements)
static bool mid256_on() {
  static const bool on = [] { const char* e = getenv("PCA_D256_MID"); return !(e && e[0] == '0'); }();
  return on;
}
void wgrad256_handoff_arm(bool on) {
  g_handoff.armed = on;
  if (!on) { g_handoff.has = false; g_handoff.has_dx = false; }
}
bool wgrad256_handoff_pending() { return g_handoff.has || g_handoff.has_dx; }

// PCA_ROWSTREAM=0: the LDS-resident-weight row-GEMMs (k_rowgemm) instead of the register-resident
// streaming ones (d256_stream.hip), for A/B measurements
static bool rowstream_on() {
  return true;
}

static size_t mab1_d256_carve(const pca_mab_shape& s, Mab1D256Ws* out, void* base) {
  Carver c(base);
  Mab1D256Ws w{};
  const size_t M = (size_t)s.B * s.nq;
  const bool abf = s.y_dtype == PCA_BF16;
  w.WoTP = c.take<__bf16>((size_t)D * D);
  w.WqTP = c.take<__bf16>((size_t)D * D);
  w.dZ = c.take<__bf16>(M * D);
  w.dO = c.take<__bf16>(M * D);
  w.dQp = c.take<__bf16>(M * D);
  if (!abf) {
    w.dYb = c.take<__bf16>(M * D);
    w.dXb = c.take<__bf16>(M * D);
    if (s.dq == D) w.Xb = c.take<__bf16>(M * D);
  }
  const size_t parts = (size_t)attn1_bwd256_parts(s.B, s.nq);
  w.dKp = c.take<float>((size_t)s.B * s.nk * D);
  w.dVp = c.take<float>((size_t)s.B * s.nk * D);
  w.dKpPart = c.take<float>((size_t)s.B * parts * s.nk * D);
  w.dVpPart = c.take<float>((size_t)s.B * parts * s.nk * D);
  {
    // serves the long {dZ, O} / {dQp, X} jobs ([B nq] rows) AND the short fc_k / fc_v ones ([B nk])
    const int64_t Mk = (int64_t)s.B * s.nk, Mw = (int64_t)M > Mk ? (int64_t)M : Mk;
    const size_t a = wgrad256_ws_bytes(2, Mw), b = wgrad_small256_ws_bytes((int64_t)M);
    w.wg = c.take<char>(a > b ? a : b);      // (the two reductions run one after the other)
  }
  if (out) *out = w;
  return c.off;
}
// mab1's deferred dX = dQp Wq, written (not accumulated) now
static int wgrad256_handoff_flush_dx(hipStream_t st) {
  if (!g_handoff.has_dx) return PCA_OK;
  g_handoff.has_dx = false;
  return rowstream256_dx1(g_handoff.dx.dQp, g_handoff.dx.WqT, g_handoff.dx.dX, g_handoff.dx.B,
                          g_handoff.dx.N, st);
}
// a handed-over job nobody took (the following block was not the projected-keys few-queries one)
int wgrad256_handoff_flush(void* ws, hipStream_t st) {
  PCA_TRY(wgrad256_handoff_flush_dx(st));
  if (!g_handoff.has) return PCA_OK;
  Wgrad256Jobs jobs{};
  jobs.j[jobs.n++] = g_handoff.job;
  g_handoff.has = false;
  return wgrad256_launch(jobs, ws, st);
}

size_t mab1_d256_bwd_ws_bytes(const pca_mab_shape& s) { return mab1_d256_carve(s, nullptr, nullptr); }

// [B*m]-row jobs (fp32 operands): queued in `defer` when the caller collects them (one launch for
// all blocks at the end of the backward: bwd_defer_flush), else launched now
static int wgrad256_short(const Wgrad256Jobs& jobs, void* ws, BwdDefer* defer, hipStream_t st) {
  if (defer != nullptr && defer->wg256_ws != nullptr && defer->wg256_n + jobs.n <= 8) {
    for (int i = 0; i < jobs.n; ++i) {
      const Wgrad256Job& j = jobs.j[i];
      defer->wg256[defer->wg256_n++] = BwdDefer::Wg256{j.G, j.A, j.dW, j.db, j.M};
    }
    return PCA_OK;
  }
  return wgrad256_launch_t(jobs, ws, true, st);
}
int wgrad256_flush_deferred(BwdDefer& D, hipStream_t st) {
  Wgrad256Jobs J{};
  for (int i = 0; i < D.wg256_n; ++i)
    J.j[J.n++] = Wgrad256Job{D.wg256[i].G, D.wg256[i].A, D.wg256[i].dW, D.wg256[i].db, D.wg256[i].M};
  D.wg256_n = 0;
  return wgrad256_launch_t(J, D.wg256_ws, true, st);
}

// PCA_BWD_O_FUSED=0: fc_o adjoint as its own row-GEMM launch (A/B measurements)
static bool fuse_o_on() {
  return true;
}
int d256_bwd_wo_mode() { return fuse_o_on() ? 3 : 2; }
int d256_bwd_wq_mode() { return rowstream_on() ? 3 : 2; }
int d256_bwd_kv_mode() { return rowstream_on() ? 3 : 2; }

int mab1_d256_bwd(const pca_mab_shape& s, const void* X, const float* Hk, const pca_mab_params& p,
                  const void* saved, const void* dY, void* dX, float* dH, int dk_accumulate,
                  const pca_mab_grads& gr, void* ws, hipStream_t st, BwdDefer* defer) {
  PCA_REQUIRE(s.d == D && s.nk == 32 && s.h == H8, "mab1_d256_bwd: shape");
  Bf16OperandScope ops(1);             // the [B*m]-row GEMMs of the K / V tail: MFMA
  Mab1Saved v;
  mab1_carve_saved(s, &v, const_cast<void*>(saved));
  Mab1D256Ws w;
  mab1_d256_carve(s, &w, ws);
  const int64_t M = (int64_t)s.B * s.nq;
  const bool small = s.dq <= 4, abf = s.y_dtype == PCA_BF16;
  const bool want_dx = dX != nullptr;
  if (want_dx && small) {
    set_error("mab1_bf16_bwd: dQ for dq <= 4 is not built");
    return PCA_EUNSUPPORTED;
  }
  // PCA_BWD_O_FUSED=0: fc_o adjoint as its own row-GEMM launch (A/B measurements)
  const bool fuse_o = fuse_o_on();
  if (want_dx)
    PCA_TRY(weight_image2(p.wo, &w.WoTP, d256_bwd_wo_mode(), p.wq, &w.WqTP, d256_bwd_wq_mode(), D, D, st));
  else
    PCA_TRY(weight_image1(p.wo, &w.WoTP, D, D, d256_bwd_wo_mode(), st));
  const __bf16* dYb = reinterpret_cast<const __bf16*>(dY);
  const __bf16* Xb = small ? nullptr : reinterpret_cast<const __bf16*>(X);
  if (!abf) {
    PCA_TRY(cvt_f32_bf16(reinterpret_cast<const float*>(dY), w.dYb, M * D, st));
    dYb = w.dYb;
    if (!small) {
      PCA_TRY(cvt_f32_bf16(reinterpret_cast<const float*>(X), w.Xb, M * D, st));
      Xb = w.Xb;
    }
  }
  // reference-formulation FLOPs / algorithmic bytes of the launches inside the scope (round 3: the
  // scope closes after the fc_o + attention adjoint - dX runs in k_rowstream, every weight gradient
  // in k_wgrad256, each with a scope of its own or none): dO = dY + dZ Wo (2 M d^2) and the
  // attention adjoint (8 M m d); dY and Qp (layer 1: the points) in, dZ and dQp out
  const double flops = (double)M * (2.0 * D * D + 8.0 * s.nk * D);
  // dZ = dY . [Z > 0] is read by the fc_o weight-gradient job only: with whole 128-point mask blocks per
  // set that job takes dY and the mask words instead (Wgrad256Job::mask) and dZ is never written
  const bool dz_masked = fuse_o && wgrad256_masked_ok(s.nq);
  const double bytes = (double)M * (2.0 * D + (small ? 4.0 * s.dq : 2.0 * D) + (dz_masked ? 2.0 : 4.0) * D);
  ProfScope ps(PCA_K_MAB1_BWD, st, flops, bytes);
  if (fuse_o) {
    // layer 1 (dq <= 4): Qp is recomputed from the points (mab1_saves_qp() == false: not saved)
    PCA_TRY(attn1_bwd256_fused(dYb, v.mask, w.WoTP, mab1_saves_qp(s) ? v.QpS : nullptr, v.KpP,
                               v.VpP, v.Kt, dz_masked ? nullptr : w.dZ, w.dQp, w.dKpPart, w.dVpPart, w.dKp, w.dVp, s.B,
                               s.nq, st, reinterpret_cast<const float*>(X), p.wq, p.bq, s.dq));
  } else {
    PCA_TRY(rowgemm256_bwd_o(dYb, v.mask, w.WoTP, w.dZ, w.dO, s.B, s.nq, st));
    PCA_TRY(attn1_bwd256(w.dO, v.QpS, v.KpP, v.VpP, v.Kt, w.dQp, w.dKpPart, w.dVpPart, w.dKp,
                         w.dVp, s.B, s.nq, st));
  }
  ps.end();
  if (want_dx) {
    __bf16* dXb = abf ? reinterpret_cast<__bf16*>(dX) : w.dXb;
    if (rowstream_on() && g_handoff.armed && abf) {
      // deferred: the few-queries block of this ISAB adds its dKp Wk + dVp Wv in the same pass (DX3)
      g_handoff.dx = DxHandoff{w.dQp, w.WqTP, dXb, s.B, s.nq};
      g_handoff.has_dx = true;
    } else if (rowstream_on()) {
      PCA_TRY(rowstream256_dx1(w.dQp, w.WqTP, dXb, s.B, s.nq, st));
    } else {
      PCA_TRY(rowgemm256_dx(w.dQp, w.WqTP, dXb, s.B, s.nq, 0, st));
    }
    if (!abf) PCA_TRY(cvt_bf16_f32(dXb, reinterpret_cast<float*>(dX), M * D, 0, st));
  }
  // weight gradients over the B*N rows
  Wgrad256Jobs jobs{};
  if (dz_masked) jobs.j[jobs.n++] = Wgrad256Job{dYb, v.OS, gr.wo, gr.bo, M, v.mask};
  else jobs.j[jobs.n++] = Wgrad256Job{w.dZ, v.OS, gr.wo, gr.bo, M};
  if (!small) {
    const Wgrad256Job jq{w.dQp, Xb, gr.wq, gr.bq, M};
    if (g_handoff.armed && abf) {          // launched by the few-queries block with its own two jobs
      g_handoff.job = jq;
      g_handoff.has = true;
    } else {
      jobs.j[jobs.n++] = jq;
    }
  }
  PCA_TRY(wgrad256_launch(jobs, w.wg, st));
  if (small)
    PCA_TRY(wgrad_small256(w.dQp, reinterpret_cast<const float*>(X), M, s.dq, gr.wq, gr.bq, w.wg, st));
  // fc_k / fc_v of the m inducing-point outputs ([B*m] rows) and dH
  const int64_t Mk = (int64_t)s.B * s.nk;
  {
    Wgrad256Jobs kv{};
    kv.j[kv.n++] = Wgrad256Job{w.dKp, Hk, gr.wk, gr.bk, Mk};
    kv.j[kv.n++] = Wgrad256Job{w.dVp, Hk, gr.wv, gr.bv, Mk};
    PCA_TRY(wgrad256_short(kv, w.wg, defer, st));
  }
  if (dH != nullptr) {
    PCA_TRY(linear_dx_acc_f32(w.dKp, p.wk, dH, Mk, D, D, dk_accumulate ? 1 : 0, st));
    PCA_TRY(linear_dx_acc_f32(w.dVp, p.wv, dH, Mk, D, D, 1, st));
  }
  return PCA_OK;
}

// =====================================================================================
// few-queries block
// =====================================================================================
bool mab0_d256_supported(const pca_mab_shape& s) {
  const int R = s.h * s.nq;
  const bool dt_ok = s.q_dtype == PCA_F32 && s.y_dtype == PCA_F32 &&
                     (s.k_dtype == PCA_F32 || (s.k_dtype == PCA_BF16 && s.dk == D));
  if (!(s.q_shared == 1 && s.d == D && s.h == H8 && s.dq == D && dt_ok)) return false;
  if (s.dk == D) return s.nq >= 1 && s.nq <= 32;
  return s.dk <= 4 && (R == 64 || R == 128 || R == 256);
}

// which of the three few-queries paths serves a shape
enum { FQ_SMALL = 0, FQ_PMA = 1, FQ_PROJ = 2 };
static inline int fq_path(const pca_mab_shape& s) {
  return s.dk != D ? FQ_SMALL : (s.h * s.nq <= 16 ? FQ_PMA : FQ_PROJ);
}

struct Fq256Saved {
  float *Qp, *O, *Z, *LSE;      // [m][D], [B][m][D] x2, [B][8][MQ] (reassociated paths: [B][R])
  __bf16 *Kp, *Vp, *Xb;         // projected keys: [B*N][D]; Xb only when the keys arrive in fp32
  float *Gf, *T;                // reassociated paths: [Rp][dk], [B][R][dk]
  __bf16* Gb;                   // PMA: [32][D]
  float *Op, *Mp, *Lp;          // forward partials per point range
};
static size_t fq_carve_saved(const pca_mab_shape& s, Fq256Saved* out, void* base) {
  Carver c(base);
  Fq256Saved v{};
  const int m = s.nq, R = s.h * m;
  const size_t Bm = (size_t)s.B * m;
  v.Qp = c.take<float>((size_t)m * D);
  v.O = c.take<float>(Bm * D);
  v.Z = c.take<float>(Bm * D);
  if (fq_path(s) == FQ_PROJ) {
    const int MQ = m > 16 ? 32 : 16, S = fq_splits256(s.B, s.nk);
    v.LSE = c.take<float>((size_t)s.B * H8 * MQ);
    v.Kp = c.take<__bf16>(nelem(s, true));
    v.Vp = c.take<__bf16>(nelem(s, true));
    if (s.k_dtype == PCA_F32) v.Xb = c.take<__bf16>(nelem(s, true));
    v.Op = c.take<float>((size_t)s.B * S * m * D);
    v.Mp = c.take<float>((size_t)s.B * S * H8 * MQ);
    v.Lp = c.take<float>((size_t)s.B * S * H8 * MQ);
  } else if (fq_path(s) == FQ_PMA) {
    const int S = pma_splits256(s.B, s.nk);
    v.LSE = c.take<float>((size_t)s.B * R);
    v.Gf = c.take<float>((size_t)32 * D);
    v.Gb = c.take<__bf16>((size_t)32 * D);
    v.T = c.take<float>((size_t)s.B * R * D);
    if (s.k_dtype == PCA_F32) v.Xb = c.take<__bf16>(nelem(s, true));
    v.Op = c.take<float>((size_t)s.B * S * 16 * D);
    v.Mp = c.take<float>((size_t)s.B * S * 16);
    v.Lp = c.take<float>((size_t)s.B * S * 16);
  } else {
    v.LSE = c.take<float>((size_t)s.B * R);
    v.Gf = c.take<float>((size_t)R * s.dk);
    v.T = c.take<float>((size_t)s.B * R * s.dk);
  }
  if (out) *out = v;
  return c.off;
}
size_t mab0_d256_saved_bytes(const pca_mab_shape& s) { return fq_carve_saved(s, nullptr, nullptr); }
size_t mab0_d256_fwd_ws_bytes(const pca_mab_shape& s) {
  // two weight images (+ fp8 scales) + (inference) what the saved block would hold
  return 256 + 2 * align256((size_t)D * D * 2) + fq_carve_saved(s, nullptr, nullptr);
}

static Mab0PrepJob fq_prep_job(const pca_mab_shape& s, const float* I, const pca_mab_params& p,
                               const Fq256Saved& v) {
  Mab0PrepJob a{};
  const int m = s.nq;
  a.I = I; a.Wq = p.wq; a.bq = p.bq; a.Wk = p.wk;
  a.m = m; a.d = D; a.dq = s.dq; a.dk = s.dk; a.h = s.h; a.Rp = (int)cdiv(s.h * m, 32) * 32;
  a.sl2e = 1.4426950408889634f / sqrtf((float)D);
  a.Qp = v.Qp; a.Gf = fq_path(s) == FQ_PROJ ? nullptr : v.Gf;
  a.Gb = fq_path(s) == FQ_PMA ? v.Gb : nullptr;
  return a;
}
// The query side of every few-queries block of a training step (Qp, G: parameters only) in ONE
// launch at the start of the forward instead of one ~20 us dependent chain per block; the blocks'
// own forward calls then skip it (engine only: the saved blocks must exist, i.e. training).
static thread_local bool g_prep256_done = false;
int mab0_d256_prep_all(int n, const pca_mab_shape* const* shapes, const float* const* I,
                       const pca_mab_params* params, void* const* saved, hipStream_t st) {
  Mab0PrepJobs J{};
  for (int i = 0; i < n; ++i) {
    Fq256Saved v;
    fq_carve_saved(*shapes[i], &v, saved[i]);
    J.j[J.n++] = fq_prep_job(*shapes[i], I[i], params[i], v);
  }
  return mab0_prep_launch(J, st);
}
void mab0_d256_prep_collect(int n, const pca_mab_shape* const* shapes, const float* const* I,
                            const pca_mab_params* params, void* const* saved, Mab0PrepJobs* out) {
  for (int i = 0; i < n; ++i) {
    Fq256Saved v;
    fq_carve_saved(*shapes[i], &v, saved[i]);
    out->j[out->n++] = fq_prep_job(*shapes[i], I[i], params[i], v);
  }
}
void mab0_d256_prep_done(bool on) { g_prep256_done = on; }
bool mab0_d256_prep_pending() { return g_prep256_done; }

// operand mode of fc_o (and its adjoint) on the [B*m] query rows of a few-queries block:
// 2 (default) hi + lo bf16 pairs on the MFMA - fp32-level Z, so the ReLU mask is the exact one;
// PCA_FQ_EPI=bf16: single bf16 operands (round 2; 0.3-0.85 % of dQ elements then differ from the
// emulation through flipped pre-activations); PCA_FQ_EPI_F32=1: the exact fp32 GEMM (22 us a call)
static int fq_epi_bf16() {
  return 2;
}
int mab0_d256_fwd(const pca_mab_shape& s, const float* I, const void* X, const pca_mab_params& p,
                  float* Hout, void* saved, void* ws, hipStream_t st) {
  PCA_REQUIRE(mab0_d256_supported(s), "mab0_d256_fwd: unsupported shape");
  PCA_REQUIRE(ws != nullptr, "mab0_d256_fwd: scratch required");
  // fc_o on the [B*m] query rows with bf16 MFMA operands (fp32 accumulation and I/O), as the
  // d = 128 path's k_mid_fwd does: the exact fp32 GEMM was 22 us per call for 0.5 GFLOP
  Bf16OperandScope ops(fq_epi_bf16());
  Carver cw(ws);
  __bf16* WkP = cw.take<__bf16>((size_t)D * D);
  __bf16* WvP = cw.take<__bf16>((size_t)D * D);
  float* invs = cw.take<float>(2);
  Fq256Saved v;
  fq_carve_saved(s, &v, saved != nullptr ? saved : (void*)(cw.base + cw.off));
  const int m = s.nq;
  const int64_t Bm = (int64_t)s.B * m, M = (int64_t)s.B * s.nk;
  if (!g_prep256_done) {   // Qp = I Wq^T + bq (and, layer 1, G = sl2e Qp_h Wk_h): batch invariant
    Mab0PrepJobs J{};
    J.j[J.n++] = fq_prep_job(s, I, p, v);
    PCA_TRY(mab0_prep_launch(J, st));
  }
  if (fq_path(s) == FQ_PMA) {
    const __bf16* Xb = reinterpret_cast<const __bf16*>(X);
    if (s.k_dtype == PCA_F32) {
      PCA_TRY(cvt_f32_bf16(reinterpret_cast<const float*>(X), v.Xb, M * D, st));
      Xb = v.Xb;
    }
    const double pts = (double)M;
    {
      // reference-formulation FLOPs (fc_k, fc_v over the keys + QK^T + AV); X read once
      ProfScope ps(PCA_K_MAB0_FWD, st, 2.0 * pts * (2.0 * D * D + 2.0 * m * D), pts * 2.0 * D);
      PCA_TRY(pma_attn_fwd256(Xb, v.Gb, s.B, s.nk, s.h * m, s.k_lengths, v.Op, v.Mp, v.Lp, v.T,
                              v.LSE, st));
      ps.end();
    }
    PCA_TRY(epi_small_fwd256(v.T, v.Qp, p.wv, p.bv, s.B, m, D, v.O, st));
  } else if (fq_path(s) == FQ_PROJ) {
    const __bf16* Xb = reinterpret_cast<const __bf16*>(X);
    if (s.k_dtype == PCA_F32) {
      PCA_TRY(cvt_f32_bf16(reinterpret_cast<const float*>(X), v.Xb, M * D, st));
      Xb = v.Xb;
    }
    // PCA_FQ_FUSED_FWD=0: projection and attention as two launches (A/B measurements)
    constexpr bool fq_fused = true;
    // PCA_FQ_FUSED_F8=0: the fp8 mode on the two-launch form of round 2 (A/B measurements)
    constexpr bool fq_fused8 = true;
    const bool f8 = s.mode == PCA_MODE_FP8;
    if ((!f8 || fq_fused8) && rowstream_on() && fq_fused && m > 16) {
      // fc_k / fc_v over the keys and the attention in one pass over X (k_fq_proj_fwd)
      const float* inv_kv = invs;
      if (f8) {              // (the engine's one-launch images of the step, when there are any)
        void *wk8 = WkP, *wv8 = WvP;
        float *ik = invs, *iv = invs + 1;
        PCA_TRY(weight_image_f8(p.wk, &wk8, D, D, 0, &ik, st));
        PCA_TRY(weight_image_f8(p.wv, &wv8, D, D, 0, &iv, st));
        PCA_REQUIRE(iv == ik + 1, "mab0_d256_fwd: fp8 inverse scales of fc_k / fc_v must be adjacent");
        WkP = reinterpret_cast<decltype(WkP)>(wk8);
        WvP = reinterpret_cast<decltype(WvP)>(wv8);
        inv_kv = ik;
      } else {
        PCA_TRY(weight_image2(p.wk, &WkP, 0, p.wv, &WvP, 0, D, D, st));
      }
      const double pts = (double)M;
      ProfScope ps(PCA_K_MAB0_FWD, st, 2.0 * pts * (2.0 * D * D + 2.0 * m * D), pts * 2.0 * D);
      PCA_TRY(fq_proj_attn_fwd256(Xb, WkP, WvP, p.bk, p.bv, v.Qp, s.B, s.nk, m, s.k_lengths, v.Kp,
                                  v.Vp, v.Op, v.Mp, v.Lp, v.O, v.LSE, st,
                                  f8 ? inv_kv : nullptr));                    // modules.py:21,28-29
      ps.end();
    } else {
    if (s.mode == PCA_MODE_FP8) {           // fc_k / fc_v with fp8 e4m3 operands
      if (rowstream_on()) {
        PCA_TRY(prep_weight_f8(p.wk, WkP, D, D, 0, invs, st));
        PCA_TRY(prep_weight_f8(p.wv, WvP, D, D, 0, invs + 1, st));
        PCA_TRY(rowstream256_proj2_f8(Xb, WkP, WvP, invs, p.bk, p.bv, v.Kp, v.Vp, s.B, s.nk, st));
      } else {
        PCA_TRY(prep_weight_f8(p.wk, WkP, D, D, 1, invs, st));
        PCA_TRY(prep_weight_f8(p.wv, WvP, D, D, 1, invs + 1, st));
        PCA_TRY(rowgemm256_proj_f8(Xb, WkP, invs, p.bk, v.Kp, s.B, s.nk, st));
        PCA_TRY(rowgemm256_proj_f8(Xb, WvP, invs + 1, p.bv, v.Vp, s.B, s.nk, st));
      }
    } else {
      if (rowstream_on()) {                  // one pass over X, weights in registers
        PCA_TRY(weight_image2(p.wk, &WkP, 0, p.wv, &WvP, 0, D, D, st));
        PCA_TRY(rowstream256_proj2(Xb, WkP, WvP, p.bk, p.bv, v.Kp, v.Vp, s.B, s.nk, st));   // modules.py:21
      } else {
        PCA_TRY(prep_weight(p.wk, WkP, D, D, 1, st));
        PCA_TRY(prep_weight(p.wv, WvP, D, D, 1, st));
        PCA_TRY(rowgemm256_proj(Xb, WkP, p.bk, v.Kp, s.B, s.nk, st));
        PCA_TRY(rowgemm256_proj(Xb, WvP, p.bv, v.Vp, s.B, s.nk, st));
      }
    }
    const double pts = (double)M;
    ProfScope ps(PCA_K_MAB0_FWD, st, 2.0 * pts * 2.0 * m * D, pts * 4.0 * D);
    PCA_TRY(fq_attn_fwd256(v.Kp, v.Vp, v.Qp, s.B, s.nk, m, s.k_lengths, v.Op, v.Mp, v.Lp, v.O,
                           v.LSE, st));                                      // :28-29
    ps.end();
    }
  } else {
    PCA_TRY(mab0_attn_small_launch(reinterpret_cast<const float*>(X), v.Gf, s.B, s.nk, s.h * m,
                                   s.dk, v.T, v.LSE, s.k_lengths, st));
    PCA_TRY(epi_small_fwd256(v.T, v.Qp, p.wv, p.bv, s.B, m, s.dk, v.O, st));
  }
  if (g_mid256.armed && mid256_on() && m == 32 && g_mid256.s1.nk == 32 && g_mid256.s1.d == D &&
      g_mid256.s1.B == s.B && fq_epi_bf16() == 2) {
    // epilogue + the next block's K / V projections and images in one launch per set (mid256.hip)
    Mab1Saved v1;
    mab1_carve_saved(g_mid256.s1, &v1, g_mid256.saved1);
    const pca_mab_params& p1 = g_mid256.p1;
    PCA_TRY(mid256_fwd(v.O, p.wo, p.bo, p1.wk, p1.bk, p1.wv, p1.bv, v.Z, Hout, v1.KpP, v1.VpP, v1.Kt,
                       v1.Vt, s.B, st));
    g_mid256.ready = true;
    return PCA_OK;
  }
  PCA_TRY(linear_fwd_f32(v.O, p.wo, p.bo, v.Z, Bm, D, D, st));              // :31
  return add_relu(v.O, v.Z, Hout, Bm * D, st);
}

struct Fq256BwdWs {
  __bf16 *WkTP, *WvTP, *dKp, *dVp, *dXb, *dTb, *TG;
  float* LSEp;
  float *dZ, *dO, *dOt, *Delta, *dQpPart, *dTf, *DG, *dQp;
  float* slabs;      // per-workgroup partial sums of the reductions that used fp32 atomics
  void* wg;
};
static size_t fq_carve_bwd(const pca_mab_shape& s, Fq256BwdWs* out, void* base) {
  Carver c(base);
  Fq256BwdWs w{};
  const int m = s.nq, R = s.h * m;
  const size_t Bm = (size_t)s.B * m;
  w.dZ = c.take<float>(Bm * D);
  w.dO = c.take<float>(Bm * D);
  w.dOt = c.take<float>(Bm * D);
  w.dQp = c.take<float>((size_t)m * D);
  if (fq_path(s) != FQ_PROJ) w.wg = c.take<char>(wgrad256_ws_bytes(1, (int64_t)Bm));
  if (fq_path(s) == FQ_PMA) {
    if (s.k_dtype == PCA_F32) w.dXb = c.take<__bf16>(nelem(s, true));
    w.Delta = c.take<float>((size_t)s.B * 16);
    w.LSEp = c.take<float>((size_t)s.B * 16);
    w.dTb = c.take<__bf16>((size_t)s.B * 16 * D);
    w.TG = c.take<__bf16>((size_t)s.B * D * 32);
    w.DG = c.take<float>((size_t)16 * D);
    w.slabs = reinterpret_cast<float*>(c.take<char>(pma_bwd256_slab_bytes(s.B)));
  } else if (s.dk == D) {
    const int MQ = m > 16 ? 32 : 16, S = fq_splits256(s.B, s.nk);
    w.WkTP = c.take<__bf16>((size_t)D * D);
    w.WvTP = c.take<__bf16>((size_t)D * D);
    w.dKp = c.take<__bf16>(nelem(s, true));
    w.dVp = c.take<__bf16>(nelem(s, true));
    if (s.k_dtype == PCA_F32) w.dXb = c.take<__bf16>(nelem(s, true));
    w.Delta = c.take<float>((size_t)s.B * H8 * MQ);
    w.dQpPart = c.take<float>((size_t)s.B * S * m * D);
    {   // three [B nk]-row jobs (+ mab1's handed-over one is among them) and the [B m]-row fc_o job
      const int64_t Mk = (int64_t)s.B * s.nk;
      w.wg = c.take<char>(wgrad256_ws_bytes(3, Mk > (int64_t)Bm ? Mk : (int64_t)Bm));
    }
    w.slabs = c.take<float>((size_t)cdiv((int64_t)Bm, 512) * D);          // column-sum partials
  } else {
    w.Delta = c.take<float>((size_t)s.B * R);
    w.dTf = c.take<float>((size_t)s.B * R * s.dk);
    w.DG = c.take<float>((size_t)R * s.dk);
    const size_t a = epi_small_bwd256_ws_bytes(s.B, m), b = (size_t)s.B * R * s.dk * sizeof(float);
    w.slabs = reinterpret_cast<float*>(c.take<char>(a > b ? a : b));     // (used one after the other)
  }
  if (out) *out = w;
  return c.off;
}
size_t mab0_d256_bwd_ws_bytes(const pca_mab_shape& s) { return fq_carve_bwd(s, nullptr, nullptr); }

// dQ -> dI [m, dq] (ACCUMULATED, may be null), dK -> dX [B, N, dk] (written or accumulated)
int mab0_d256_bwd(const pca_mab_shape& s, const float* I, const void* X, const pca_mab_params& p,
                  const void* saved, const float* dH, float* dI, void* dX, int dk_accumulate,
                  const pca_mab_grads& gr, void* ws, hipStream_t st, BwdDefer* defer) {
  Fq256Saved v;
  fq_carve_saved(s, &v, const_cast<void*>(saved));
  Fq256BwdWs w;
  fq_carve_bwd(s, &w, ws);
  Bf16OperandScope ops(fq_epi_bf16());       // dO = dH + dZ Wo on the MFMA (see mab0_d256_fwd)
  const int m = s.nq, R = s.h * m;
  const int64_t Bm = (int64_t)s.B * m, M = (int64_t)s.B * s.nk;
  if (s.dk != D && dX != nullptr) {
    set_error("mab0_bf16_bwd: dK for dk <= 4 is not built (the set is the model input)");
    return PCA_EUNSUPPORTED;
  }
  // ---- epilogue adjoint: H = O + relu(O Wo^T + bo) ----
  PCA_TRY(relu_bwd_copy(dH, v.Z, w.dZ, w.dO, Bm * D, st));       // dZ, and dO = dH
  {
    Wgrad256Jobs ej{};
    ej.j[ej.n++] = Wgrad256Job{w.dZ, v.O, gr.wo, gr.bo, Bm};
    PCA_TRY(wgrad256_short(ej, w.wg, defer, st));
  }
  PCA_TRY(linear_dx_acc_f32(w.dZ, p.wo, w.dO, Bm, D, D, 1, st));
  const float sl2e = 1.4426950408889634f / sqrtf((float)D);
  Mab0PostJob pj{};
  pj.Qp = v.Qp; pj.Wk = p.wk; pj.I = I; pj.Wq = p.wq;
  pj.dWk = gr.wk; pj.dQp = w.dQp; pj.dWq = gr.wq; pj.dbq = gr.bq; pj.dI = dI;
  pj.m = m; pj.d = D; pj.dk = s.dk; pj.dq = s.dq; pj.h = s.h; pj.sl2e = sl2e; pj.B = s.B;
  if (fq_path(s) == FQ_PMA) {
    const __bf16* Xb = s.k_dtype == PCA_F32 ? v.Xb : reinterpret_cast<const __bf16*>(X);
    PCA_TRY(pma_epi_bwd256(w.dO, v.T, v.LSE, p.wv, v.Gf, s.B, m, R, w.dTb, w.TG, w.Delta, w.LSEp,
                           gr.wv, st));
    PCA_TRY(colsum(w.dO, Bm, D, gr.bv, 1, st));
    const bool f32 = s.k_dtype == PCA_F32;
    __bf16* dXb = dX == nullptr ? nullptr : (f32 ? w.dXb : reinterpret_cast<__bf16*>(dX));
    const double pts = (double)M;
    {
      ProfScope ps(PCA_K_MAB0_BWD, st, 4.0 * pts * (2.0 * D * D + 2.0 * m * D),
                   pts * 2.0 * D * (1.0 + (dX != nullptr ? (dk_accumulate ? 2.0 : 1.0) : 0.0)));
      PCA_TRY(pma_attn_bwd256(Xb, v.Gb, w.dTb, w.TG, w.LSEp, w.Delta, s.B, s.nk, R, s.k_lengths,
                              dXb, (!f32 && dk_accumulate) ? 1 : 0, w.DG, w.slabs, st));
      ps.end();
    }
    if (dX != nullptr && f32)
      PCA_TRY(cvt_bf16_f32(dXb, reinterpret_cast<float*>(dX), M * D, dk_accumulate ? 1 : 0, st));
    pj.DG = w.DG;
    pj.dO = w.dO;
  } else if (s.dk == D) {
    const __bf16* Xb = s.k_dtype == PCA_F32 ? v.Xb : reinterpret_cast<const __bf16*>(X);
    const double pts = (double)M;
    {
      ProfScope ps(PCA_K_MAB0_BWD, st, 4.0 * pts * 2.0 * m * D, pts * 8.0 * D);
      PCA_TRY(fq_attn_bwd256(v.Kp, v.Vp, v.Qp, w.dO, v.O, v.LSE, w.Delta, s.B, s.nk, m,
                             s.k_lengths, w.dKp, w.dVp, w.dQpPart, w.dOt, st));
      ps.end();
    }
    // fc_v bias: every row of A sums to one, so colsum(dVp) = sum over sets and queries of dO;
    // fc_k bias: identically zero (softmax shift invariance), left untouched
    if (Bm > 256) {                         // (per-block partials + ordered sum, no atomics; <= 256
                                            //  rows: colsum's one-launch form has none either)
      int np = 0;
      PCA_TRY(colsum_parts(w.dO, Bm, D, w.slabs, &np, st));
      PCA_TRY(slab_sum(w.slabs, np, D, gr.bv, 1, st));
    } else {
      PCA_TRY(colsum(w.dO, Bm, D, gr.bv, 1, st));
    }
    if (dX != nullptr) {
      const bool f32 = s.k_dtype == PCA_F32;
      __bf16* dXb = f32 ? w.dXb : reinterpret_cast<__bf16*>(dX);
      if (rowstream_on()) {                  // dKp Wk + dVp Wv in one pass, weights in registers
        PCA_TRY(weight_image2(p.wk, &w.WkTP, 3, p.wv, &w.WvTP, 3, D, D, st));
        if (g_handoff.has_dx && !f32 && dk_accumulate && g_handoff.dx.dX == dXb &&
            g_handoff.dx.B == s.B && g_handoff.dx.N == s.nk) {
          // ... and mab1's dQp Wq (it would have written dX first; nothing to accumulate onto)
          g_handoff.has_dx = false;
          PCA_TRY(rowstream256_dx3(g_handoff.dx.dQp, w.dKp, w.dVp, g_handoff.dx.WqT, w.WkTP, w.WvTP,
                                   dXb, s.B, s.nk, st));
        } else {
          // a pending hand-over that cannot be fused here: mab1's dX has to exist before anything
          // is accumulated onto it (it used to be written by the later flush,
          // over this block's sum)
          if (g_handoff.has_dx) PCA_TRY(wgrad256_handoff_flush_dx(st));
          PCA_TRY(rowstream256_dx2(w.dKp, w.dVp, w.WkTP, w.WvTP, dXb, s.B, s.nk,
                                   (!f32 && dk_accumulate) ? 1 : 0, st));
        }
      } else {
        if (g_handoff.has_dx) PCA_TRY(wgrad256_handoff_flush_dx(st));
        PCA_TRY(prep_weight(p.wk, w.WkTP, D, D, 2, st));
        PCA_TRY(prep_weight(p.wv, w.WvTP, D, D, 2, st));
        PCA_TRY(rowgemm256_dx(w.dKp, w.WkTP, dXb, s.B, s.nk, (!f32 && dk_accumulate) ? 1 : 0, st));
        PCA_TRY(rowgemm256_dx(w.dVp, w.WvTP, dXb, s.B, s.nk, 1, st));
      }
      if (f32)
        PCA_TRY(cvt_bf16_f32(dXb, reinterpret_cast<float*>(dX), M * D, dk_accumulate ? 1 : 0, st));
    }
    Wgrad256Jobs jobs{};
    jobs.j[jobs.n++] = Wgrad256Job{w.dKp, Xb, gr.wk, nullptr, M};
    jobs.j[jobs.n++] = Wgrad256Job{w.dVp, Xb, gr.wv, nullptr, M};
    if (g_handoff.has && g_handoff.job.A == Xb && g_handoff.job.M == M) {     // mab1's {dQp, X}
      jobs.j[jobs.n++] = g_handoff.job;
      g_handoff.has = false;
    }
    PCA_TRY(wgrad256_launch(jobs, w.wg, st));
    pj.DG = nullptr;
    pj.dO = w.dOt;
  } else {
    // (both reductions through per-workgroup / per-set partials in w.slabs and a fixed-order sum:
    //  no fp32 atomics anywhere in the d = 256 step, so it is bit-reproducible run to run)
    PCA_TRY(epi_small_bwd256(w.dO, v.T, p.wv, s.B, m, s.dk, w.dTf, w.Delta, gr.wv, gr.bv, w.slabs,
                             st));
    if ((R * s.dk) % 4 == 0) {
      PCA_TRY(mab0_bwd_small_launch(reinterpret_cast<const float*>(X), v.Gf, w.dTf, v.LSE, w.Delta,
                                    s.B, s.nk, R, R, s.dk, w.DG, s.k_lengths, st, w.slabs));
      PCA_TRY(slab_sum(w.slabs, s.B, R * s.dk, w.DG, 0, st));
    } else {
      PCA_TRY(fill_zero(w.DG, (int64_t)R * s.dk, st));
      PCA_TRY(mab0_bwd_small_launch(reinterpret_cast<const float*>(X), v.Gf, w.dTf, v.LSE, w.Delta,
                                    s.B, s.nk, R, R, s.dk, w.DG, s.k_lengths, st));
    }
    pj.DG = w.DG;
    pj.dO = w.dO;
  }
  if (defer != nullptr && defer->posts.n < 3) {
    defer->posts.j[defer->posts.n++] = pj;
    return PCA_OK;
  }
  Mab0PostJobs one{};
  one.j[one.n++] = pj;
  return mab0_post_launch(one, st);
}

}  // namespace pca
