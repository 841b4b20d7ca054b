// ISAB-level orchestration of the fused bf16 path (set_transformer-master/modules.py:51-53):
//   forward   mab0 attention partials -> k_mid_fwd (epilogue of mab0 + K/V of mab1) -> mab1
//   backward  mab1 chain (dX, dKp, dVp) -> batched wgrad -> k_mid_bwd -> mab0 backward
// Only host code here; kernels live in mab0_*, mab1_*, mid_bf16.hip.
#include "mab1_bf16.hpp"

namespace pca {


// s0 = mab0(I, X) shape, s1 = mab1(X, H) shape of the same ISAB
bool isab_bf16_supported(const pca_mab_shape& s0, const pca_mab_shape& s1) {
  return mab_kind(s0) == 2 && mab_kind(s1) == 1 && s0.nq == 16 && s1.nk == 16 && s0.h == 4 &&
         s0.d == 128 && s0.B == s1.B && s0.nk == s1.nq;
}

namespace {
constexpr size_t DD = 128 * 128;
}  // namespace

// images of one ISAB: 13 bf16 [128][128] blocks
size_t isab_img_bytes() { return 13 * align256(DD * 2); }
void isab_img_carve(void* base, IsabImg* im) {
  Carver c(base);
  __bf16** f[13] = {&im->Wv0, &im->Wo0, &im->Wk1, &im->Wv1, &im->WqB, &im->WoP, &im->Wk1T,
                    &im->Wv1T, &im->Wo0TP, &im->Wv0TP, &im->Wv0T, &im->WoTP, &im->WqTP};
  for (auto* q : f) *q = c.take<__bf16>(DD);
}
// append the preparation jobs of one ISAB (dk = input width of the layer: 128 or <= 4)
void isab_collect_prep(const pca_mab_shape& s0, const pca_mab_params& p0,
                       const pca_mab_params& p1, const IsabImg& im, bool training,
                       bool need_dx, PrepJobs* J) {
  const int d = 128, dk = s0.dk;
  auto add = [&](const float* src, __bf16* dst, int rows, int cols, int mode) {
    J->j[J->n++] = PrepJob{src, dst, rows, cols, mode};
  };
  if (dk > 4) add(p0.wv, im.Wv0, d, dk, 0);
  add(p0.wo, im.Wo0, d, d, 0);
  add(p1.wk, im.Wk1, d, d, 0);
  add(p1.wv, im.Wv1, d, d, 0);
  if (dk > 4) add(p1.wq, im.WqB, d, d, 0);
  add(p1.wo, im.WoP, d, d, 1);
  if (!training) return;
  add(p1.wk, im.Wk1T, d, d, 3);
  add(p1.wv, im.Wv1T, d, d, 3);
  add(p0.wo, im.Wo0TP, d, d, 2);
  if (dk > 4) {
    add(p0.wv, im.Wv0TP, d, dk, 2);
    add(p0.wv, im.Wv0T, d, dk, 3);
  }
  add(p1.wo, im.WoTP, d, d, 2);
  if (need_dx) add(p1.wq, im.WqTP, d, d, 2);
}

size_t isab_bf16_fwd_ws_bytes(const pca_mab_shape& s0, const pca_mab_shape& s1) {
  (void)s0;
  return mab1_bf16_fwd_ws_bytes(s1);
}
size_t isab_bf16_bwd_ws_bytes(const pca_mab_shape& s0, const pca_mab_shape& s1) {
  return mab1_carve_bwd_ws(s1, nullptr, nullptr) + mab0_carve_bwd_ws(s0, nullptr, nullptr);
}

int isab_bf16_fwd(const pca_mab_shape& s0, const pca_mab_shape& s1, const float* I,
                  const void* X, const pca_mab_params& p0, const pca_mab_params& p1, float* H,
                  void* Y, void* saved0, void* saved1, void* ws, const IsabImg& im,
                  hipStream_t st) {
  PCA_REQUIRE(saved0 && saved1 && ws, "isab_bf16_fwd: null block");
  const int dk = s0.dk;
  Mab0Saved v0;
  mab0_carve_saved(s0, &v0, saved0);
  Mab1Saved v1;
  mab1_carve_saved(s1, &v1, saved1);

  PCA_TRY(mab0_bf16_fwd_ex(s0, I, X, p0, H, saved0, nullptr,
                           PCA_F_SKIP_EPILOGUE | PCA_F_PREP_DONE, st));
  MidFwdLaunch L{};
  L.B = s0.B; L.dk = dk; L.S = dk > 4 ? mab0_splits(s0) : 0;
  L.Tp = v0.Tp; L.Mp = v0.Mp; L.Lp = v0.Lp; L.T = v0.T; L.LSE = v0.LSE; L.Qp = v0.Qp;
  L.Wv0 = im.Wv0; L.Wv0f = p0.wv; L.bv0 = p0.bv; L.bo0 = p0.bo; L.Wo0 = im.Wo0;
  L.Wk1 = im.Wk1; L.Wv1 = im.Wv1; L.bk1 = p1.bk; L.bv1 = p1.bv;
  L.O = v0.O; L.Z = v0.Z; L.H = H;
  L.KpP = v1.KpP; L.VpP = v1.VpP; L.Kt = v1.Kt; L.Vt = v1.Vt;
  PCA_TRY(mid_fwd_launch(L, st));
  return mab1_bf16_fwd_ex(s1, X, H, p1, Y, saved1, ws, PCA_F_KV_READY, st, &im);
}

// dX: gradient w.r.t. the ISAB input (null for the first layer, whose input is the data);
// written here (mab1's dQ) and then accumulated into (mab0's dK).  dI accumulated.
int isab_bf16_bwd(const pca_mab_shape& s0, const pca_mab_shape& s1, const float* I,
                  const void* X, const float* H, const pca_mab_params& p0,
                  const pca_mab_params& p1, const void* saved0, const void* saved1,
                  const void* dY, float* dI, void* dX, const pca_mab_grads& g0,
                  const pca_mab_grads& g1, void* ws, const IsabImg& im, hipStream_t st,
                  BwdDefer* defer) {
  const int dk = s0.dk, m = 16;
  Carver c(ws);
  void* ws1 = c.take<char>(mab1_carve_bwd_ws(s1, nullptr, nullptr));
  void* ws0 = c.take<char>(mab0_carve_bwd_ws(s0, nullptr, nullptr));
  Mab1BwdWs w1;
  mab1_carve_bwd_ws(s1, &w1, ws1);
  Mab0BwdWs w0;
  mab0_carve_bwd_ws(s0, &w0, ws0);
  Mab0Saved v0;
  mab0_carve_saved(s0, &v0, const_cast<void*>(saved0));

  int nparts = 0;
  PCA_TRY(mab1_bf16_bwd_ex(s1, X, H, p1, saved1, dY, dX, nullptr, 0, g1, ws1,
                           PCA_F_SKIP_KV_TAIL, st, &im, nullptr, 0, &nparts, defer));
  const int64_t Bm = (int64_t)s0.B * m;
  const int Rp = 64;
  MidBwdLaunch L{};
  L.B = s0.B; L.dk = dk;
  L.dKpPart = w1.dKpPart; L.dVpPart = w1.dVpPart; L.nparts = nparts;
  L.dKp = w1.dKp; L.dVp = w1.dVp;
  L.zero_ptr = w0.DG; L.zero_n = Rp * dk;         // cleared for k_mab0_bwd's atomics
  L.Z = v0.Z; L.T = v0.T; L.LSE = v0.LSE;
  L.Wk1T = im.Wk1T; L.Wv1T = im.Wv1T; L.Wo0TP = im.Wo0TP; L.Wv0TP = im.Wv0TP; L.Wv0T = im.Wv0T;
  L.Wv0f = p0.wv;
  L.dZ = w0.dZ; L.dO = w0.dO; L.Th = w0.Th; L.dQs = w0.dQs; L.dTf = w0.dTf; L.dTb = w0.dTb;
  L.dTt = w0.dTt; L.Delta = w0.Delta; L.LSEp = w0.LSEp;
  PCA_TRY(mid_bwd_launch(L, st));
  {   // every [B*m]-row weight gradient of the ISAB in ONE launch
    WgradJobs jobs{};
    jobs.j[jobs.n++] = WgradJob{w1.dKp, H, g1.wk, g1.bk, Bm, 0, 128};
    jobs.j[jobs.n++] = WgradJob{w1.dVp, H, g1.wv, g1.bv, Bm, 0, 128};
    jobs.j[jobs.n++] = WgradJob{w0.dZ, v0.O, g0.wo, g0.bo, Bm, 0, 128};
    if (dk > 4)
      for (int j = 0; j < 4; ++j)
        jobs.j[jobs.n++] = WgradJob{w0.dO, w0.Th + (int64_t)j * Bm * dk, g0.wv,
                                    j == 0 ? g0.bv : nullptr, Bm, 32 * j, 32 * (j + 1)};
    hipStream_t ts = terminal_stream(st);
    PCA_TRY(wgrad128_defer(defer, jobs, false, 64, ts));
    if (dk <= 4)
      PCA_TRY(wgrad_small_f32_launch(w0.dO, w0.Th, Bm, dk, (int64_t)Bm * dk, g0.wv, g0.bv, ts,
                                     defer));
  }
  return mab0_bf16_bwd_ex(s0, I, X, p0, saved0, nullptr, dI, dX, dX != nullptr ? 1 : 0, g0, ws0,
                          PCA_F_SKIP_HEAD | PCA_F_SKIP_WGRAD, st, defer);
}

}  // namespace pca
