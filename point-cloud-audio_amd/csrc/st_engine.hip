// Whole-model engine for the ST classifier (Code/models.py:13-44) and its train step
// (Code/settransformer.py:100-108): one host call enqueues every kernel of
// forward -> cross-entropy -> backward against flat parameter / gradient vectors laid out
// in state_dict order.  No Python, autograd or allocator sits between the kernels, so the
// caller can capture the call in a hipGraph.
#include <stdlib.h>
#include "mab1_bf16.hpp"
#include "pack_body.hpp"
#include "d256_bf16.hpp"
#include "set128.hpp"

namespace pca {

// per-block dispatch (api_mab.hip): fused bf16 kernel where one exists for the block's shape
// and the config asks for PCA_MODE_BF16, exact fp32 GEMM chain otherwise

namespace {

struct MabOff {   // element offsets of one MAB's 8 tensors in the flat vector
  int64_t wq, bq, wk, bk, wv, bv, wo, bo;
};

struct Layout {
  int64_t I[2];
  MabOff mab0[2], mab1[2];
  int64_t S;
  MabOff pma;
  int64_t wc, bc;
  int64_t total;
  int64_t enc1_begin;
};

inline MabOff lay_mab(int64_t& o, int dq, int dk, int d) {
  MabOff m;
  m.wq = o; o += (int64_t)d * dq;
  m.bq = o; o += d;
  m.wk = o; o += (int64_t)d * dk;
  m.bk = o; o += d;
  m.wv = o; o += (int64_t)d * dk;
  m.bv = o; o += d;
  m.wo = o; o += (int64_t)d * d;
  m.bo = o; o += d;
  return m;
}

// state_dict order of Code/models.py:34-41 (ISAB: I, mab0, mab1 -- modules.py:45-49)
inline Layout layout(const pca_st_config& c) {
  Layout L;
  int64_t o = 0;
  for (int li = 0; li < 2; ++li) {
    const int din = li == 0 ? c.din : c.d;
    if (li == 1) L.enc1_begin = o;
    L.I[li] = o; o += (int64_t)c.m * c.d;
    L.mab0[li] = lay_mab(o, c.d, din, c.d);   // MAB(dim_out, dim_in, dim_out)
    L.mab1[li] = lay_mab(o, din, c.d, c.d);   // MAB(dim_in, dim_out, dim_out)
  }
  L.S = o; o += (int64_t)c.k * c.d;
  L.pma = lay_mab(o, c.d, c.d, c.d);
  L.wc = o; o += (int64_t)c.C * c.d;
  L.bc = o; o += c.C;
  L.total = o;
  return L;
}

inline pca_mab_params params_at(const float* base, const MabOff& m) {
  return pca_mab_params{base + m.wq, base + m.bq, base + m.wk, base + m.bk,
                        base + m.wv, base + m.bv, base + m.wo, base + m.bo,
                        nullptr, nullptr, nullptr, nullptr};
}
inline pca_mab_grads grads_at(float* base, const MabOff& m) {
  return pca_mab_grads{base + m.wq, base + m.bq, base + m.wk, base + m.bk,
                       base + m.wv, base + m.bv, base + m.wo, base + m.bo,
                       nullptr, nullptr, nullptr, nullptr};
}

inline pca_mab_shape shape(const pca_st_config& c, int nq, int nk, int dq, int dk,
                           int q_shared) {
  pca_mab_shape s{};
  s.B = c.B; s.nq = nq; s.nk = nk; s.dq = dq; s.dk = dk; s.d = c.d; s.h = c.h;
  s.q_shared = q_shared; s.mode = c.mode;
  s.q_dtype = s.k_dtype = s.y_dtype = PCA_F32;
  return s;
}

struct Shapes {
  pca_mab_shape m0[2], m1[2], pma;
  bool act_bf16;      // hidden activations Y1, Y2 (and their gradients) travel in bf16
};
inline Shapes shapes(const pca_st_config& c, bool training, const int32_t* lengths = nullptr) {
  Shapes s;
  for (int li = 0; li < 2; ++li) {
    const int din = li == 0 ? c.din : c.d;
    s.m0[li] = shape(c, c.m, c.N, c.d, din, 1);
    s.m1[li] = shape(c, c.N, c.m, din, c.d, 0);
  }
  s.pma = shape(c, c.k, c.N, c.d, c.d, 1);
  s.act_bf16 = false;
  if (training && c.mode != PCA_MODE_F32) {
    // bf16 activations only when EVERY block runs on a fused kernel that understands them
    Shapes t = s;
    t.m1[0].y_dtype = PCA_BF16;
    t.m0[1].k_dtype = PCA_BF16;
    t.m1[1].q_dtype = PCA_BF16;
    t.m1[1].y_dtype = PCA_BF16;
    t.pma.k_dtype = PCA_BF16;
    const bool isab128 = isab_bf16_supported(t.m0[0], t.m1[0]) &&
                         isab_bf16_supported(t.m0[1], t.m1[1]);
    // d = 256: every block has a fused kernel of its own (no ISAB-level fusion)
    const bool blocks256 = c.d == 256 && mab_kind(t.m0[0]) == 2 && mab_kind(t.m1[0]) == 1 &&
                           mab_kind(t.m0[1]) == 2 && mab_kind(t.m1[1]) == 1;
    if ((isab128 || blocks256) && mab_kind(t.pma) == 2) {
      s = t;
      s.act_bf16 = true;
    }
  }
  // variable-size sets: the points are the KEYS of the three blocks that attend over them
  s.m0[0].k_lengths = s.m0[1].k_lengths = s.pma.k_lengths = lengths;
  return s;
}

// the PMA epilogue + classifier + loss launch (k_pma_head) exists for d = 128
inline bool pma_head_ok(const Shapes& s) { return mab_kind(s.pma) == 2 && s.pma.d == 128; }

// the set-resident forward (set128_fwd.hip) takes a training step whose blocks are all on the fused
// d = 128 kernels, whose sets fit one workgroup's LDS and are dense (PCA_SET128=0: the per-block launches)
inline bool set128_on(const pca_st_config& c, const Shapes& s) {
  const char* e = getenv("PCA_SET128");          // (read per call: tests compare the two forms in-process)
  const bool off = e != nullptr && e[0] == '0';
  return !off && c.mode == PCA_MODE_BF16 && s.act_bf16 &&
         set128_shape_ok(c.B, c.N, c.din, c.d, c.h, c.m, c.k) &&
         isab_bf16_supported(s.m0[0], s.m1[0]) && isab_bf16_supported(s.m0[1], s.m1[1]) &&
         pma_head_ok(s) && s.pma.k_lengths == nullptr;
}

struct Ws {
  void* saved[5];            // mab0[0], mab1[0], mab0[1], mab1[1], pma
  float *H[2], *Y[2], *P, *logits, *dlogits;
  float *dP, *dY2, *dY1, *dH, *clsws;
  void* scratch;             // forward + PMA backward
  void* scratch_bw[2];       // backward of enc.0 / enc.1: separate, because their terminal
                             // reductions run on the helper stream while the main stream moves on
  void* scratch_pma;         // d = 256: the PMA's backward workspace, kept until the deferred post
                             // stages of all three few-queries blocks have run (else = scratch)
  IsabImg img[2];            // weight images of the two ISABs (fused bf16 path)
  bool fused[2];
  float* wg_slabs;           // weight-gradient partials of the deferred reductions (fused d = 128)
  size_t wg_slab_bytes;
  __bf16* img256;            // d = 256: 24 weight images [256][256] prepared in one launch per step
  uint8_t* img256f8;         // d = 256, fp8 mode: e4m3 weight images of the step (one launch) ...
  float* inv256f8;           // ... and their inverse scales
  void* wg256_def;           // d = 256: slabs of the deferred [B*m]-row weight-gradient launch
  void* scratch_m0;          // d = 256: enc.0's few-queries backward (its ISAB partner's [B*m]-row
                             // operands in scratch_bw[0] stay in place until that launch)
  void* set128_ws;           // set-resident forward: pair flags + hand-off slots (its fused head writes the
                             // PMA's backward operands into `scratch` while other pairs still exchange)
};

inline size_t carve(const pca_st_config& c, int training, Ws* out, void* base) {
  const Shapes s = shapes(c, training != 0);
  Carver cv(base);
  Ws w{};
  const pca_mab_shape* order[5] = {&s.m0[0], &s.m1[0], &s.m0[1], &s.m1[1], &s.pma};
  size_t max_scratch = 0;
  for (int i = 0; i < 5; ++i) {
    const size_t fb = mab_fwd_ws_bytes_any(*order[i]);
    max_scratch = fb > max_scratch ? fb : max_scratch;
    if (training) {
      const size_t bb = mab_bwd_ws_bytes_any(*order[i]);
      max_scratch = bb > max_scratch ? bb : max_scratch;
      w.saved[i] = cv.take<char>(mab_saved_bytes_any(*order[i]));
    }
  }
  for (int li = 0; li < 2; ++li) {
    w.fused[li] = training && isab_bf16_supported(s.m0[li], s.m1[li]);
    if (w.fused[li]) {
      const size_t fb = isab_bf16_fwd_ws_bytes(s.m0[li], s.m1[li]);
      const size_t bb = isab_bf16_bwd_ws_bytes(s.m0[li], s.m1[li]);
      max_scratch = fb > max_scratch ? fb : max_scratch;
      max_scratch = bb > max_scratch ? bb : max_scratch;
      char* ib = cv.take<char>(isab_img_bytes());
      if (base != nullptr) isab_img_carve(ib, &w.img[li]);
    }
  }
  if (training && set128_shape_ok(c.B, c.N, c.din, c.d, c.h, c.m, c.k))
    w.set128_ws = cv.take<char>(set128_fwd_ws_bytes(c.B));
  const size_t BN = (size_t)c.B * c.N, Bm = (size_t)c.B * c.m;
  w.H[0] = cv.take<float>(Bm * c.d);
  w.H[1] = cv.take<float>(Bm * c.d);
  w.Y[0] = cv.take<float>(BN * c.d);
  w.Y[1] = cv.take<float>(BN * c.d);
  w.P = cv.take<float>((size_t)c.B * c.k * c.d);
  w.logits = cv.take<float>((size_t)c.B * c.k * c.C);
  if (training) {
    w.dlogits = cv.take<float>((size_t)c.B * c.k * c.C);
    w.dP = cv.take<float>((size_t)c.B * c.k * c.d);
    w.dY2 = cv.take<float>(BN * c.d);
    w.dY1 = cv.take<float>(BN * c.d);
    w.dH = cv.take<float>(Bm * c.d);
    w.clsws = cv.take<float>(2 * (size_t)c.B);
  }
  w.scratch = cv.take<char>(max_scratch);
  for (int li = 0; li < 2; ++li)
    w.scratch_bw[li] = training ? (void*)cv.take<char>(max_scratch) : w.scratch;
  w.scratch_pma = w.scratch;
  if (training && s.pma.d == 256 && mab_kind(s.pma) == 2)
    w.scratch_pma = cv.take<char>(mab_bwd_ws_bytes_any(s.pma));
  if ((w.fused[0] || w.fused[1]) && wgrad_slabs_on()) {
    // two lists (B*N-row and B*m-row jobs) of up to ~600 [128 x 128 (+128)] fp32 slabs each;
    // bwd_defer_flush gives a workgroup more rows when a list would not fit
    constexpr int slab_mb = 40;
    w.wg_slab_bytes = 2 * (size_t)slab_mb * 1024 * 1024;
    w.wg_slabs = cv.take<float>(w.wg_slab_bytes / sizeof(float));
  }
  if (training && c.d == 256 && (c.mode == PCA_MODE_BF16 || c.mode == PCA_MODE_FP8))
    w.img256 = cv.take<__bf16>((size_t)24 * 256 * 256);
  if (training && c.d == 256 && c.mode == PCA_MODE_FP8) {
    w.img256f8 = cv.take<uint8_t>((size_t)8 * 256 * 256);
    w.inv256f8 = cv.take<float>(16);
  }
  w.scratch_m0 = w.scratch_bw[0];
  if (training && c.d == 256 && mab_kind(s.m0[0]) == 2 && mab_kind(s.m1[0]) == 1) {
    w.wg256_def = cv.take<char>(wgrad256_ws_bytes(8, (int64_t)c.B * c.m));
    w.scratch_m0 = cv.take<char>(mab_bwd_ws_bytes_any(s.m0[0]));
  }
  if (out) *out = w;
  return cv.off;
}

// Every bf16 weight image the d = 256 blocks of a training step ask for (weight_image1 / 2 in
// mab1_bf16.hip, d256_host.hip), registered in `tab`; launch != 0 also converts them, all in one
// launch.  A request this list does not foresee is converted on the spot by the block itself.
// (jobs_out != nullptr: the bf16 image jobs are handed to the caller, who launches them together with the
//  step's other preparation work - forward() -, instead of being launched here)
inline int images256_prepare(const pca_st_config& c, const Layout& L, const Shapes& s, const float* p,
                             const Ws& w, WeightImages* tab, bool launch, hipStream_t st,
                             PrepJobs* jobs_out = nullptr) {
  tab->n = 0;
  if (w.img256 == nullptr) return PCA_OK;
  PrepJobs J{};
  auto add = [&](const float* src, int mode) {
    if (tab->n >= 24) return;
    __bf16* img = w.img256 + (size_t)tab->n * 256 * 256;
    tab->e[tab->n++] = WeightImages::E{src, mode, 256, 256, img};
    J.j[J.n++] = PrepJob{src, img, 256, 256, mode};
  };
  for (int li = 0; li < 2; ++li) {
    if (mab_kind(s.m1[li]) == 1 && s.m1[li].d == 256 && s.m1[li].nk == 32) {
      const pca_mab_params pm = params_at(p, L.mab1[li]);
      const bool small = s.m1[li].dq <= 4;
      add(pm.wo, mab1_fwd_wo_mode(s.m1[li]));                 // forward: fc_o (and fc_q of a d -> d block)
      if (!small) add(pm.wq, 0);
      add(pm.wo, d256_bwd_wo_mode());                        // backward
      if (!small) add(pm.wq, d256_bwd_wq_mode());
    }
    if (mab_kind(s.m0[li]) == 2 && s.m0[li].d == 256 && s.m0[li].dk == 256) {
      const pca_mab_params pk = params_at(p, L.mab0[li]);
      add(pk.wk, 0); add(pk.wv, 0);                          // forward: fc_k / fc_v over the keys
      add(pk.wk, d256_bwd_kv_mode()); add(pk.wv, d256_bwd_kv_mode());
    }
  }
  // fp8 mode: the e4m3 images of the forward projections (fc_o of the many-queries blocks, fc_k / fc_v of
  // the d -> d few-queries block; natural layout: a block that wants another one converts its own)
  PrepF8Jobs F{};
  tab->nf8 = 0;
  if (c.mode == PCA_MODE_FP8 && w.img256f8 != nullptr) {
    auto add8 = [&](const float* src, int slot) {       // slot: index of the inverse scale
      if (tab->nf8 >= 8) return;
      const WeightImages::F8 e{src, 0, 256, 256, w.img256f8 + (size_t)tab->nf8 * 256 * 256,
                               w.inv256f8 + slot};
      tab->f8[tab->nf8++] = e;
      F.j[F.n++] = e;
    };
    for (int li = 0; li < 2; ++li) {
      if (mab_kind(s.m1[li]) == 1 && s.m1[li].d == 256 && s.m1[li].nk == 32)
        add8(params_at(p, L.mab1[li]).wo, 4 * li + 1);                 // [., o]
      if (mab_kind(s.m0[li]) == 2 && s.m0[li].d == 256 && s.m0[li].dk == 256) {
        add8(params_at(p, L.mab0[li]).wk, 4 * li + 2);                 // [k, v]: adjacent
        add8(params_at(p, L.mab0[li]).wv, 4 * li + 3);
      }
    }
  }
  if (!launch) return PCA_OK;
  if (jobs_out != nullptr) *jobs_out = J;
  else PCA_TRY(prep_jobs_launch(J, st));
  return prep_f8_jobs_launch(F, st);
}

// the d = 256 few-queries blocks' post stages: one deferred launch pair per step (per block in round 2)
inline bool defer256_on() { return true; }

int validate(const pca_st_config* c) {
  PCA_REQUIRE(c != nullptr, "st: null config");
  PCA_REQUIRE(c->B > 0 && c->N > 0 && c->din > 0 && c->d > 0 && c->h > 0 && c->m > 0 &&
                  c->k > 0 && c->C > 0,
              "st: non-positive extent");
  PCA_REQUIRE(c->d % c->h == 0, "st: d=%d not divisible by h=%d", c->d, c->h);
  PCA_REQUIRE(c->mode == PCA_MODE_F32 || c->mode == PCA_MODE_BF16 || c->mode == PCA_MODE_FP8,
              "st: unknown mode %d",
              c->mode);
  return PCA_OK;
}

int forward(const pca_st_config& c, const Layout& L, const Shapes& s, const float* p,
            const float* X, Ws& w, bool training, hipStream_t st, const PrepJobs* image_jobs = nullptr,
            const PmaHeadArgs* head = nullptr) {
  const void* in = X;
  // d = 256: the query side of all three few-queries blocks in the same launch (mab0_d256_prep_collect)
  const bool prep256 = training && s.m0[0].d == 256 && mab_kind(s.m0[0]) == 2 &&
                       mab_kind(s.m0[1]) == 2 && mab_kind(s.pma) == 2 && !pma_head_ok(s);
  struct PrepGuard {
    bool on;
    explicit PrepGuard(bool o) : on(o) {}
    ~PrepGuard() { if (on) mab0_d256_prep_done(false); }
  } prep_guard(prep256);
  if (training) {               // all weight images of the step in ONE launch
    PrepJobs J{};
    if (image_jobs != nullptr) J = *image_jobs;      // (d = 256: the step's image table, images256_prepare)
    for (int li = 0; li < 2; ++li)
      if (w.fused[li])
        isab_collect_prep(s.m0[li], params_at(p, L.mab0[li]), params_at(p, L.mab1[li]),
                          w.img[li], true, li == 1, &J);
    if (set128_on(c, s))           // the pair flags of the set-resident forward start every step at zero
      // (not the 16-byte header in front of them: word 0 counts expired spin-waits and is the CALLER's to
      // clear and read - pca_st_handoff_counter)
      J.j[J.n++] = PrepJob{nullptr, reinterpret_cast<__bf16*>(static_cast<char*>(w.set128_ws) + 16), 1,
                           (int)((set128_flag_bytes(c.B) - 16) / 2), 4};
    // (launched together with the query-side jobs below)
    // query-side preparation (Qp, G images) of every fused mab0 / PMA, also one launch
    Mab0PrepJobs MJ{};
    for (int li = 0; li < 2; ++li)
      if (w.fused[li]) {
        Mab0Saved v;
        mab0_carve_saved(s.m0[li], &v, w.saved[2 * li]);
        mab0_collect_prep(s.m0[li], p + L.I[li], params_at(p, L.mab0[li]), v, true, false, &MJ);
      }
    if (pma_head_ok(s)) {
      Mab0Saved v;
      mab0_carve_saved(s.pma, &v, w.saved[4]);
      mab0_collect_prep(s.pma, p + L.S, params_at(p, L.pma), v, true, true, &MJ);
    }
    if (prep256) {
      const pca_mab_shape* sh[3] = {&s.m0[0], &s.m0[1], &s.pma};
      const float* Iq[3] = {p + L.I[0], p + L.I[1], p + L.S};
      const pca_mab_params pr[3] = {params_at(p, L.mab0[0]), params_at(p, L.mab0[1]),
                                    params_at(p, L.pma)};
      void* sv[3] = {w.saved[0], w.saved[2], w.saved[4]};
      mab0_d256_prep_collect(3, sh, Iq, pr, sv, &MJ);
    }
    PCA_TRY(prep_all_launch(J, MJ, st));    // (takes a deferred pack along: pca_pack_defer)
    if (prep256) mab0_d256_prep_done(true);
  }
  PCA_TRY(pack_flush(st));                  // a deferred pack nobody took runs now, before X is read
  if (training && set128_on(c, s)) {
    // one launch: both ISABs and the PMA's attention partials, the set resident in one workgroup's LDS
    Set128FwdArgs a{};
    a.X = X; a.B = c.B; a.N = c.N; a.din = c.din;
    a.scale_log2e = 1.4426950408889634f / sqrtf((float)c.d);
    for (int li = 0; li < 2; ++li) {
      Mab0Saved v0;
      mab0_carve_saved(s.m0[li], &v0, w.saved[2 * li]);
      Mab1Saved v1;
      mab1_carve_saved(s.m1[li], &v1, w.saved[2 * li + 1]);
      const pca_mab_params p0 = params_at(p, L.mab0[li]), p1 = params_at(p, L.mab1[li]);
      const IsabImg& im = w.img[li];
      Set128Layer& S = a.L[li];
      S.Gf = v0.Gf; S.Gb = v0.Gb; S.Qp0 = v0.Qp; S.Wv0 = im.Wv0; S.Wv0f = p0.wv; S.bv0 = p0.bv;
      S.bo0 = p0.bo; S.Wo0 = im.Wo0; S.T = v0.T; S.LSE = v0.LSE; S.O0 = v0.O; S.Z0 = v0.Z; S.H = w.H[li];
      S.Wk1 = im.Wk1; S.Wv1 = im.Wv1; S.bk1 = p1.bk; S.bv1 = p1.bv;
      S.KpP = v1.KpP; S.VpP = v1.VpP; S.Kt = v1.Kt; S.Vt = v1.Vt;
      S.WqB = im.WqB; S.WqF = p1.wq; S.bq1 = p1.bq; S.WoP = im.WoP; S.bo1 = p1.bo;
      S.QpS = v1.QpS; S.OS = v1.OS; S.Y = reinterpret_cast<__bf16*>(w.Y[li]); S.mask = v1.mask;
    }
    Carver cs(w.set128_ws);
    a.flags = reinterpret_cast<uint32_t*>(cs.take<char>(set128_flag_bytes(c.B)));   // (cleared by k_prep_all)
    a.ex2 = cs.take<float>((size_t)c.B * 2 * 9216);
    a.exP = cs.take<float>((size_t)c.B * 2 * 528);
    if (head != nullptr) {             // the PMA epilogue, the classifier and the loss in the same launch
      a.fuse_head = 1;
      a.head = *head;
    }
    Mab0Saved vp;
    mab0_carve_saved(s.pma, &vp, w.saved[4]);
    a.Gpma = vp.Gb; a.TpP = vp.Tp; a.MpP = vp.Mp; a.LpP = vp.Lp; a.Sp = mab0_splits(s.pma);
    return set128_fwd_launch(a, st);
  }
  for (int li = 0; li < 2; ++li) {
    void* sv0 = training ? w.saved[2 * li] : nullptr;
    void* sv1 = training ? w.saved[2 * li + 1] : nullptr;
    if (training && w.fused[li]) {
      PCA_TRY(isab_bf16_fwd(s.m0[li], s.m1[li], p + L.I[li], in, params_at(p, L.mab0[li]),
                            params_at(p, L.mab1[li]), w.H[li], w.Y[li], sv0, sv1, w.scratch,
                            w.img[li], st));
      in = w.Y[li];
      continue;
    }
    // d = 256 training: the few-queries block ends in the per-set mid kernel, which also prepares
    // the K / V images of the many-queries block (mid256.hip; the blocks' saved areas are disjoint)
    const pca_mab_params p1 = params_at(p, L.mab1[li]);
    const bool mid = training && s.m0[li].d == 256 && mab_kind(s.m0[li]) == 2 && mab_kind(s.m1[li]) == 1;
    if (mid) mid256_arm(&s.m1[li], &p1, sv1);
    const int rc0 = mab_fwd_any(s.m0[li], p + L.I[li], in, params_at(p, L.mab0[li]), w.H[li], sv0,
                                w.scratch, st);                     // modules.py:52
    const int rc1 = rc0 != PCA_OK ? rc0
                                  : mab_fwd_any(s.m1[li], in, w.H[li], p1, w.Y[li], sv1, w.scratch,
                                                st);                // modules.py:53
    if (mid) mid256_arm(nullptr, nullptr, nullptr);
    PCA_TRY(rc1);
    in = w.Y[li];
  }
  if (training && pma_head_ok(s))                                         // modules.py:63
    // (its epilogue runs inside k_pma_head together with the classifier and the loss)
    PCA_TRY(mab0_bf16_fwd_ex(s.pma, p + L.S, w.Y[1], params_at(p, L.pma), w.P, w.saved[4],
                             w.scratch, PCA_F_PREP_DONE | (c.k == 1 ? PCA_F_SKIP_EPILOGUE : 0),
                             st));
  else
    PCA_TRY(mab_fwd_any(s.pma, p + L.S, w.Y[1], params_at(p, L.pma), w.P,
                        training ? w.saved[4] : nullptr, w.scratch, st));
  if (!training)
    PCA_TRY(linear_fwd_f32(w.P, p + L.wc, p + L.bc, w.logits, (int64_t)c.B * c.k, c.d, c.C,
                           st));                                    // models.py:40
  return PCA_OK;
}

}  // namespace
}  // namespace pca

extern "C" {

int64_t pca_st_param_count(const pca_st_config* c) {
  if (pca::validate(c) != PCA_OK) return -1;
  return pca::layout(*c).total;
}

int64_t pca_st_bucket_split(const pca_st_config* c) {
  if (pca::validate(c) != PCA_OK) return -1;
  return pca::layout(*c).enc1_begin;
}

size_t pca_st_ws_bytes(const pca_st_config* c, int training) {
  if (pca::validate(c) != PCA_OK) return 0;
  return pca::carve(*c, training, nullptr, nullptr);
}

int pca_st_handoff_counter(const pca_st_config* c, void* ws, uint32_t** counter) {
  PCA_TRY(pca::validate(c));
  PCA_REQUIRE(ws != nullptr && counter != nullptr, "st_handoff_counter: null pointer");
  pca::Ws w;
  pca::carve(*c, 1, &w, ws);
  *counter = static_cast<uint32_t*>(w.set128_ws);      // nullptr: no set-resident launch for this shape
  return PCA_OK;
}

int pca_st_ws_layout(const pca_st_config* c, int64_t* out) {
  PCA_TRY(pca::validate(c));
  PCA_REQUIRE(out != nullptr, "st_ws_layout: null pointer");
  pca::Ws w;
  char* const base = reinterpret_cast<char*>(256);      // (never dereferenced: offsets only)
  const size_t total = pca::carve(*c, 1, &w, base);
  for (int i = 0; i < 5; ++i) out[i] = reinterpret_cast<char*>(w.saved[i]) - base;
  for (int i = 0; i < 2; ++i) {
    out[5 + i] = reinterpret_cast<char*>(w.H[i]) - base;
    out[7 + i] = reinterpret_cast<char*>(w.Y[i]) - base;
  }
  out[9] = reinterpret_cast<char*>(w.scratch) - base;
  out[10] = (int64_t)total;
  return PCA_OK;
}

static int st_forward_impl(const pca_st_config* c, const float* params, const float* X,
                           const int32_t* lengths, float* logits, void* ws, void* stream) {
  PCA_TRY(pca::validate(c));
  PCA_REQUIRE(params && X && logits && ws, "st_forward: null pointer");
  hipStream_t st = pca::as_stream(stream);
  pca::Ws w;
  pca::carve(*c, 0, &w, ws);
  const pca::Layout L = pca::layout(*c);
  const pca::Shapes s = pca::shapes(*c, false, lengths);
  float* own = w.logits;
  w.logits = logits;
  (void)own;
  return pca::forward(*c, L, s, params, X, w, false, st);
}

int pca_st_forward(const pca_st_config* c, const float* params, const float* X,
                   const int32_t* lengths, float* logits, void* ws, void* stream) {
  PCA_TRY(pca::handoffs_empty("pca_st_forward", true));
  const int rc = st_forward_impl(c, params, X, lengths, logits, ws, stream);
  return rc != PCA_OK ? rc : pca::handoffs_empty("pca_st_forward (exit)", false);
}

static int st_train_fwd_bwd_impl(const pca_st_config* c, const float* params, const float* X,
                                 const int32_t* lengths, const int64_t* labels, float* grads,
                                 float* loss_out, float* stats,
                                 float* logits, float grad_scale, int phase, void* ws,
                                 void* stream) {
  PCA_TRY(pca::validate(c));
  PCA_REQUIRE(params && X && labels && grads && loss_out && ws,
              "st_train_fwd_bwd: null pointer");
  PCA_REQUIRE(c->k == 1, "st_train_fwd_bwd: the train step needs k == 1 (got %d)", c->k);
  PCA_REQUIRE(phase >= -1 && phase <= 1, "st_train_fwd_bwd: phase=%d", phase);
  hipStream_t st = pca::as_stream(stream);
  pca::Ws w;
  pca::carve(*c, 1, &w, ws);
  const pca::Layout L = pca::layout(*c);
  const pca::Shapes s = pca::shapes(*c, true, lengths);
  const float* p = params;
  float* g = grads;
  if (logits != nullptr) w.logits = logits;
  // terminal gradient reductions overlap the critical path on the helper stream; the guard
  // joins it back into `st` on every exit path
  struct SideGuard {
    hipStream_t st;
    explicit SideGuard(hipStream_t s) : st(s) { pca::terminal_enable(true); }
    ~SideGuard() { pca::terminal_join(st); pca::terminal_enable(false); }
  } side_guard(st);

  // shared-query gradients of the fused blocks of this call: one pair of launches at the end
  // (their inputs live in per-block workspaces, which stay untouched until then)
  pca::BwdDefer posts{};
  posts.slab_ws = w.wg_slabs;
  posts.slab_cap = w.wg_slab_bytes;
  // d = 256: all weight images of the step in one launch (phase 1 of a split step finds the images
  // of phase 0 still in place: the parameters do not change in between)
  pca::WeightImages images{};
  struct ImagesGuard {
    explicit ImagesGuard(const pca::WeightImages* t) { pca::weight_images_use(t); }
    ~ImagesGuard() { pca::weight_images_use(nullptr); }
  };
  pca::PrepJobs image_jobs{};
  PCA_TRY(pca::images256_prepare(*c, L, s, p, w, &images, phase != 1, st, &image_jobs));
  ImagesGuard images_guard(images.n > 0 ? &images : nullptr);
  // d = 256: the [B*m]-row weight-gradient jobs of all five blocks in one launch at the flush -
  // needs every block's operands in place until then: the hand-over form of enc.1 (its few-queries
  // block works in w.scratch) and a workspace of its own for enc.0's few-queries block
  const bool hand1 = s.m1[1].d == 256 && pca::mab_kind(s.m1[1]) == 1 && pca::mab_kind(s.m0[1]) == 2;
  const bool defer_wg = w.wg256_def != nullptr && hand1 && pca::defer256_on() &&
                        pca::mab_kind(s.pma) == 2 && s.pma.d == 256;
  if (defer_wg) posts.wg256_ws = w.wg256_def;
  if (phase != 1) {
    // the set-resident forward runs the head stages in its own tail (PCA_SET128_HEAD=0: as a launch)
    const char* he = getenv("PCA_SET128_HEAD");
    const bool fuse_head = pca::set128_on(*c, s) && c->C <= 64 && !(he != nullptr && he[0] == '0');
    pca::PmaHeadArgs head{};
    if (fuse_head)
      PCA_TRY(pca::pma_head_args(s.pma, pca::params_at(p, L.pma), w.saved[4], w.scratch, w.P,
                                 p + L.wc, p + L.bc, labels, c->C, grad_scale, w.logits, w.dlogits,
                                 w.dP, g + L.wc, g + L.bc, loss_out, stats, w.clsws, &posts, &head));
    PCA_TRY(pca::forward(*c, L, s, p, X, w, true, st, &image_jobs, fuse_head ? &head : nullptr));
    if (pca::pma_head_ok(s)) {
      // dec.0 epilogue + dec.1 (Linear) + mean cross-entropy forward and backward + dec.0
      // backward epilogue: one launch, one workgroup per set
      if (!fuse_head)
      PCA_TRY(pca::pma_head_launch(s.pma, pca::params_at(p, L.pma), w.saved[4], w.scratch, w.P,
                                   p + L.wc, p + L.bc, labels, c->C, grad_scale, w.logits,
                                   w.dlogits, w.dP, g + L.wc, g + L.bc, loss_out, stats,
                                   w.clsws, &posts, st));
      PCA_TRY(pca::mab0_bf16_bwd_ex(s.pma, p + L.S, w.Y[1], pca::params_at(p, L.pma),
                                    w.saved[4], w.dP, g + L.S, w.dY2, 0,
                                    pca::grads_at(g, L.pma), w.scratch, pca::PCA_F_SKIP_HEAD, st,
                                    &posts));
    } else {
    // dec.1 (Linear) + mean cross-entropy, forward and backward
    PCA_TRY(pca::cls_train_head(w.P, p + L.wc, p + L.bc, labels, c->B, c->d, c->C, grad_scale,
                                w.logits, w.dlogits, w.dP, g + L.wc, g + L.bc, loss_out, stats,
                                w.clsws, st, &posts));
    if (s.pma.d == 256 && pca::mab_kind(s.pma) == 2 && pca::defer256_on())   // post stages deferred
      PCA_TRY(pca::mab0_bf16_bwd_ex(s.pma, p + L.S, w.Y[1], pca::params_at(p, L.pma), w.saved[4],
                                    w.dP, g + L.S, w.dY2, 0, pca::grads_at(g, L.pma),
                                    w.scratch_pma, 0, st, &posts));
    else
    PCA_TRY(pca::mab_bwd_any(s.pma, p + L.S, w.Y[1], pca::params_at(p, L.pma), w.saved[4],
                             w.dP, g + L.S, w.dY2, 0, pca::grads_at(g, L.pma), w.scratch,
                             st));
    }
    // enc.1: mab1(Y1, H2) then mab0(I2, Y1); Y1 feeds both, so dY1 accumulates
    if (w.fused[1]) {
      PCA_TRY(pca::isab_bf16_bwd(s.m0[1], s.m1[1], p + L.I[1], w.Y[0], w.H[1],
                                 pca::params_at(p, L.mab0[1]), pca::params_at(p, L.mab1[1]),
                                 w.saved[2], w.saved[3], w.dY2, g + L.I[1], w.dY1,
                                 pca::grads_at(g, L.mab0[1]), pca::grads_at(g, L.mab1[1]),
                                 w.scratch_bw[1], w.img[1], st, &posts));
    } else {
    // d = 256: mab1's fc_q weight-gradient job is handed to the few-queries block, whose two jobs
    // read the same Y1 (wgrad256_handoff, d256_host.hip); the two blocks then need separate
    // workspaces - mab0 takes the forward / PMA scratch, which is free by now
    const bool hand = s.m1[1].d == 256 && pca::mab_kind(s.m1[1]) == 1 && pca::mab_kind(s.m0[1]) == 2;
    struct HandGuard {
      bool on;
      explicit HandGuard(bool o) : on(o) { if (on) pca::wgrad256_handoff_arm(true); }
      ~HandGuard() { if (on) pca::wgrad256_handoff_arm(false); }
    } hand_guard(hand);
    if (defer_wg)
      PCA_TRY(pca::mab1_bf16_bwd_ex(s.m1[1], w.Y[0], w.H[1], pca::params_at(p, L.mab1[1]),
                                    w.saved[3], w.dY2, w.dY1, w.dH, 0, pca::grads_at(g, L.mab1[1]),
                                    w.scratch_bw[1], 0, st, nullptr, nullptr, 0, nullptr, &posts));
    else
    PCA_TRY(pca::mab_bwd_any(s.m1[1], w.Y[0], w.H[1], pca::params_at(p, L.mab1[1]),
                             w.saved[3], w.dY2, w.dY1, w.dH, 0, pca::grads_at(g, L.mab1[1]),
                             w.scratch_bw[1], st));
    if (hand && pca::defer256_on())   // (its post stage waits for the flush: w.scratch stays untouched)
      PCA_TRY(pca::mab0_bf16_bwd_ex(s.m0[1], p + L.I[1], w.Y[0], pca::params_at(p, L.mab0[1]),
                                    w.saved[2], w.dH, g + L.I[1], w.dY1, 1,
                                    pca::grads_at(g, L.mab0[1]), w.scratch, 0, st, &posts));
    else
    PCA_TRY(pca::mab_bwd_any(s.m0[1], p + L.I[1], w.Y[0], pca::params_at(p, L.mab0[1]),
                             w.saved[2], w.dH, g + L.I[1], w.dY1, 1,
                             pca::grads_at(g, L.mab0[1]), hand ? w.scratch : w.scratch_bw[1], st));
    if (hand && pca::wgrad256_handoff_pending())        // (nobody took it: run it on its own)
      PCA_TRY(pca::wgrad256_handoff_flush(w.scratch, st));
    }
  }
  if (phase != 0) {
    // enc.0: the set itself needs no gradient
    if (w.fused[0]) {
      PCA_TRY(pca::isab_bf16_bwd(s.m0[0], s.m1[0], p + L.I[0], X, w.H[0],
                                 pca::params_at(p, L.mab0[0]), pca::params_at(p, L.mab1[0]),
                                 w.saved[0], w.saved[1], w.dY1, g + L.I[0], nullptr,
                                 pca::grads_at(g, L.mab0[0]), pca::grads_at(g, L.mab1[0]),
                                 w.scratch_bw[0], w.img[0], st, &posts));
    } else {
    if (defer_wg)
      PCA_TRY(pca::mab1_bf16_bwd_ex(s.m1[0], X, w.H[0], pca::params_at(p, L.mab1[0]), w.saved[1],
                                    w.dY1, nullptr, w.dH, 0, pca::grads_at(g, L.mab1[0]),
                                    w.scratch_bw[0], 0, st, nullptr, nullptr, 0, nullptr, &posts));
    else
    PCA_TRY(pca::mab_bwd_any(s.m1[0], X, w.H[0], pca::params_at(p, L.mab1[0]), w.saved[1],
                             w.dY1, nullptr, w.dH, 0, pca::grads_at(g, L.mab1[0]),
                             w.scratch_bw[0], st));
    if (s.m0[0].d == 256 && pca::mab_kind(s.m0[0]) == 2 && pca::defer256_on())
      PCA_TRY(pca::mab0_bf16_bwd_ex(s.m0[0], p + L.I[0], X, pca::params_at(p, L.mab0[0]),
                                    w.saved[0], w.dH, g + L.I[0], nullptr, 0,
                                    pca::grads_at(g, L.mab0[0]),
                                    defer_wg ? w.scratch_m0 : w.scratch_bw[0], 0, st, &posts));
    else
    PCA_TRY(pca::mab_bwd_any(s.m0[0], p + L.I[0], X, pca::params_at(p, L.mab0[0]),
                             w.saved[0], w.dH, g + L.I[0], nullptr, 0,
                             pca::grads_at(g, L.mab0[0]), w.scratch_bw[0], st));
    }
  }
  return pca::bwd_defer_flush(posts, st);
}

int pca_st_train_fwd_bwd(const pca_st_config* c, const float* params, const float* X,
                         const int32_t* lengths, const int64_t* labels, float* grads,
                         float* loss_out, float* stats,
                         float* logits, float grad_scale, int phase, void* ws,
                         void* stream) {
  // phase 1 of a split step reads X again: the pack was consumed by phase 0
  PCA_TRY(pca::handoffs_empty("pca_st_train_fwd_bwd", phase != 1));
  const int rc = st_train_fwd_bwd_impl(c, params, X, lengths, labels, grads, loss_out, stats, logits,
                                       grad_scale, phase, ws, stream);
  return rc != PCA_OK ? rc : pca::handoffs_empty("pca_st_train_fwd_bwd (exit)", false);
}
}
