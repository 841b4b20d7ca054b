// Set-resident kernels of the d = 128 / 4 heads / m = 16 / k = 1 train step (BASELINE configs[1]):
// a PAIR of workgroups carries a whole set through ISAB -> ISAB -> PMA (set_transformer-master/
// modules.py:51-53, 62-63 with :19-33 inside; Code/models.py:34-44), each its half of the set's
// [N, 128] activations resident in LDS.  Declarations shared by set128_fwd.hip and the ST engine.
#pragma once
#include "mab1_bf16.hpp"

namespace pca {

// everything one ISAB of the set-resident forward reads and writes (pointers into the engine's
// weight images, the blocks' saved areas and the parameter vector; layouts are those the per-block
// kernels use, so the backward kernels read what this forward saved)
struct Set128Layer {
  // mab0 = MAB(I, X): few shared queries
  const float* Gf;        // [64][dk] fp32, sl2e folded in      (layer 1: dk = din <= 4)
  const __bf16* Gb;       // [64][128] bf16                      (layer 2)
  const float* Qp0;       // [16][128] fc_q(I)
  const __bf16* Wv0;      // [128][128] natural bf16 image       (layer 2)
  const float* Wv0f;      // [128][dk] fp32                      (layer 1)
  const float *bv0, *bo0;
  const __bf16* Wo0;      // natural
  float *T, *LSE;         // saved: [B][64][dk], [B][64]
  float *O0, *Z0, *H;     // [B][16][128] fp32
  // mab1 = MAB(X, H): many queries
  const __bf16 *Wk1, *Wv1;   // natural
  const float *bk1, *bv1;
  __bf16 *KpP, *VpP, *Kt, *Vt;
  const __bf16* WqB;      // natural                              (layer 2)
  const float* WqF;       // [128][dq] fp32                       (layer 1)
  const float* bq1;
  const __bf16* WoP;      // K-permuted
  const float* bo1;
  __bf16 *QpS, *OS, *Y;   // saved Qp (layer 2 only), saved O, the layer's output [B][N][128]
  uint32_t* mask;         // ReLU mask words, mab1_mask_index<128> layout
};

struct Set128FwdArgs {
  const float* X;         // [B][N][din] fp32
  int B, N, din;
  float scale_log2e;
  Set128Layer L[2];
  // pair exchange (scratch; `flags` is zeroed by the step's preparation launch):
  uint32_t* flags;        // [4] header (word 0 counts spin timeouts) + [B][2 exchanges][2 halves]
  float* ex2;             // [B][2][8192 + 1024] layer-2 attention partials (T as float4 [8][4][64], m / l)
  const __bf16* Gpma;     // [>= 16][128] bf16, rows >= 4 zero
  float *TpP, *MpP, *LpP; // PMA attention partials [B][Sp][4][128], [B][Sp][4] x 2 (read by k_pma_head1)
  int Sp;                 // 2 or 4
  // fuse_head != 0: the stages of k_pma_head1 (PMA epilogue, classifier, cross-entropy forward and
  // backward, PMA backward epilogue) run in this launch's tail on workgroup 0 of the pair, which gets
  // the partner's PMA partial through exP [B][2][528]; TpP / MpP / LpP are then not written
  int fuse_head;
  float* exP;
  PmaHeadArgs head;
};

// bytes of `flags` (the block the preparation launch clears) / of the whole exchange area
size_t set128_flag_bytes(int B);
size_t set128_fwd_ws_bytes(int B);
// N = 256 or 512 (each half a multiple of 128 points); din <= 4; 2 B workgroups resident at once
bool set128_shape_ok(int B, int N, int din, int d, int h, int m, int k);
int set128_fwd_launch(const Set128FwdArgs& a, hipStream_t st);

}  // namespace pca
