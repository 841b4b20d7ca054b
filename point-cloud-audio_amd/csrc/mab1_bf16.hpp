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
  float* T;       // [B][R][dk] A X
  float* LSE;     // [B][R]     log2-domain
  float *O, *Z;   // [B][m][d]
};
size_t mab0_carve_saved(const pca_mab_shape& s, Mab0Saved* out, void* base);

// fp32 weight -> bf16 image; mode 0 natural, 1 K-permuted, 2 transposed + K-permuted
int prep_weight(const float* src, __bf16* dst, int rows, int cols, int mode, hipStream_t st);

}  // namespace pca
