// Point-set packing (Code/dataset.py:50-54, 160-166) as a device body that runs either in the
// k_pack_2d / k_pack_3d launches of features.hip or as RIDER ROWS of the training step's k_prep_all
// launch (mab0_bf16.hip): the pack of a captured step is independent of the parameter-only
// preparation that opens the step, so the two share one launch (round 3: k_pack_* leaves the step's
// kernel table; pca_pack_defer in include/pca_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pca {

struct PackJob {
  int kind;                       // 0 none, 2 = 2-D (ESC_pc), 3 = 3-D (ESC_pc_temp)
  const float* spec;
  int64_t stride_f, stride_t, stride_s;
  const float *farr, *tarr;
  const int64_t* idx;             // index sequence (see pca_pack_points_*_seq)
  const int32_t *step_dev, *base_dev, *nt_valid;
  const int64_t* labels;
  int64_t* labels_out;
  int32_t* lengths_out;
  float* out;
  int B, F, Nt;
  int bx;                         // 256-thread blocks per set
};
inline int pack_blocks(const PackJob& j) { return j.bx * j.B; }

// block = set * bx + x : the work of workgroup (x, set) of k_pack_2d / k_pack_3d
__device__ __forceinline__ void pack_body(const PackJob& a, int block) {
  const int b = block / a.bx, x = block - b * a.bx;
  const int64_t* idx = a.idx;
  // device cursor: batch number (step - base) of a pre-staged index sequence, so that a
  // captured step needs no per-step index upload
  if (a.step_dev != nullptr) idx += (int64_t)(a.step_dev[0] - a.base_dev[0]) * a.B;
  const int64_t item = idx[b];
  if (a.kind == 2) {
    if (a.labels != nullptr && a.labels_out != nullptr && x == 0 && threadIdx.x == 0)
      a.labels_out[b] = a.labels[item];
    const int f = x * 256 + threadIdx.x;
    if (f >= a.F) return;
    float2 p;
    p.x = a.farr[f];
    p.y = a.spec[f * a.stride_f + item * a.stride_t];
    reinterpret_cast<float2*>(a.out)[(int64_t)b * a.F + f] = p;
    return;
  }
  // variable-size sets: chunk s holds nt_valid[s] <= Nt frames; time-major point order makes
  // its points a prefix of the padded set, the padding rows are written as zeros
  const int Nt = a.Nt, F = a.F;
  const int nt = a.nt_valid != nullptr ? (a.nt_valid[item] < Nt ? a.nt_valid[item] : Nt) : Nt;
  if (x == 0 && threadIdx.x == 0) {
    if (a.labels != nullptr && a.labels_out != nullptr) a.labels_out[b] = a.labels[item];
    if (a.lengths_out != nullptr) a.lengths_out[b] = nt * F;
  }
  const int p = x * 256 + threadIdx.x;               // point index = t*F + f (time-major)
  if (p >= F * Nt) return;
  const int t = p / F, f = p - t * F;
  float* o = a.out + ((int64_t)b * F * Nt + p) * 3;
  const bool ok = t < nt;
  o[0] = ok ? a.farr[f] : 0.f;
  o[1] = ok ? a.tarr[t] : 0.f;
  o[2] = ok ? a.spec[f * a.stride_f + t * a.stride_t + item * a.stride_s] : 0.f;
}

// hand-off of a deferred pack (features.hip; thread-local like the library's other hand-offs)
bool pack_stream_ok(hipStream_t st);  // the pending pack (if any) was submitted on `st`
bool pack_take(PackJob* out);          // true: *out is the pending pack, now the caller's to launch
int pack_flush(hipStream_t st);        // launches a pending pack on its own (no-op without one)

}  // namespace pca
