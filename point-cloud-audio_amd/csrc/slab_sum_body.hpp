// Fixed-order sum of per-workgroup partial tensors: out[n] (+)= sum_w slabs[w * stride + i].
// Shared by the stand-alone launch (k_slab_sum_jobs, d256_bf16.hip) and by the kernels that carry
// such sums as extra workgroup rows ("riders": k_wgrad128, k_terminal1) so that the reduction costs
// no launch of its own.  Uses the first 256 threads of the workgroup; EVERY thread of the workgroup
// must call it (one barrier inside).  `red` = 4 x 64 float4 of LDS.
#pragma once
#include "mab1_bf16.hpp"

namespace pca {

__device__ __forceinline__ void slab_sum_body(const SlabSumJob& j, int bx, int tid, float4* red) {
  if (bx * 256 >= j.n) return;                       // (uniform per workgroup)
  const int64_t stride = j.stride > 0 ? j.stride : j.n;
  const int sg = (tid >> 6) & 3, c = tid & 63;
  const int i = bx * 256 + 4 * c;
  const bool on = tid < 256 && i < j.n;
  float4 t = {0.f, 0.f, 0.f, 0.f};
  if (on) {
    // four lane groups take every fourth slab, eight 16-byte loads in flight, then a fixed-order merge
    const float* s = j.slabs + i;
    int w = sg;
    for (; w + 28 < j.S; w += 32) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(s + (w + 4 * u) * stride);
#pragma unroll
      for (int u = 0; u < 8; ++u) { t.x += v[u].x; t.y += v[u].y; t.z += v[u].z; t.w += v[u].w; }
    }
    for (; w < j.S; w += 4) {
      const float4 v = *reinterpret_cast<const float4*>(s + w * stride);
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
  }
  if (tid < 256) red[sg * 64 + c] = t;
  __syncthreads();
  if (on && sg == 0) {
    float4 o = j.accumulate ? *reinterpret_cast<float4*>(j.out + i) : float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = red[q * 64 + c];
      o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
    }
    *reinterpret_cast<float4*>(j.out + i) = o;
  }
}

}  // namespace pca
