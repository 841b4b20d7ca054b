// Loss and optimiser of the train step (Code/settransformer.py:88-91,104-108):
// nn.CrossEntropyLoss (mean) forward+gradient in one launch, and torch.optim.Adam with
// coupled weight decay as one fused pass over a flat parameter vector.
#include "pca_common.h"

namespace pca {
namespace {

// one wave per sample
__global__ __launch_bounds__(256) void k_cross_entropy(const float* __restrict__ logits,
                                                        const int64_t* __restrict__ labels,
                                                        int B, int C, float grad_scale,
                                                        float* __restrict__ loss_out,
                                                        float* __restrict__ dlogits,
                                                        float* __restrict__ stats) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const float* x = logits + (int64_t)b * C;
  float m = -INFINITY;
  int am = 0x7fffffff;
  for (int j = lane; j < C; j += 64) {
    const float v = x[j];
    if (v > m) { m = v; am = j; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float om = __shfl_xor(m, o, 64);
    const int oa = __shfl_xor(am, o, 64);
    if (om > m || (om == m && oa < am)) { m = om; am = oa; }
  }
  float s = 0.f;
  for (int j = lane; j < C; j += 64) s += expf(x[j] - m);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const int64_t y = labels[b];
  const float lse = m + logf(s);
  const float inv_s = 1.f / s;
  const float gs = grad_scale / (float)B;
  if (dlogits != nullptr)
    for (int j = lane; j < C; j += 64) {
      const float pj = expf(x[j] - m) * inv_s;
      dlogits[(int64_t)b * C + j] = (pj - (j == y ? 1.f : 0.f)) * gs;
    }
  if (lane == 0) {
    const float li = lse - x[y];
    atomicAdd(loss_out, li / (float)B);
    if (stats != nullptr) {
      atomicAdd(&stats[0], li);
      if (am == (int)y) atomicAdd(&stats[1], 1.f);
    }
  }
}

__global__ void k_inc_step(int32_t* step) { *step += 1; }

__global__ __launch_bounds__(256) void k_adam(float* __restrict__ p,
                                              const float* __restrict__ g,
                                              float* __restrict__ m, float* __restrict__ v,
                                              int64_t n, float lr, float b1, float b2,
                                              float eps, float wd, float gscale,
                                              const int32_t* __restrict__ step) {
  const float t = (float)(*step);
  const float bc1 = 1.f - powf(b1, t);
  const float bc2 = 1.f - powf(b2, t);
  const float step_size = lr / bc1;
  const float inv_sqrt_bc2 = 1.f / sqrtf(bc2);
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const float w = p[i];
    const float gi = g[i] * gscale + wd * w;         // coupled L2 (torch.optim.Adam)
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
    p[i] = w - step_size * (mi / denom);
  }
}

}  // namespace

}  // namespace pca

extern "C" {

int pca_cross_entropy(const float* logits, const int64_t* labels, int B, int C,
                      float grad_scale, float* loss_out, float* dlogits, float* stats_out,
                      void* stream) {
  PCA_REQUIRE(logits && labels && loss_out, "cross_entropy: null pointer");
  PCA_REQUIRE(B > 0 && C > 0, "cross_entropy: B=%d C=%d", B, C);
  hipStream_t st = pca::as_stream(stream);
  PCA_TRY(pca::fill_zero(loss_out, 1, st));
  hipLaunchKernelGGL(pca::k_cross_entropy, dim3((unsigned)pca::cdiv(B, 4)), dim3(256), 0, st,
                     logits, labels, B, C, grad_scale, loss_out, dlogits, stats_out);
  return pca::check_launch("k_cross_entropy");
}

int pca_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                  int64_t n, float lr, float beta1, float beta2, float eps,
                  float weight_decay, float grad_scale, int32_t* step_count_dev,
                  void* stream) {
  PCA_REQUIRE(param && grad && exp_avg && exp_avg_sq && step_count_dev,
              "adam_step: null pointer");
  PCA_REQUIRE(n >= 0, "adam_step: n=%lld", (long long)n);
  hipStream_t st = pca::as_stream(stream);
  hipLaunchKernelGGL(pca::k_inc_step, dim3(1), dim3(1), 0, st, step_count_dev);
  PCA_TRY(pca::check_launch("k_inc_step"));
  if (n == 0) return PCA_OK;
  int64_t blocks = pca::cdiv(n, 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(pca::k_adam, dim3((unsigned)blocks), dim3(256), 0, st, param, grad,
                     exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, grad_scale,
                     step_count_dev);
  return pca::check_launch("k_adam");
}
}
