// Loss and optimiser of the train step (Code/settransformer.py:88-91,104-108):
// nn.CrossEntropyLoss (mean) forward+gradient in one launch, and torch.optim.Adam with
// coupled weight decay as one fused pass over a flat parameter vector.
#include "mab1_bf16.hpp"
#include "terminal_bodies.hpp"
#include "pma_head_bodies.hpp"

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

// ---------------------------------------------------------------------------------
// Fused classifier head of the train step (Code/models.py:40 + Code/settransformer.py:104):
//   k_cls_fwd_bwd  one 128-thread workgroup per sample: logits = P Wc^T + bc, softmax,
//                  per-sample loss / correctness, dlogits = (softmax - onehot) gs / B,
//                  dP = dlogits Wc
//   k_cls_wgrad    dWc += dlogits^T P, dbc += colsum(dlogits); block 0 also reduces the
//                  per-sample losses (deterministic sum) into loss_out and the counters
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void k_cls_fwd_bwd(
    const float* __restrict__ P, const float* __restrict__ Wc, const float* __restrict__ bc,
    const int64_t* __restrict__ labels, int B, int d, int C, float grad_scale,
    float* __restrict__ logits, float* __restrict__ dlogits, float* __restrict__ dP,
    float* __restrict__ lossv, float* __restrict__ corrv) {
  cls_fwd_bwd_body(P, Wc, bc, labels, B, d, C, grad_scale, logits, dlogits, dP, lossv, corrv, blockIdx.x);
}

__global__ __launch_bounds__(128) void k_cls_wgrad(
    const float* __restrict__ dlogits, const float* __restrict__ P,
    const float* __restrict__ lossv, const float* __restrict__ corrv, int B, int d, int C,
    float* __restrict__ dWc, float* __restrict__ dbc, float* __restrict__ loss_out,
    float* __restrict__ stats) {
  cls_wgrad_body(dlogits, P, lossv, corrv, B, d, C, dWc, dbc, loss_out, stats, blockIdx.x);
}

// step[0] = optimiser step count, step[1] = arrival ticket (zero between launches).  Every
// workgroup reads step[0] before it draws its ticket; the workgroup drawing the last ticket
// publishes step[0] + 1 and clears the ticket, so the count advances inside this launch
// (graph replays stay correct) without a second kernel.
__global__ __launch_bounds__(256) void k_adam(float* __restrict__ p, float* __restrict__ g,
                                              float* __restrict__ m, float* __restrict__ v,
                                              int64_t n, float lr, float b1, float b2,
                                              float eps, float wd, float gscale,
                                              int32_t* __restrict__ step, int zero_grad) {
  __shared__ int s_t;
  if (threadIdx.x == 0) s_t = step[0] + 1;
  __syncthreads();
  const int ti = s_t;
  const float t = (float)ti;
  const float bc1 = 1.f - powf(b1, t);
  const float bc2 = 1.f - powf(b2, t);
  const float step_size = lr / bc1;
  const float inv_sqrt_bc2 = 1.f / sqrtf(bc2);
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
#pragma unroll 4
  for (; i < n; i += stride) {
    const float w = p[i];
    const float gi = g[i] * gscale + wd * w;         // coupled L2 (torch.optim.Adam)
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
    p[i] = w - step_size * (mi / denom);
    if (zero_grad) g[i] = 0.f;
  }
  if (threadIdx.x == 0) {
    const int ticket = atomicAdd(&step[1], 1);
    if (ticket == (int)gridDim.x - 1) {
      step[1] = 0;
      step[0] = ti;
    }
  }
}

}  // namespace

// classifier head of the train step; ws needs 2*B floats
int cls_train_head(const float* P, const float* Wc, const float* bc, const int64_t* labels,
                   int B, int d, int C, float grad_scale, float* logits, float* dlogits,
                   float* dP, float* dWc, float* dbc, float* loss_out, float* stats, float* ws,
                   hipStream_t st, BwdDefer* defer) {
  float* lossv = ws;
  float* corrv = ws + B;
  hipLaunchKernelGGL(k_cls_fwd_bwd, dim3(B), dim3(128), (size_t)(d + C) * sizeof(float), st, P, Wc,
                     bc, labels, B, d, C, grad_scale, logits, dlogits, dP, lossv, corrv);
  PCA_TRY(check_launch("k_cls_fwd_bwd"));
  if (defer != nullptr) {          // rides in the phase's terminal launch
    defer->cls = ClsWgradArgs{dlogits, P, lossv, corrv, B, d, C, dWc, dbc, loss_out, stats};
    defer->has_cls = 1;
    return PCA_OK;
  }
  hipLaunchKernelGGL(k_cls_wgrad, dim3(C), dim3(128), 0, terminal_stream(st), dlogits, P, lossv,
                     corrv, B, d, C, dWc, dbc, loss_out, stats);
  return check_launch("k_cls_wgrad");
}

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

int pca_adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                  float lr, float beta1, float beta2, float eps, float weight_decay,
                  float grad_scale, int32_t* step_count_dev, int zero_grad, void* stream) {
  PCA_REQUIRE(param && grad && exp_avg && exp_avg_sq && step_count_dev,
              "adam_step: null pointer");
  PCA_REQUIRE(n >= 0, "adam_step: n=%lld", (long long)n);
  hipStream_t st = pca::as_stream(stream);
  // few, longer workgroups: the arrival tickets are serialised atomics on one address
  int64_t blocks = pca::cdiv(n, 256 * 8);
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;                      // n == 0 still advances the step count
  hipLaunchKernelGGL(pca::k_adam, dim3((unsigned)blocks), dim3(256), 0, st, param, grad,
                     exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, grad_scale,
                     step_count_dev, zero_grad);
  return pca::check_launch("k_adam");
}
}
