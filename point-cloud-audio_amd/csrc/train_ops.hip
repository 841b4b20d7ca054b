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
  // The kernel is one dependent chain of L2 round trips long (1.2 MB of parameters at configs[1]):
  // the step count and the first batch of operands are requested together, 16-byte accesses, each
  // thread two float4 quadruples in flight.
  const int ti = step[0] + 1;
  const int64_t n4 = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  float4* p4 = reinterpret_cast<float4*>(p);
  float4* g4 = reinterpret_cast<float4*>(g);
  float4* m4 = reinterpret_cast<float4*>(m);
  float4* v4 = reinterpret_cast<float4*>(v);
  const bool vec = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0;
  float4 w0, g0, m0, v0, w1, g1, m1, v1;
  const bool h0 = vec && i < n4, h1 = vec && i + stride < n4;
  if (h0) { w0 = p4[i]; g0 = g4[i]; m0 = m4[i]; v0 = v4[i]; }
  if (h1) { w1 = p4[i + stride]; g1 = g4[i + stride]; m1 = m4[i + stride]; v1 = v4[i + stride]; }
  const float t = (float)ti;
  const float bc1 = 1.f - powf(b1, t);
  const float bc2 = 1.f - powf(b2, t);
  const float step_size = lr / bc1;
  const float inv_sqrt_bc2 = 1.f / sqrtf(bc2);
  auto upd = [&](float& w, float& gg, float& mm, float& vv) {
    const float gi = gg * gscale + wd * w;            // coupled L2 (torch.optim.Adam)
    mm = b1 * mm + (1.f - b1) * gi;
    vv = b2 * vv + (1.f - b2) * gi * gi;
    const float denom = sqrtf(vv) * inv_sqrt_bc2 + eps;
    w = w - step_size * (mm / denom);
    if (zero_grad) gg = 0.f;
  };
  auto upd4 = [&](float4& w, float4& gg, float4& mm, float4& vv) {
    upd(w.x, gg.x, mm.x, vv.x); upd(w.y, gg.y, mm.y, vv.y);
    upd(w.z, gg.z, mm.z, vv.z); upd(w.w, gg.w, mm.w, vv.w);
  };
  if (h0) {
    upd4(w0, g0, m0, v0);
    p4[i] = w0; m4[i] = m0; v4[i] = v0;
    if (zero_grad) g4[i] = g0;
  }
  if (h1) {
    upd4(w1, g1, m1, v1);
    p4[i + stride] = w1; m4[i + stride] = m1; v4[i + stride] = v1;
    if (zero_grad) g4[i + stride] = g1;
  }
  if (vec) {
    for (int64_t k = i + 2 * stride; k < n4; k += stride) {
      float4 w = p4[k], gg = g4[k], mm = m4[k], vv = v4[k];
      upd4(w, gg, mm, vv);
      p4[k] = w; m4[k] = mm; v4[k] = vv;
      if (zero_grad) g4[k] = gg;
    }
  }
  // the tail (n % 4 elements), or everything when a pointer is not 16-byte aligned
  for (int64_t k = (vec ? (n4 << 2) : 0) + i; k < n; k += stride) {
    float w = p[k], gg = g[k], mm = m[k], vv = v[k];
    upd(w, gg, mm, vv);
    p[k] = w; m[k] = mm; v[k] = vv;
    if (zero_grad) g[k] = gg;
  }
  // every wave of the workgroup has its step[0] (loaded at the top, consumed by the updates above)
  // before thread 0 draws the ticket that may publish the next count
  __syncthreads();
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
  // (two float4 quadruples per thread in the first pass)
  int64_t blocks = pca::cdiv(n, 256 * 8);
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;                      // n == 0 still advances the step count
  hipLaunchKernelGGL(pca::k_adam, dim3((unsigned)blocks), dim3(256), 0, st, param, grad,
                     exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, grad_scale,
                     step_count_dev, zero_grad);
  return pca::check_launch("k_adam");
}
}
