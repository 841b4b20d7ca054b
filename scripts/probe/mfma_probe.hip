// Verifies the lane layouts assumed by the fused kernels with exact small-integer data.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// test 1: D[16x16] = A[16x32] * B[32x16], A row-major [16][32], B given as Bt[col][k]
__global__ void k32(const float* A, const float* Bt, float* D) {
  int l = threadIdx.x, r = l & 15, g = l >> 4;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)A[r * 32 + 8 * g + j]; b[j] = (__bf16)Bt[r * 32 + 8 * g + j]; }
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int i = 0; i < 4; ++i) D[(4 * g + i) * 16 + r] = c[i];
}
// test 2: 16x16x16: A[16][16], Bt[col][16]
__global__ void k16(const float* A, const float* Bt, float* D) {
  int l = threadIdx.x, r = l & 15, g = l >> 4;
  bf16x4 a, b;
  for (int j = 0; j < 4; ++j) { a[j] = (__bf16)A[r * 16 + 4 * g + j]; b[j] = (__bf16)Bt[r * 16 + 4 * g + j]; }
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
  for (int i = 0; i < 4; ++i) D[(4 * g + i) * 16 + r] = c[i];
}
// test 3: chain. X^T[32 feat x 16 pts] = W1[32x32] * In^T (two 16-row tiles), then
// Y^T[16 x 16pts] = W2[16 x 32] * X^T using the accumulators as the B operand with the
// k-slot permutation slot(g,j) -> feature (j<4 ? 4g+j : 16+4g+j-4); W2 read permuted.
__global__ void kchain(const float* W1, const float* In, const float* W2, float* Y) {
  int l = threadIdx.x, r = l & 15, g = l >> 4;
  bf16x8 b;
  for (int j = 0; j < 8; ++j) b[j] = (__bf16)In[r * 32 + 8 * g + j];      // In[pt][k]
  f32x4 x0 = {0,0,0,0}, x1 = {0,0,0,0};
  bf16x8 a0, a1;
  for (int j = 0; j < 8; ++j) { a0[j] = (__bf16)W1[r * 32 + 8 * g + j]; a1[j] = (__bf16)W1[(16 + r) * 32 + 8 * g + j]; }
  x0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b, x0, 0, 0, 0);   // features 0..15
  x1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b, x1, 0, 0, 0);   // features 16..31
  bf16x8 xb, w2;
  for (int j = 0; j < 4; ++j) { xb[j] = (__bf16)x0[j]; xb[4 + j] = (__bf16)x1[j]; }
  for (int j = 0; j < 8; ++j) {
    int feat = j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4);
    w2[j] = (__bf16)W2[r * 32 + feat];
  }
  f32x4 y = {0,0,0,0};
  y = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2, xb, y, 0, 0, 0);
  for (int i = 0; i < 4; ++i) Y[(4 * g + i) * 16 + r] = y[i];   // Y^T[outfeat][pt]
}
// test 4: ds_read_tr16_b64: LDS image [16 rows(points)][16 cols(features)] bf16, row stride 32 B.
// group of 16 lanes: lane 4q+p supplies address of row (r0+q), cols 4p..4p+3; lane i receives
// column i of the 4 rows.  We output what each lane got.
__global__ void ktr(const float* img, float* out) {
  __shared__ __attribute__((aligned(16))) __bf16 lds[16 * 16];
  int l = threadIdx.x;
  for (int i = l; i < 256; i += 64) lds[i] = (__bf16)img[i];
  __syncthreads();
  int grp = l >> 4, i16 = l & 15, q = i16 >> 2, p = i16 & 3;
  int row = 4 * grp + q;                       // group grp covers rows 4grp..4grp+3
  typedef __attribute__((address_space(3))) s16x4 lds_v4;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(&lds[row * 16 + 4 * p]));
  bf16x4 vb = __builtin_bit_cast(bf16x4, v);
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = (float)vb[e];
}

int main() {
  std::vector<float> A(16 * 32), Bt(16 * 32), D(256), W1(32 * 32), In(16 * 32), W2(16 * 32), img(256), o4(256);
  srand(1);
  auto rnd = []() { return (float)((rand() % 7) - 3); };
  for (auto& v : A) v = rnd(); for (auto& v : Bt) v = rnd(); for (auto& v : W1) v = (float)((rand() % 3) - 1);
  for (auto& v : In) v = rnd(); for (auto& v : W2) v = (float)((rand() % 3) - 1);
  for (int i = 0; i < 256; ++i) img[i] = (float)i;
  float *dA, *dB, *dD, *dW1, *dIn, *dW2, *dimg, *do4;
  hipMalloc(&dA, 4 * 512); hipMalloc(&dB, 4 * 512); hipMalloc(&dD, 4 * 256); hipMalloc(&dW1, 4 * 1024);
  hipMalloc(&dIn, 4 * 512); hipMalloc(&dW2, 4 * 512); hipMalloc(&dimg, 4 * 256); hipMalloc(&do4, 4 * 256);
  hipMemcpy(dA, A.data(), 4 * 512, hipMemcpyHostToDevice); hipMemcpy(dB, Bt.data(), 4 * 512, hipMemcpyHostToDevice);
  hipMemcpy(dW1, W1.data(), 4 * 1024, hipMemcpyHostToDevice); hipMemcpy(dIn, In.data(), 4 * 512, hipMemcpyHostToDevice);
  hipMemcpy(dW2, W2.data(), 4 * 512, hipMemcpyHostToDevice); hipMemcpy(dimg, img.data(), 4 * 256, hipMemcpyHostToDevice);
  int bad;
  k32<<<1, 64>>>(dA, dB, dD); hipMemcpy(D.data(), dD, 4 * 256, hipMemcpyDeviceToHost);
  bad = 0; for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float s = 0; for (int k = 0; k < 32; ++k) s += A[i * 32 + k] * Bt[j * 32 + k]; bad += (s != D[i * 16 + j]); }
  printf("16x16x32 mismatches: %d\n", bad);
  k16<<<1, 64>>>(dA, dB, dD); hipMemcpy(D.data(), dD, 4 * 256, hipMemcpyDeviceToHost);
  bad = 0; for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float s = 0; for (int k = 0; k < 16; ++k) s += A[i * 16 + k] * Bt[j * 16 + k]; bad += (s != D[i * 16 + j]); }
  printf("16x16x16 mismatches: %d\n", bad);
  kchain<<<1, 64>>>(dW1, dIn, dW2, dD); hipMemcpy(D.data(), dD, 4 * 256, hipMemcpyDeviceToHost);
  bad = 0;
  for (int o = 0; o < 16; ++o) for (int p = 0; p < 16; ++p) {
    float s = 0;
    for (int f = 0; f < 32; ++f) { float x = 0; for (int k = 0; k < 32; ++k) x += W1[f * 32 + k] * In[p * 32 + k]; s += W2[o * 32 + f] * x; }
    bad += (s != D[o * 16 + p]);
  }
  printf("chain mismatches: %d\n", bad);
  ktr<<<1, 64>>>(dimg, do4); hipMemcpy(o4.data(), do4, 4 * 256, hipMemcpyDeviceToHost);
  // expectation: lane l (grp, i) element e == img[(4grp+e)*16 + i]
  bad = 0; for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) bad += (o4[l * 4 + e] != img[(4 * (l >> 4) + e) * 16 + (l & 15)]);
  printf("tr16_b64 mismatches: %d  (lane0: %g %g %g %g, lane17: %g %g %g %g)\n", bad, o4[0], o4[1], o4[2], o4[3], o4[68], o4[69], o4[70], o4[71]);
  return 0;
}
