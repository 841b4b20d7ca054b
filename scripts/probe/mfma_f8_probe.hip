// v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 operands: (1) does "lane (r, g) holds the 32
// consecutive k = 32 g .. 32 g + 31 of row / column r, scales = 2^0" give D = A B with the
// D layout of the other 16x16 MFMAs?  (2) cycles per instruction against v_mfma_f32_16x16x32_bf16
// and the unscaled fp8 form (s_memtime around 4096 back-to-back instructions, one wave per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

__device__ unsigned pack4_f8(float a, float b, float c, float d) {
  int p = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  p = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, p, true);
  return (unsigned)p;
}
__global__ void k128(const float* A, const float* Bt, float* D) {   // A[16][128], Bt[col][128]
  int l = threadIdx.x, r = l & 15, g = l >> 4;
  v8i a, b;
  for (int w = 0; w < 8; ++w) {
    const float* pa = A + r * 128 + 32 * g + 4 * w;
    const float* pb = Bt + r * 128 + 32 * g + 4 * w;
    a[w] = (int)pack4_f8(pa[0], pa[1], pa[2], pa[3]);
    b[w] = (int)pack4_f8(pb[0], pb[1], pb[2], pb[3]);
  }
  v4f c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  for (int i = 0; i < 4; ++i) D[(4 * g + i) * 16 + r] = c[i];
}
template <int KIND>
__global__ void krate(long long* out, float* sink) {
  v8i a, b;
  for (int w = 0; w < 8; ++w) { a[w] = 0x38383838 + threadIdx.x; b[w] = 0x3c3c3838 ^ threadIdx.x; }
  bf16x8 ha, hb;
  for (int w = 0; w < 8; ++w) { ha[w] = (__bf16)(1.f + w + threadIdx.x); hb[w] = (__bf16)(0.5f * w); }
  v4f c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < 1024; ++i) {
    if (KIND == 0) {
      c0 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c0, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      c1 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c1, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      c2 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c2, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      c3 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c3, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    } else if (KIND == 1) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, c3, 0, 0, 0);
    } else {
      long la = ((long)a[0] << 32) | (unsigned)a[1], lb = ((long)b[0] << 32) | (unsigned)b[1];
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(la, lb, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(la, lb, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(la, lb, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(la, lb, c3, 0, 0, 0);
    }
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * 64 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
int main() {
  std::vector<float> A(16 * 128), Bt(16 * 128), D(256), R(256);
  srand(1);
  const float vals[8] = {0.f, 0.5f, 1.f, -1.f, 2.f, -0.5f, 1.5f, -2.f};   // exact in e4m3
  for (auto& x : A) x = vals[rand() % 8];
  for (auto& x : Bt) x = vals[rand() % 8];
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      float s = 0;
      for (int k = 0; k < 128; ++k) s += A[i * 128 + k] * Bt[j * 128 + k];
      R[i * 16 + j] = s;
    }
  float *dA, *dB, *dD;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, Bt.size() * 4); hipMalloc(&dD, 1024);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, Bt.data(), Bt.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k128, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
  double e = 0;
  for (int i = 0; i < 256; ++i) e = fmax(e, fabs(D[i] - R[i]));
  printf("16x16x128 f8f6f4 (e4m3, scale 2^0), assumed layout: max |D - ref| = %g  (%s)\n", e, e == 0 ? "OK" : "MISMATCH");
  long long* dT; float* dS;
  hipMalloc(&dT, 8 * 1024); hipMalloc(&dS, 1024 * 64 * 4);
  const char* names[3] = {"scale_16x16x128_f8f6f4", "16x16x32_bf16", "16x16x32_fp8_fp8"};
  for (int kind = 0; kind < 3; ++kind) {
    for (int rep = 0; rep < 2; ++rep) {
      if (kind == 0) hipLaunchKernelGGL(krate<0>, dim3(1024), dim3(64), 0, 0, dT, dS);
      if (kind == 1) hipLaunchKernelGGL(krate<1>, dim3(1024), dim3(64), 0, 0, dT, dS);
      if (kind == 2) hipLaunchKernelGGL(krate<2>, dim3(1024), dim3(64), 0, 0, dT, dS);
    }
    hipDeviceSynchronize();
    std::vector<long long> t(1024);
    hipMemcpy(t.data(), dT, 8 * 1024, hipMemcpyDeviceToHost);
    double m = 0; for (auto v : t) m += v; m /= 1024;
    printf("%-24s %.1f cycles per instruction (one wave per SIMD, 4 accumulators)\n", names[kind], m / 4096.0);
  }
  return 0;
}
