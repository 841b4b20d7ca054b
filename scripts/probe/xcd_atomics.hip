// Probe: (1) HW_REG_XCC_ID per workgroup, (2) cost and correctness of fp32 atomic adds of many
// workgroups onto one [16384] accumulator: device scope vs workgroup scope into per-XCD copies.
// build: hipcc -O3 --offload-arch=gfx950 scripts/probe/xcd_atomics.hip -o gpurun_out/xcd_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ int xcc_id() {
  // s_getreg_b32 hwreg(HW_REG_XCC_ID = 20, offset 0, size 4)
  return __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) & 15;
}

__global__ void k_ids(int* out) {
  if (threadIdx.x == 0) out[blockIdx.x] = xcc_id();
}

template <int MODE>   // 0: device-scope atomics on one copy; 1: workgroup-scope on the XCD's copy
__global__ __launch_bounds__(256) void k_add(float* acc, int n) {
  float* dst = acc;
  if (MODE == 1) dst = acc + (size_t)xcc_id() * n;
  for (int i = threadIdx.x; i < n; i += 256) {
    if (MODE == 0) atomicAdd(&dst[i], 1.0f);
    else __hip_atomic_fetch_add(&dst[i], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}

int main() {
  const int n = 16384, W = 128, reps = 20;
  int* ids; hipMalloc(&ids, 1024 * sizeof(int));
  k_ids<<<1024, 64>>>(ids);
  std::vector<int> h(1024); hipMemcpy(h.data(), ids, 1024 * sizeof(int), hipMemcpyDeviceToHost);
  printf("xcc ids of workgroups 0..31:");
  for (int i = 0; i < 32; ++i) printf(" %d", h[i]);
  int mism = 0; for (int i = 0; i < 1024; ++i) mism += (h[i] != i % 8);
  printf("\nworkgroups whose xcc id != id %% 8: %d of 1024\n", mism);
  float* acc; hipMalloc(&acc, 8 * n * sizeof(float));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode) {
    hipMemset(acc, 0, 8 * n * sizeof(float));
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) {
      if (mode == 0) k_add<0><<<W, 256>>>(acc, n); else k_add<1><<<W, 256>>>(acc, n);
    }
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<float> r(8 * n); hipMemcpy(r.data(), acc, 8 * n * sizeof(float), hipMemcpyDeviceToHost);
    double bad = 0;
    for (int i = 0; i < n; ++i) {
      double s = 0; for (int c = 0; c < (mode ? 8 : 1); ++c) s += r[c * n + i];
      if (s != (double)W * reps) bad += 1;
    }
    printf("mode %d: %.2f us per launch of %d workgroups x %d atomics, wrong sums: %.0f\n", mode,
           ms * 1e3 / reps, W, n, bad);
  }
  return 0;
}
