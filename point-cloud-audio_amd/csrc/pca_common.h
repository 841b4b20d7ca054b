// Internal helpers shared by the HIP translation units of libpca_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "pca_hip.h"

namespace pca {

// thread-local error string (the only mutable global state of the library)
void set_error(const char* fmt, ...);

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return PCA_ELAUNCH;
  }
  return PCA_OK;
}

#define PCA_REQUIRE(cond, ...)        \
  do {                                \
    if (!(cond)) {                    \
      ::pca::set_error(__VA_ARGS__);  \
      return PCA_EINVAL;              \
    }                                 \
  } while (0)

#define PCA_TRY(expr)                 \
  do {                                \
    int _rc = (expr);                 \
    if (_rc != PCA_OK) return _rc;    \
  } while (0)

inline size_t align256(size_t n) { return (n + 255) & ~size_t(255); }

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// bump allocator over a caller-provided block
struct Carver {
  char* base;
  size_t off = 0;
  explicit Carver(void* p) : base(reinterpret_cast<char*>(p)) {}
  template <typename T>
  T* take(size_t count) {
    T* r = reinterpret_cast<T*>(base + off);
    off += align256(count * sizeof(T));
    return r;
  }
};

// ---- measurement hook (see pca_prof_start in pca_hip.h) -------------------------
// Usage in a launcher:  ProfScope ps(PCA_K_X, stream, flops, bytes);  <launch>;  ps.end();
struct ProfScope {
  int slot = -1;
  hipStream_t st;
  ProfScope(int kernel_id, hipStream_t stream, double flops, double bytes);
  void end();
};

// ---- helper stream for work that is off the critical path ------------------------------
// While enabled (by the ST engine, per calling thread), launch sites of terminal gradient
// reductions call terminal_stream(main): it makes the helper stream wait for everything
// enqueued on `main` so far (event fork) and returns the helper stream, so the reduction
// overlaps whatever `main` does next.  terminal_join(main) makes `main` wait for the helper
// stream.  Both are plain event record / wait operations, so a stream capture of `main`
// captures the fork and the join as graph edges.  Disabled: returns `main` itself.
void terminal_enable(bool on);
hipStream_t terminal_stream(hipStream_t main);
void terminal_join(hipStream_t main);

// ---- per-thread hand-off state (documented in include/pca_hip.h, "Per-thread state") ----------------
// Every hand-off between two internal calls of ONE public entry (the step's weight-image table, the
// d = 256 mid-stage arm / K-V-ready flags, the query-side "prepared" flag, the weight-gradient job
// hand-over) must be empty when a public pca_* call is entered and when it returns; the deferred pack is
// the one hand-off BETWEEN public calls and is allowed at the entry of its documented consumers only.
bool mid256_pending();
bool mab0_d256_prep_pending();
bool weight_images_active();
bool wgrad256_handoff_pending();
bool pack_pending();
// PCA_EINVAL with a message naming the stale hand-off (a previous call on this thread left it behind:
// an error path that skipped its guard, or a pack armed and never consumed)
int handoffs_empty(const char* where, bool pack_allowed);

// ---- internal launchers used across translation units -------------------
int gemm_f32(const pca_gemm_desc& g, const float* A, const float* B, const float* bias,
             float* C, hipStream_t st);
// same contract, operands rounded to bf16 on the way into the MFMA (gemm_bf16.hip)
int gemm_bf16(const pca_gemm_desc& g, const float* A, const float* B, const float* bias,
              float* C, hipStream_t st);
// same contract, operands as hi + lo bf16 pairs (three MFMAs per K step): fp32-level results
int gemm_bf16_hl(const pca_gemm_desc& g, const float* A, const float* B, const float* bias,
                 float* C, hipStream_t st);
int softmax_rows(float* X, int64_t rows, int n, float scale, hipStream_t st,
                 const int32_t* lengths = nullptr, int64_t rows_per_set = 0);
int layernorm_fwd(const float* X, const float* w, const float* b, float* Y, float* mean,
                  float* rstd, int64_t rows, int d, hipStream_t st);
int layernorm_bwd(const float* dY, const float* X, const float* mean, const float* rstd,
                  const float* w, float* dX, float* dw, float* db, int64_t rows, int d,
                  hipStream_t st);
int softmax_bwd_rows(const float* A, float* dA, int64_t rows, int n, float scale,
                     hipStream_t st);
int colsum_parts(const float* X, int64_t rows, int cols, float* part, int* nparts, hipStream_t st);
int colsum(const float* X, int64_t rows, int cols, float* out, int accumulate,
           hipStream_t st);
// Y = O + relu(Z)
int add_relu(const float* O, const float* Z, float* Y, int64_t n, hipStream_t st);
// dZ = dY * [Z > 0]
int relu_bwd(const float* dY, const float* Z, float* dZ, int64_t n, hipStream_t st);
int relu_bwd_copy(const float* dY, const float* Z, float* dZ, float* dO, int64_t n, hipStream_t st);
// dst[r, :] = src[(r % src_rows), :]   (broadcast copy when src_rows < rows)
int copy_rows(const float* src, int64_t src_rows, float* dst, int64_t rows, int64_t cols,
              hipStream_t st);
// dst += src (n elements)
int add_inplace(float* dst, const float* src, int64_t n, hipStream_t st);
int fill_zero(float* dst, int64_t n, hipStream_t st);

}  // namespace pca
