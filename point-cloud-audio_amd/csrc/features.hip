// STFT log-magnitude and point-set packing: the data side of the hot path.
//
//   k_stft_logmag  : Code/settransformer.py:49-50  librosa.stft(...)/Nfft ; log(1e-8+|.|)
//   k_pack_2d      : Code/dataset.py:50-54         ESC_pc.__getitem__ for a whole batch
//   k_pack_3d      : Code/dataset.py:160-166       ESC_pc_temp.__getitem__ for a whole batch
//
// The reference does this per item in Python/numpy on the host (45.6 us / 331 us per set);
// here the spectrogram stays resident in HBM and a batch is assembled by one launch.
#include "pca_common.h"
#include "pack_body.hpp"

#include <mutex>

namespace pca {
namespace {

// One workgroup per frame.  In-place radix-2 decimation-in-time FFT in LDS, in float64:
// librosa computes the transform in double (numpy.fft) and only then rounds to complex64,
// so matching it to ~1e-7 in log-magnitude -- including the near-silent bins that the
// log(1e-8 + .) floor amplifies -- needs double butterflies; the op is a one-off pre-pass
// and stays HBM/LDS-bound.  LDS: n_fft complex doubles (data) + n_fft/2 (twiddles)
// = 24 B * n_fft (96 KiB at n_fft = 4096).  Frame t covers samples
// [t*hop - n_fft/2, t*hop + n_fft/2) of the reflect-padded signal (center=True).
// Batched form (blockIdx.y = clip): clip c is wave[wave_off[c] .. wave_off[c + 1]) and its frame t
// goes to output column frame_off[c] + t; a clip shorter than the longest one leaves the workgroups
// past its last frame idle.  wave_off == nullptr: one clip of L samples, columns from 0.
__global__ __launch_bounds__(256) void k_stft_logmag(const float* __restrict__ wave,
                                                      int64_t L, int n_fft, int log2n,
                                                      int win_length, int hop, int n_bins,
                                                      float* __restrict__ out,
                                                      int64_t stride_f, int64_t stride_t,
                                                      const int64_t* __restrict__ wave_off,
                                                      const int64_t* __restrict__ frame_off) {
  extern __shared__ __attribute__((aligned(16))) double2 lds_c[];
  double2* x = lds_c;                 // [n_fft]
  double2* tw = lds_c + n_fft;        // [n_fft/2]  exp(-2 pi i k / n_fft)
  const int tid = threadIdx.x;
  const int64_t t = blockIdx.x;
  if (wave_off != nullptr) {
    const int c = blockIdx.y;
    const int64_t w0 = wave_off[c];
    L = wave_off[c + 1] - w0;
    if (t >= 1 + L / hop) return;     // (uniform over the workgroup)
    wave += w0;
    out += frame_off[c] * stride_t;
  }
  const int lpad = (n_fft - win_length) / 2;
  const int64_t start = t * hop - n_fft / 2;
  const int half = n_fft >> 1;

  for (int k = tid; k < half; k += 256) {
    double sn, cs;
    sincospi(-2.0 * (double)k / (double)n_fft, &sn, &cs);
    tw[k] = make_double2(cs, sn);
  }
  for (int n = tid; n < n_fft; n += 256) {
    int64_t src = start + n;
    if (src < 0) src = -src;
    if (src >= L) src = 2 * (L - 1) - src;
    if (src < 0) src = 0;  // only reachable when L <= n_fft/2 (rejected on the host)
    const int nw = n - lpad;
    double w = 0.0;
    if (nw >= 0 && nw < win_length)
      w = 0.5 - 0.5 * cospi(2.0 * (double)nw / (double)win_length);   // periodic Hann
    const int r = (int)(__brev((unsigned)n) >> (32 - log2n));          // bit-reversed slot
    x[r] = make_double2((double)wave[src] * w, 0.0);
  }
  __syncthreads();

  for (int s = 1; s <= log2n; ++s) {
    const int hm = 1 << (s - 1);               // half butterfly span
    const int tstride = n_fft >> s;            // twiddle index stride
    for (int j = tid; j < half; j += 256) {
      const int k = j & (hm - 1);
      const int i0 = ((j - k) << 1) + k;
      const int i1 = i0 + hm;
      const double2 w = tw[k * tstride];
      const double2 a = x[i0];
      const double2 b = x[i1];
      const double2 bw = make_double2(b.x * w.x - b.y * w.y, b.x * w.y + b.y * w.x);
      x[i0] = make_double2(a.x + bw.x, a.y + bw.y);
      x[i1] = make_double2(a.x - bw.x, a.y - bw.y);
    }
    __syncthreads();
  }

  const double inv = 1.0 / (double)n_fft;
  for (int f = tid; f < n_bins; f += 256) {
    const double2 v = x[f];
    // the reference rounds the spectrum to complex64 before |.| (librosa dtype=complex64)
    const float re = (float)(v.x * inv), im = (float)(v.y * inv);
    const float mag = sqrtf(re * re + im * im);
    out[f * stride_f + t * stride_t] = logf(1.0e-8f + mag);
  }
}

// Smith's band-limited interpolation (resampy's resample_f restated): one thread per output sample,
// both wings of the filter, table entries linearly interpolated, fp64 accumulation.  Per output sample
// 2 * num_zeros / min(1, ratio) input samples are read (L2-resident: neighbouring threads share them).
__global__ __launch_bounds__(256) void k_resample(const float* __restrict__ x, int64_t n_in, double ratio,
                                                  const double* __restrict__ win,
                                                  const double* __restrict__ delta, int nwin,
                                                  int num_table, float gain, float* __restrict__ y,
                                                  int64_t n_out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_out) return;
  const double scale = ratio < 1.0 ? ratio : 1.0;
  const int index_step = (int)(scale * num_table);
  const double time_register = (double)t / ratio;
  const int64_t n = (int64_t)time_register;
  double acc = 0.0;
  {   // left wing: x[n], x[n - 1], ...
    const double frac = scale * (time_register - (double)n);
    const double index_frac = frac * num_table;
    const int offset = (int)index_frac;
    const double eta = index_frac - offset;
    int64_t i_max = (nwin - offset) / index_step;
    if (n + 1 < i_max) i_max = n + 1;
    for (int64_t i = 0; i < i_max; ++i) {
      const int k = offset + (int)i * index_step;
      acc += (win[k] + eta * delta[k]) * (double)x[n - i];
    }
  }
  {   // right wing: x[n + 1], x[n + 2], ...
    const double frac = scale - scale * (time_register - (double)n);
    const double index_frac = frac * num_table;
    const int offset = (int)index_frac;
    const double eta = index_frac - offset;
    int64_t k_max = (nwin - offset) / index_step;
    if (n_in - n - 1 < k_max) k_max = n_in - n - 1;
    for (int64_t k2 = 0; k2 < k_max; ++k2) {
      const int k = offset + (int)k2 * index_step;
      acc += (win[k] + eta * delta[k]) * (double)x[n + k2 + 1];
    }
  }
  y[t] = (float)(acc * (double)gain);
}

__global__ __launch_bounds__(256) void k_pack(const PackJob a) {
  pack_body(a, blockIdx.y * a.bx + blockIdx.x);
}

thread_local bool t_pack_armed = false;
thread_local PackJob t_pack_pending{};
thread_local hipStream_t t_pack_stream = nullptr;   // stream the pending pack was submitted on

int pack_launch(const PackJob& j, hipStream_t st) {
  hipLaunchKernelGGL(k_pack, dim3((unsigned)j.bx, (unsigned)j.B), dim3(256), 0, st, j);
  return check_launch(j.kind == 2 ? "k_pack_2d" : "k_pack_3d");
}
// armed: remember the pack for the next k_prep_all launch of this thread instead of launching it
int pack_submit(const PackJob& j, hipStream_t st) {
  if (t_pack_armed) {
    if (t_pack_pending.kind != 0) PCA_TRY(pack_launch(t_pack_pending, st));   // (never two pending)
    t_pack_pending = j;
    t_pack_stream = st;
    return PCA_OK;
  }
  return pack_launch(j, st);
}

}  // namespace

bool pack_pending() { return t_pack_pending.kind != 0; }
bool pack_take(PackJob* out) {
  if (t_pack_pending.kind == 0) return false;
  *out = t_pack_pending;
  t_pack_pending.kind = 0;
  return true;
}
// the documented contract of a deferred pack: consumed on the thread AND stream it was submitted on
bool pack_stream_ok(hipStream_t st) { return t_pack_pending.kind == 0 || t_pack_stream == st; }
int pack_flush(hipStream_t st) {
  PCA_REQUIRE(pack_stream_ok(st), "deferred pack: submitted on another stream than the engine call's");
  PackJob j;
  return pack_take(&j) ? pack_launch(j, st) : PCA_OK;
}
}  // namespace pca

extern "C" {

int pca_pack_defer(int on) {
  // disarm FIRST: a pending job holds raw pointers to the caller's batch and must never ride a later
  // engine call of this thread, whatever this call returns
  const bool stale = on == 0 && pca::t_pack_pending.kind != 0;
  pca::t_pack_armed = on != 0;
  if (on == 0) pca::t_pack_pending.kind = 0;
  PCA_REQUIRE(!stale,
              "pack_defer(0): a deferred pack was never consumed (no engine call followed it); dropped");
  return PCA_OK;
}


int pca_resample(const float* x, int64_t n_in, double ratio, const double* win, const double* delta,
                 int nwin, int num_table, float gain, float* y, int64_t n_out, void* stream) {
  PCA_REQUIRE(x && win && delta && y, "resample: null pointer");
  PCA_REQUIRE(n_in > 0 && n_out > 0 && ratio > 0.0, "resample: n_in=%lld n_out=%lld ratio=%g",
              (long long)n_in, (long long)n_out, ratio);
  PCA_REQUIRE(nwin > 1 && num_table > 0 && (int)((ratio < 1.0 ? ratio : 1.0) * num_table) >= 1,
              "resample: filter table of %d entries, %d per zero crossing, ratio %g", nwin, num_table, ratio);
  // the last output sample must have an input sample under it
  PCA_REQUIRE((int64_t)((double)(n_out - 1) / ratio) < n_in, "resample: n_out=%lld beyond the input",
              (long long)n_out);
  hipLaunchKernelGGL(pca::k_resample, dim3((unsigned)pca::cdiv(n_out, 256)), dim3(256), 0,
                     pca::as_stream(stream), x, n_in, ratio, win, delta, nwin, num_table, gain, y, n_out);
  return pca::check_launch("k_resample");
}

int64_t pca_stft_num_frames(int64_t L, int hop) { return hop > 0 ? 1 + L / hop : 0; }

int pca_stft_logmag(const float* wave, int64_t L, int n_fft, int win_length, int hop,
                    int n_bins, float* out, int64_t stride_f, int64_t stride_t,
                    void* stream) {
  PCA_REQUIRE(wave && out, "stft_logmag: null pointer");
  PCA_REQUIRE(n_fft >= 64 && n_fft <= 4096 && (n_fft & (n_fft - 1)) == 0,
              "stft_logmag: n_fft=%d must be a power of two in [64, 4096]", n_fft);
  PCA_REQUIRE(win_length > 0 && win_length <= n_fft, "stft_logmag: win_length=%d", win_length);
  PCA_REQUIRE(hop > 0, "stft_logmag: hop=%d", hop);
  PCA_REQUIRE(n_bins > 0 && n_bins <= n_fft / 2 + 1, "stft_logmag: n_bins=%d", n_bins);
  PCA_REQUIRE(L > n_fft / 2, "stft_logmag: reflect padding needs L=%lld > n_fft/2",
              (long long)L);
  int log2n = 0;
  while ((1 << log2n) < n_fft) ++log2n;
  const int64_t T = pca_stft_num_frames(L, hop);
  const size_t lds = ((size_t)n_fft + n_fft / 2) * sizeof(double2);
  static std::once_flag lds_once;   // allow > 64 KiB of dynamic LDS (96 KiB at n_fft 4096)
  std::call_once(lds_once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pca::k_stft_logmag),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  });
  hipLaunchKernelGGL(pca::k_stft_logmag, dim3((unsigned)T), dim3(256), lds,
                     pca::as_stream(stream), wave, L, n_fft, log2n, win_length, hop, n_bins,
                     out, stride_f, stride_t, (const int64_t*)nullptr, (const int64_t*)nullptr);
  return pca::check_launch("k_stft_logmag");
}

int pca_stft_logmag_batch(const float* waves, const int64_t* wave_off, const int64_t* frame_off,
                          int n_clips, int64_t max_len, int64_t min_len, int n_fft,
                          int win_length, int hop, int n_bins, float* out, int64_t stride_f,
                          int64_t stride_t, void* stream) {
  PCA_REQUIRE(waves && wave_off && frame_off && out, "stft_logmag_batch: null pointer");
  PCA_REQUIRE(n_clips > 0 && n_clips <= 65535, "stft_logmag_batch: n_clips=%d", n_clips);
  PCA_REQUIRE(n_fft >= 64 && n_fft <= 4096 && (n_fft & (n_fft - 1)) == 0,
              "stft_logmag_batch: n_fft=%d must be a power of two in [64, 4096]", n_fft);
  PCA_REQUIRE(win_length > 0 && win_length <= n_fft, "stft_logmag_batch: win_length=%d",
              win_length);
  PCA_REQUIRE(hop > 0, "stft_logmag_batch: hop=%d", hop);
  PCA_REQUIRE(n_bins > 0 && n_bins <= n_fft / 2 + 1, "stft_logmag_batch: n_bins=%d", n_bins);
  PCA_REQUIRE(min_len > n_fft / 2 && max_len >= min_len,
              "stft_logmag_batch: reflect padding needs every clip longer than n_fft/2 "
              "(shortest %lld, longest %lld)", (long long)min_len, (long long)max_len);
  int log2n = 0;
  while ((1 << log2n) < n_fft) ++log2n;
  const int64_t T = pca_stft_num_frames(max_len, hop);
  const size_t lds = ((size_t)n_fft + n_fft / 2) * sizeof(double2);
  static std::once_flag lds_once;
  std::call_once(lds_once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pca::k_stft_logmag),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  });
  hipLaunchKernelGGL(pca::k_stft_logmag, dim3((unsigned)T, (unsigned)n_clips), dim3(256), lds,
                     pca::as_stream(stream), waves, (int64_t)0, n_fft, log2n, win_length, hop,
                     n_bins, out, stride_f, stride_t, wave_off, frame_off);
  return pca::check_launch("k_stft_logmag(batch)");
}

int pca_pack_points_2d(const float* spec, int64_t stride_f, int64_t stride_t,
                       const float* farr, const int64_t* idx, int B, int F, float* out,
                       const int64_t* labels, int64_t* labels_out, void* stream) {
  return pca_pack_points_2d_seq(spec, stride_f, stride_t, farr, idx, nullptr, nullptr, B, F, out,
                                labels, labels_out, stream);
}

int pca_pack_points_2d_seq(const float* spec, int64_t stride_f, int64_t stride_t,
                           const float* farr, const int64_t* idx_seq, const int32_t* step_dev,
                           const int32_t* base_dev, int B, int F, float* out,
                           const int64_t* labels, int64_t* labels_out, void* stream) {
  PCA_REQUIRE(spec && farr && idx_seq && out, "pack_points_2d: null pointer");
  PCA_REQUIRE((step_dev == nullptr) == (base_dev == nullptr), "pack_points_2d: cursor needs both");
  PCA_REQUIRE(B > 0 && F > 0 && B <= 65535, "pack_points_2d: B=%d F=%d", B, F);
  pca::PackJob j{};
  j.kind = 2; j.spec = spec; j.stride_f = stride_f; j.stride_t = stride_t; j.farr = farr;
  j.idx = idx_seq; j.step_dev = step_dev; j.base_dev = base_dev; j.labels = labels;
  j.labels_out = labels_out; j.out = out; j.B = B; j.F = F; j.Nt = 1;
  j.bx = (int)pca::cdiv(F, 256);
  // (only the cursor form is deferred: it is the one a captured training step issues)
  return step_dev != nullptr ? pca::pack_submit(j, pca::as_stream(stream))
                             : pca::pack_launch(j, pca::as_stream(stream));
}

int pca_pack_points_3d(const float* spec, int64_t stride_f, int64_t stride_t,
                       int64_t stride_s, const float* farr, const float* tarr,
                       const int64_t* idx, int B, int F, int Nt, float* out,
                       const int64_t* labels, int64_t* labels_out, void* stream) {
  PCA_REQUIRE(spec && farr && tarr && idx && out, "pack_points_3d: null pointer");
  PCA_REQUIRE(B > 0 && F > 0 && Nt > 0 && B <= 65535, "pack_points_3d: B=%d F=%d Nt=%d", B,
              F, Nt);
  pca::PackJob j{};
  j.kind = 3; j.spec = spec; j.stride_f = stride_f; j.stride_t = stride_t; j.stride_s = stride_s;
  j.farr = farr; j.tarr = tarr; j.idx = idx; j.step_dev = nullptr; j.base_dev = nullptr;
  j.nt_valid = nullptr; j.lengths_out = nullptr; j.labels = labels; j.labels_out = labels_out;
  j.out = out; j.B = B; j.F = F; j.Nt = Nt;
  j.bx = (int)pca::cdiv((int64_t)F * Nt, 256);
  return pca::pack_launch(j, pca::as_stream(stream));
}

int pca_pack_points_3d_seq(const float* spec, int64_t stride_f, int64_t stride_t,
                           int64_t stride_s, const float* farr, const float* tarr,
                           const int32_t* nt_valid, const int64_t* idx_seq,
                           const int32_t* step_dev, const int32_t* base_dev, int B, int F,
                           int Nt, float* out, int32_t* lengths_out, const int64_t* labels,
                           int64_t* labels_out, void* stream) {
  PCA_REQUIRE(spec && farr && tarr && idx_seq && out && step_dev && base_dev,
              "pack_points_3d_seq: null pointer");
  PCA_REQUIRE((nt_valid == nullptr) == (lengths_out == nullptr),
              "pack_points_3d_seq: nt_valid and lengths_out go together");
  PCA_REQUIRE(B > 0 && F > 0 && Nt > 0 && B <= 65535, "pack_points_3d_seq: B=%d F=%d Nt=%d", B,
              F, Nt);
  pca::PackJob j{};
  j.kind = 3; j.spec = spec; j.stride_f = stride_f; j.stride_t = stride_t; j.stride_s = stride_s;
  j.farr = farr; j.tarr = tarr; j.idx = idx_seq; j.step_dev = step_dev; j.base_dev = base_dev;
  j.nt_valid = nt_valid; j.lengths_out = lengths_out; j.labels = labels; j.labels_out = labels_out;
  j.out = out; j.B = B; j.F = F; j.Nt = Nt;
  j.bx = (int)pca::cdiv((int64_t)F * Nt, 256);
  return pca::pack_submit(j, pca::as_stream(stream));
}

int pca_pack_points_3d_var(const float* spec, int64_t stride_f, int64_t stride_t,
                           int64_t stride_s, const float* farr, const float* tarr,
                           const int32_t* nt_valid, const int64_t* idx, int B, int F, int Nt,
                           float* out, int32_t* lengths_out, const int64_t* labels,
                           int64_t* labels_out, void* stream) {
  PCA_REQUIRE(spec && farr && tarr && nt_valid && idx && out && lengths_out,
              "pack_points_3d_var: null pointer");
  PCA_REQUIRE(B > 0 && F > 0 && Nt > 0 && B <= 65535, "pack_points_3d_var: B=%d F=%d Nt=%d",
              B, F, Nt);
  pca::PackJob j{};
  j.kind = 3; j.spec = spec; j.stride_f = stride_f; j.stride_t = stride_t; j.stride_s = stride_s;
  j.farr = farr; j.tarr = tarr; j.idx = idx; j.step_dev = nullptr; j.base_dev = nullptr;
  j.nt_valid = nt_valid; j.lengths_out = lengths_out; j.labels = labels; j.labels_out = labels_out;
  j.out = out; j.B = B; j.F = F; j.Nt = Nt;
  j.bx = (int)pca::cdiv((int64_t)F * Nt, 256);
  return pca::pack_launch(j, pca::as_stream(stream));
}
}
