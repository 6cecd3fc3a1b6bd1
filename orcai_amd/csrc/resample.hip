// resample.hip -- rational-ratio polyphase resampler for gfx950.
//
// Replaces the resampling half of librosa.load(sr=...) (reference src/orcAI/spectrogram.py:23-27, which uses
// libsoxr "soxr_hq").  soxr is not available in this image, so bit parity with it is impossible ("parity
// unpinned"); this is a Kaiser-windowed-sinc polyphase filter (64 zero crossings, beta 14.77,
// roll-off 0.9475) whose table is designed on the host in float64 (orcai_amd/resample.py).
//   out[n] = sum_j x[i0 - half + 1 + j] * table[phase][j],   i0 = floor(n*M/L), phase = (n*M) mod L
#include <hip/hip_runtime.h>

#include <cstdint>

#include "orcai_hip.h"

namespace {

__global__ __launch_bounds__(256) void resample_kernel(const float* __restrict__ x, int64_t n_in, float* __restrict__ out, int64_t n_out, int L, int M,
                                                        const float* __restrict__ table /*[L][ntaps]*/, int ntaps) {
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= n_out) return;
  const int64_t num = n * (int64_t)M;
  const int64_t i0 = num / L;
  const int ph = (int)(num - i0 * L);
  const float* h = table + (int64_t)ph * ntaps;
  const int64_t k0 = i0 - ntaps / 2 + 1;
  float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;  // 4 partial sums: shorter dependency chains, fixed order
  if (k0 >= 0 && k0 + ntaps <= n_in) {
    const float* xp = x + k0;
    for (int j = 0; j < ntaps; j += 4) {
      acc0 = fmaf(xp[j], h[j], acc0);
      acc1 = fmaf(xp[j + 1], h[j + 1], acc1);
      acc2 = fmaf(xp[j + 2], h[j + 2], acc2);
      acc3 = fmaf(xp[j + 3], h[j + 3], acc3);
    }
  } else {
    for (int j = 0; j < ntaps; j += 4) {
      const int64_t k = k0 + j;
      acc0 = fmaf((k >= 0 && k < n_in) ? x[k] : 0.f, h[j], acc0);
      acc1 = fmaf((k + 1 >= 0 && k + 1 < n_in) ? x[k + 1] : 0.f, h[j + 1], acc1);
      acc2 = fmaf((k + 2 >= 0 && k + 2 < n_in) ? x[k + 2] : 0.f, h[j + 2], acc2);
      acc3 = fmaf((k + 3 >= 0 && k + 3 < n_in) ? x[k + 3] : 0.f, h[j + 3], acc3);
    }
  }
  out[n] = (acc0 + acc1) + (acc2 + acc3);
}

}  // namespace

extern "C" int orcai_resample_polyphase(const float* x, int64_t n_in, float* out, int64_t n_out, int L, int M, const float* table, int ntaps,
                                        void* stream) {
  if (!x || !out || !table || n_in <= 0 || n_out <= 0 || L <= 0 || M <= 0 || ntaps <= 0 || (ntaps & 3)) return ORCAI_E_BADARG;
  hipLaunchKernelGGL(resample_kernel, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, n_in, out, n_out, L, M, table, ntaps);
  return (int)hipGetLastError();
}
