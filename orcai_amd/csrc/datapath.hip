// datapath.hip -- training-data path on the GPU for gfx950.
//
// The reference materialises every snippet into a tf.data snapshot (io.py:187-218: 240 000 x (736 x 171 f32) = 120 GB) that
// DataLoader.__getitem__ (io.py:128-147) fills by slicing rows [row_start, row_stop) of a recording's spectrogram / label
// arrays.  Here the per-recording arrays stay resident in HBM, concatenated along time ([sum T][171] f32 and [sum T][L]
// f32), and a batch is B row offsets:
//   gather_snippets:    out[b][r][c] = store[(start_b + r)][c]           -- B contiguous blocks, 16-byte copies
//   downsample_labels:  out[b][s][l] = rint(mean_{k<f} labels[start_b + s*f + k][l])   (DataLoader.reshape_labels, io.py:101-126:
//                       tf.reduce_mean over groups of 2**n_filters rows, tf.round = half to even; a group of -1 stays -1)
#include <hip/hip_runtime.h>

#include <cstdint>

#include "orcai_hip.h"

namespace {

__global__ __launch_bounds__(256) void gather_snippets_kernel(const float* __restrict__ store, const int64_t* __restrict__ row_starts, int64_t n_per_snippet,
                                                               int cols, float* __restrict__ out) {
  const int b = blockIdx.y;
  const float* src = store + row_starts[b] * (int64_t)cols;
  float* dst = out + (int64_t)b * n_per_snippet;
  // the source block starts at an arbitrary row: peel to a 16-byte boundary of the SOURCE only if source and destination agree
  const int64_t n4 = n_per_snippet >> 2;
  const bool vec = (((uintptr_t)src | (uintptr_t)dst) & 15) == 0;
  if (vec) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256)
      reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(src)[i];
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_per_snippet; i += (int64_t)gridDim.x * 256) dst[i] = src[i];
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_per_snippet; i += (int64_t)gridDim.x * 256) dst[i] = src[i];
  }
}

__global__ __launch_bounds__(256) void downsample_labels_kernel(const float* __restrict__ labels, const int64_t* __restrict__ row_starts, int B, int steps, int L,
                                                                 int factor, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)B * steps * L) return;
  const int l = (int)(i % L);
  const int s = (int)((i / L) % steps);
  const int b = (int)(i / ((int64_t)L * steps));
  const float* src = labels + (row_starts[b] + (int64_t)s * factor) * L + l;
  float sum = 0.0f;
  for (int k = 0; k < factor; ++k) sum += src[(int64_t)k * L];
  out[i] = rintf(sum / (float)factor);  // round half to even, like tf.round
}

}  // namespace

extern "C" {

int orcai_gather_snippets(const float* store, const int64_t* row_starts, int B, int rows, int cols, float* out, void* stream) {
  if (!store || !row_starts || !out || B <= 0 || rows <= 0 || cols <= 0) return ORCAI_E_BADARG;
  const int64_t n = (int64_t)rows * cols;
  int gx = (int)((n / 4 + 255) / 256);
  if (gx > 64) gx = 64;
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(gather_snippets_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, store, row_starts, n, cols, out);
  return (int)hipGetLastError();
}

int orcai_downsample_labels(const float* labels, const int64_t* row_starts, int B, int rows, int L, int factor, float* out, void* stream) {
  if (!labels || !row_starts || !out || B <= 0 || rows <= 0 || L <= 0 || factor <= 0) return ORCAI_E_BADARG;
  if (rows % factor) return ORCAI_E_BADARG;  // io.py:123-126 raises ValueError
  const int steps = rows / factor;
  const int64_t n = (int64_t)B * steps * L;
  hipLaunchKernelGGL(downsample_labels_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, labels, row_starts, B, steps, L, factor, out);
  return (int)hipGetLastError();
}

}  // extern "C"
