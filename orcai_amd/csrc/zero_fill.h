// zero_fill.h -- stream-ordered zero fill as a KERNEL node.
// hipMemsetAsync captured into a hipGraph in front of a kernel that accumulates with atomics came back with a corrupted fill pattern on the
// second and later replays of a captured training step on ROCm 7.2 (every odd double -1.0 instead of 0; tools/debug_graph_stats.py), while
// the eager launch order was right.  Every accumulator of this library is therefore cleared by a kernel.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

extern "C" void orcai_profile_take(void** ev_start, void** ev_stop);  // capi.hip: the event pair registered by orcai_profile_bracket (consumed)
extern "C" int orcai_arena_take(const void* p, size_t bytes);  // capi.hip: 1 = [p, p + bytes) is a fresh slot of the step's pre-cleared accumulator arena

namespace orcai_zero {

__global__ __launch_bounds__(256) static void zero_words_kernel(uint32_t* __restrict__ p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0u;
}

// zero `bytes` (a multiple of 4) starting at the 4-byte aligned p
inline hipError_t zero_async(void* p, size_t bytes, hipStream_t st) {
  const size_t n = bytes / 4;
  if (n == 0) return hipSuccess;
  if (orcai_arena_take(p, bytes)) return hipSuccess;  // cleared with the whole arena at the start of the step (orcai_scratch_arena)
  const size_t want = (n + 255) / 256;
  hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)(want < 4096 ? want : 4096)), dim3(256), 0, st, static_cast<uint32_t*>(p), n);
  return hipGetLastError();
}

}  // namespace orcai_zero
