// lds_dma.h -- LDS-DMA helpers shared by the f32 and f16 LDS-shared-row kernels (model_fwd.hip, half_fwd.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace orcai_lds {

// global_load_lds_dwordx4: 64 lanes x 16 bytes from per-lane global addresses to lds_dst + 16 * lane (wave-uniform base in M0), no VGPR
// destination.  Inline asm so that the compiler's wait bookkeeping neither sees nor drains it (a builtin-issued LDS-DMA makes hipcc
// wait vmcnt(0) in front of every LDS read that may alias it); completion is counted by the caller with wait_vm_barrier.
__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// all but the N youngest vector-memory operations of this wave are done, then the workgroup barrier: every wave's DMAs of the current
// step have landed when the barrier is passed
template <int N>
__device__ __forceinline__ void wait_vm_barrier() {
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

}  // namespace orcai_lds
