// capi.hip -- library identification and shared helpers of the C ABI (include/orcai_hip.h).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

#include "orcai_hip.h"
#include "zero_fill.h"

extern "C" const char* orcai_version(void) { return "orcai_hip 0.1.0 gfx950"; }

// ---- the accumulator arena: a training step clears ALL of its reduction scratch (BatchNorm sums, weight-gradient shards, ...) with one launch.
// The caller hands out 32-KiB slots of one buffer to the launchers that accumulate; a launcher's own zero fill (orcai_zero::zero_async) is skipped
// when its range lies inside a slot of the registered arena that nobody has taken since the arena was cleared -- and happens as before for any
// other pointer, for a slot used a second time, and when no arena is registered: correctness never depends on the caller's bookkeeping.
namespace {
constexpr size_t kSlot = 32768;
constexpr int kMaxSlots = 1024;
char* g_arena = nullptr;
size_t g_arena_slots = 0;
uint64_t g_arena_used[kMaxSlots / 64];
}  // namespace

extern "C" int orcai_arena_take(const void* p, size_t bytes) {
  const char* q = static_cast<const char*>(p);
  if (!g_arena || q < g_arena || bytes == 0) return 0;
  const size_t off = (size_t)(q - g_arena), slot = off / kSlot;
  if (slot >= g_arena_slots || (off + bytes - 1) / kSlot != slot) return 0;
  if (g_arena_used[slot >> 6] >> (slot & 63) & 1) return 0;  // taken before: its contents are somebody's sums by now
  g_arena_used[slot >> 6] |= 1ull << (slot & 63);
  return 1;
}

extern "C" int orcai_scratch_arena(void* base, size_t bytes, void* stream) {
  if (!base || bytes < kSlot) {  // unregister
    g_arena = nullptr;
    g_arena_slots = 0;
    return 0;
  }
  if ((uintptr_t)base & 15) return ORCAI_E_BADARG;
  size_t slots = bytes / kSlot;
  if (slots > kMaxSlots) slots = kMaxSlots;
  g_arena = nullptr;  // (the fill below must not be skipped)
  hipError_t e = orcai_zero::zero_async(base, slots * kSlot, (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  g_arena = static_cast<char*>(base);
  g_arena_slots = slots;
  std::memset(g_arena_used, 0, sizeof(g_arena_used));
  return 0;
}

// ---- measurement hook: HIP events around the MAIN kernel of the next fused weight-gradient launcher call (orcai_bn_bwd_pointwise_wgrad /
// orcai_h_bn_bwd_pointwise_wgrad launch two small kernels after it; a bracket around the whole call, which is all a caller can place, reads ~35 us
// above the kernel rocprofv3 lists).  bench.py registers a pair per call; the launcher records them on its stream and the registration is consumed.
namespace {
hipEvent_t g_prof_e0 = nullptr, g_prof_e1 = nullptr;
}
extern "C" int orcai_profile_bracket(void* ev_start, void* ev_stop) {
  g_prof_e0 = (hipEvent_t)ev_start;
  g_prof_e1 = (hipEvent_t)ev_stop;
  return 0;
}
extern "C" void orcai_profile_take(void** ev_start, void** ev_stop) {  // library-internal (zero_fill.h declares it)
  *ev_start = g_prof_e0;
  *ev_stop = g_prof_e1;
  g_prof_e0 = g_prof_e1 = nullptr;
}
// the events themselves, for callers without a HIP binding of their own (ctypes): created with timing enabled
extern "C" int orcai_event_create(void** ev) {
  if (!ev) return ORCAI_E_BADARG;
  hipEvent_t e = nullptr;
  const hipError_t err = hipEventCreate(&e);
  *ev = e;
  return (int)err;
}
extern "C" int orcai_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms) {  // both events must have completed (synchronise first)
  if (!ev_start || !ev_stop || !ms) return ORCAI_E_BADARG;
  return (int)hipEventElapsedTime(ms, (hipEvent_t)ev_start, (hipEvent_t)ev_stop);
}
extern "C" int orcai_event_destroy(void* ev) { return ev ? (int)hipEventDestroy((hipEvent_t)ev) : 0; }
