// graph_memset.hip -- does a hipMemsetAsync node replay with the pattern it was captured with?  (ROCm 7.2, gfx950)
// Round 2 saw an accumulator that a captured hipMemsetAsync(p, 0, bytes) clears come back "1.0 too low in every odd double" on the second
// and later replays of a three-node graph (memset, conv kernel with f64 atomics, finish kernel).  This probe takes the library and torch out:
// graph = { hipMemsetAsync(acc, 0, bytes) ; add_kernel(acc) ; snapshot(acc -> snap[r]) } replayed R times, with eager work between the
// replays (other memsets with other patterns, kernels with double arguments) that could recycle whatever staging the runtime keeps for a
// memset node.  Prints every replay whose snapshot differs from the expected sums, with the raw bit patterns.
// build: hipcc -O2 --offload-arch=gfx950 tools/microbench/graph_memset.hip -o tools/bin/graph_memset
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)

__global__ void add_kernel(double* acc, int n) {  // every accumulator receives 256 atomic adds of (i % 7 + 1)
  const int i = blockIdx.x % n;
  atomicAdd(&acc[i], (double)(i % 7 + 1));
}
__global__ void fill_doubles(double* p, int n, double v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
__global__ void snapshot(const double* acc, double* snap, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) snap[i] = acc[i];
}

int main(int argc, char** argv) {
  const int R = 6;
  // argv[1] = "destroy": hipGraphDestroy(graph) right after hipGraphInstantiate -- what torch.cuda.CUDAGraph.capture_end does -- followed by
  // host-heap churn with a recognisable pattern: does the executable graph's memset node still point into the destroyed graph?
  // argv[1]: any of the letters d (destroy the source graph after instantiate), a (instantiate with AutoFreeOnLaunch), n (replay and eager work on the
  // NULL stream) -- "dan" is what torch.cuda.CUDAGraph does when replayed on the default stream
  const char* flags = argc > 1 ? argv[1] : "";
  const bool destroy_early = strchr(flags, 'd'), autofree = strchr(flags, 'a'), null_stream = strchr(flags, 'n');
  int rtv = 0;
  (void)hipRuntimeGetVersion(&rtv);
  printf("HIP runtime version %d, flags \"%s\": destroy source graph %d, AutoFreeOnLaunch %d, null stream %d\n", rtv, flags, (int)destroy_early, (int)autofree, (int)null_stream);
  int bad_total = 0;
  for (int n : {3 * 8 * 32, 8 * 8 * 32, 15 * 8 * 32, 4096, 1000}) {
    for (int misalign = 0; misalign < 2; ++misalign) {
      double *base, *acc, *snap, *other;
      CK(hipMalloc(&base, (n + 2) * sizeof(double)));
      acc = base + misalign;  // 8-byte but not 16-byte aligned when misalign = 1
      CK(hipMalloc(&snap, (size_t)R * n * sizeof(double)));
      CK(hipMalloc(&other, 1 << 20));
      hipStream_t st;
      CK(hipStreamCreate(&st));
      hipGraph_t g;
      hipGraphExec_t ge;
      double* snap_arg = snap;  // the snapshot kernel of replay r writes snap + r * n: updated through hipGraphExecKernelNodeSetParams? no: one slot, copied out per replay
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
      CK(hipMemsetAsync(acc, 0, (size_t)n * sizeof(double), st));
      hipLaunchKernelGGL(add_kernel, dim3(256 * n), dim3(1), 0, st, acc, n);
      hipLaunchKernelGGL(snapshot, dim3((n + 255) / 256), dim3(256), 0, st, acc, snap_arg, n);
      CK(hipStreamEndCapture(st, &g));
      if (autofree) CK(hipGraphInstantiateWithFlags(&ge, g, hipGraphInstantiateFlagAutoFreeOnLaunch)); else CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      hipStream_t rs = null_stream ? (hipStream_t)0 : st;  // replay stream
      std::vector<std::vector<double>> churn;
      if (destroy_early) {
        CK(hipGraphDestroy(g));
        for (int k = 0; k < 4096; ++k) churn.emplace_back(8 + (k % 61), -1.0);  // freed graph nodes are now likely to sit under vectors of -1.0
      }
      std::vector<double> host(n);
      for (int r = 0; r < R; ++r) {
        // eager work between the replays on the same stream: byte memsets with other values, a D32 memset, a kernel with a double argument of -1.0
        CK(hipMemsetAsync(other, 0x5A, 1 << 20, rs));
        CK(hipMemsetD32Async((hipDeviceptr_t)other, 0xBFF00000u, (1 << 20) / 4, rs));
        for (int k = 0; k < 32; ++k) hipLaunchKernelGGL(fill_doubles, dim3(512), dim3(256), 0, rs, other, 1 << 17, -1.0);
        CK(hipGraphLaunch(ge, rs));
        CK(hipStreamSynchronize(rs));
        CK(hipMemcpy(host.data(), snap, n * sizeof(double), hipMemcpyDeviceToHost));
        int bad = 0;
        for (int i = 0; i < n; ++i) {
          const double want = 256.0 * (i % 7 + 1);
          if (host[i] != want) {
            if (bad < 4) {
              double d = host[i] - want;
              uint64_t bits;
              memcpy(&bits, &d, 8);
              printf("  n %d misalign %d replay %d acc[%d] = %.17g want %.17g (diff %.17g, bits %016llx)\n", n, misalign, r, i, host[i], want, d, (unsigned long long)bits);
            }
            ++bad;
          }
        }
        if (bad) printf("n %d misalign %d replay %d: %d of %d accumulators wrong\n", n, misalign, r, bad, n);
        bad_total += bad;
      }
      CK(hipGraphExecDestroy(ge));
      if (!destroy_early) CK(hipGraphDestroy(g));
      CK(hipStreamDestroy(st));
      CK(hipFree(base));
      CK(hipFree(snap));
      CK(hipFree(other));
    }
  }
  printf("graph_memset: %s (%d wrong accumulators over all replays)\n", bad_total ? "MEMSET NODE REPLAYS WRONG" : "all replays exact", bad_total);
  return bad_total ? 1 : 0;
}
