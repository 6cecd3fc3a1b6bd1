// What does a read-only pass over 1 GB reach on this part, and what costs it bandwidth?  Variants of a sum reduction over float4:
//   loads in flight per thread (1, 2, 4, 8), f32 or f64 accumulation, grid size (blocks per CU), block-contiguous or grid-strided.
// build: hipcc --offload-arch=gfx950 -O3 -o read_bw read_bw.hip
#include <hip/hip_runtime.h>

#include <cstdio>

template <int U, bool F64, bool CONTIG>
__global__ __launch_bounds__(256) void reduce(const float4* __restrict__ x, long n, double* out) {
  // CONTIG: a block owns n / gridDim.x consecutive elements; otherwise grid-strided (consecutive blocks read consecutive 4 KiB)
  const long per = (n + gridDim.x - 1) / gridDim.x;
  long i = CONTIG ? blockIdx.x * per + threadIdx.x : (long)blockIdx.x * 256 + threadIdx.x;
  const long end = CONTIG ? (blockIdx.x * per + per < n ? blockIdx.x * per + per : n) : n;
  const long step = CONTIG ? 256 : (long)gridDim.x * 256;
  double d[4] = {0, 0, 0, 0};
  float f[4] = {0, 0, 0, 0};
  for (; i + (U - 1) * step < end; i += U * step) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = x[i + u * step];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (F64) {
        d[0] += v[u].x; d[1] += v[u].y; d[2] += v[u].z; d[3] += v[u].w;
      } else {
        f[0] += v[u].x; f[1] += v[u].y; f[2] += v[u].z; f[3] += v[u].w;
      }
    }
  }
  for (; i < end; i += step) {
    const float4 v = x[i];
    d[0] += v.x; d[1] += v.y; d[2] += v.z; d[3] += v.w;
  }
  const double s = d[0] + d[1] + d[2] + d[3] + f[0] + f[1] + f[2] + f[3];
  if (s == 12345.678) out[0] = s;  // keep the loads
}

__global__ void fill_random(float4* x, long n) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  unsigned h = (unsigned)(i * 2654435761u) ^ (unsigned)(i >> 17);
  float v[4];
  for (int k = 0; k < 4; ++k) {
    h = h * 1664525u + 1013904223u;
    v[k] = (float)(h >> 8) * (1.0f / 8388608.0f) - 1.0f;
  }
  x[i] = make_float4(v[0], v[1], v[2], v[3]);
}

template <int U, bool F64, bool CONTIG>
void run(const float4* x, long n, double* out, int blocks, const char* what) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((reduce<U, F64, CONTIG>), dim3(blocks), dim3(256), 0, 0, x, n, out);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((reduce<U, F64, CONTIG>), dim3(blocks), dim3(256), 0, 0, x, n, out);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  printf("%-34s blocks %6d  loads in flight %d  %s  %.3f ms  %.2f TB/s\n", what, blocks, U, F64 ? "f64" : "f32", ms, n * 16.0 / (ms * 1e-3) / 1e12);
}

int main() {
  const long n = (1l << 30) / 16;  // 1 GiB of float4
  float4* x;
  double* out;
  (void)hipMalloc(&x, n * 16);
  (void)hipMalloc(&out, 8);
  (void)hipMemset(x, 0, n * 16);
  for (int pass = 0; pass < 2; ++pass) {
  printf("---- %s data\n", pass ? "random" : "zero");
  if (pass) {
    hipLaunchKernelGGL(fill_random, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, x, n);
    (void)hipDeviceSynchronize();
  }
  for (int blocks : {1024, 4096}) {
    run<1, false, false>(x, n, out, blocks, "grid-strided");
    run<1, true, false>(x, n, out, blocks, "grid-strided");
    run<4, false, false>(x, n, out, blocks, "grid-strided");
    run<4, true, false>(x, n, out, blocks, "grid-strided");
    run<8, false, false>(x, n, out, blocks, "grid-strided");
    run<4, false, true>(x, n, out, blocks, "block-contiguous");
    run<4, true, true>(x, n, out, blocks, "block-contiguous");
  }
  run<4, true, true>(x, n, out, 65536, "block-contiguous");
  run<1, true, true>(x, n, out, 65536, "block-contiguous");
  }
  return 0;
}
