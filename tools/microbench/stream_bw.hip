// Streaming-rate microbenchmark for the access mixes of the training kernels: R reads + W writes of separate 1 GiB tensors, 16 bytes per lane,
// block-contiguous (a block owns a contiguous 16 KiB piece per tensor and iteration).  What can a kernel that does nothing else move on this part
// with 2 reads + 1 write (the marching depthwise backward), 3 reads + 1 write (BatchNorm backward + pointwise weight gradient), 1 + 1 (a copy)?
// build: hipcc -O3 --offload-arch=gfx950 stream_bw.hip -o stream_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int R, int W>
__global__ __launch_bounds__(256) void stream_kernel(const float4* __restrict__ a, const float4* __restrict__ b, const float4* __restrict__ c, float4* __restrict__ d,
                                                      float4* __restrict__ e, size_t n4) {
  const size_t per_block = 1024;  // float4 per block and iteration: 4 per thread
  for (size_t base = (size_t)blockIdx.x * per_block; base < n4; base += (size_t)gridDim.x * per_block) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const size_t i = base + k * 256 + threadIdx.x;
      if (i < n4) {
        float4 v = a[i];
        if (R > 1) { const float4 u = b[i]; v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
        if (R > 2) { const float4 u = c[i]; v.x *= u.x; v.y *= u.y; v.z *= u.z; v.w *= u.w; }
        if (W > 0) d[i] = v;
        if (W > 1) e[i] = v;
        if (W == 0 && v.x == 1.2345e-30f) d[i] = v;
      }
    }
  }
}

template <int R, int W>
void run(const char* name, float4* buf[5], size_t n4, int blocks) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((stream_kernel<R, W>), dim3(blocks), dim3(256), 0, 0, buf[0], buf[1], buf[2], buf[3], buf[4], n4);
  hipEventRecord(e0);
  const int reps = 10;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((stream_kernel<R, W>), dim3(blocks), dim3(256), 0, 0, buf[0], buf[1], buf[2], buf[3], buf[4], n4);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  const double bytes = (double)(R + W) * n4 * 16;
  printf("%-28s blocks %6d  %.3f ms  %.2f TB/s\n", name, blocks, ms, bytes / ms * 1e-9);
}

int main() {
  const size_t n4 = (size_t)1 << 26;  // 1 GiB per tensor
  float4* buf[5];
  for (int i = 0; i < 5; ++i) {
    if (hipMalloc(&buf[i], n4 * 16) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf[i], 0, n4 * 16);
  }
  for (int blocks : {2048, 8192, 65536}) {
    run<1, 0>("1 read", buf, n4, blocks);
    run<1, 1>("1 read + 1 write (copy)", buf, n4, blocks);
    run<2, 1>("2 reads + 1 write", buf, n4, blocks);
    run<3, 1>("3 reads + 1 write", buf, n4, blocks);
    run<1, 2>("1 read + 2 writes", buf, n4, blocks);
  }
  return 0;
}
