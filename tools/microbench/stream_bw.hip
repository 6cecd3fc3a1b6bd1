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

// The marching depthwise backward's access pattern with no arithmetic and few registers: a wave walks down a 64-lane column strip of one
// channel-quad plane (row pitch WP pixels of 16 bytes), loads one row of two tensors per step and stores one (block 1 of orcai-V1 at batch 64:
// planes 738 x 172, 3 strips of 62 columns, segments of 69 rows).  DEPTH rows requested ahead.
template <int DEPTH>
__global__ __launch_bounds__(256) void march_copy_kernel(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ d, int H, int W, int WP, int nstrip,
                                                          int nseg, int rps, int CQ, int sstep, int halo) {
  const int lane = threadIdx.x & 63;
  const int task = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (task >= nstrip * nseg) return;
  const int strip = task % nstrip, seg = task / nstrip;
  const int xcol = strip * sstep - halo + lane;
  const int plane = (H + 2) * WP;
  const size_t pbase = ((size_t)blockIdx.z * CQ + blockIdx.y) * plane;
  const bool out = lane >= halo && lane < halo + sstep && xcol < W;
  const int r0 = seg * rps, r1 = min(r0 + rps, H);
  auto pix = [&](int row) { int i = (row + 1) * WP + xcol; return i < 0 ? 0 : (i >= plane ? plane - 1 : i); };
  float4 pa[DEPTH], pb[DEPTH];
#pragma unroll
  for (int k = 0; k < DEPTH; ++k) { pa[k] = a[pbase + pix(r0 + k)]; pb[k] = b[pbase + pix(r0 + k)]; }
  for (int r = r0; r < r1; r += DEPTH) {
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) {
      const float4 x = pa[k], y = pb[k];
      pa[k] = a[pbase + pix(r + k + DEPTH)];
      pb[k] = b[pbase + pix(r + k + DEPTH)];
      if (out && r + k < r1) d[pbase + (size_t)(r + k + 1) * WP + xcol] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
    }
  }
}

// the whole row by one wave: NS consecutive 1-KiB pieces per tensor and step
template <int NS>
__global__ __launch_bounds__(256) void march_row_kernel(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ d, int H, int W, int WP, int nseg, int rps,
                                                         int CQ) {
  const int lane = threadIdx.x & 63;
  const int seg = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (seg >= nseg) return;
  const int plane = (H + 2) * WP;
  const size_t pbase = ((size_t)blockIdx.z * CQ + blockIdx.y) * plane;
  const int r0 = seg * rps, r1 = min(r0 + rps, H);
  auto pix = [&](int row, int s) { int i = (row + 1) * WP + s * 64 + lane; return i >= plane ? plane - 1 : i; };
  float4 pa[2][NS], pb[2][NS];
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int s = 0; s < NS; ++s) { pa[k][s] = a[pbase + pix(r0 + k, s)]; pb[k][s] = b[pbase + pix(r0 + k, s)]; }
  for (int r = r0; r < r1; r += 2) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const float4 x = pa[k][s], y = pb[k][s];
        pa[k][s] = a[pbase + pix(r + k + 2, s)];
        pb[k][s] = b[pbase + pix(r + k + 2, s)];
        if (s * 64 + lane < W && r + k < r1) d[pbase + (size_t)(r + k + 1) * WP + s * 64 + lane] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
      }
    }
  }
}

void run_row(float4* buf[5], int nseg) {
  const int B = 64, CQ = 8, H = 736, W = 171, WP = 176;
  const int rps = (H + nseg - 1) / nseg;
  dim3 grid((nseg + 3) / 4, CQ, B);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((march_row_kernel<3>), grid, dim3(256), 0, 0, buf[0], buf[1], buf[3], H, W, WP, nseg, rps, CQ);
  (void)hipEventRecord(e0);
  const int reps = 10;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((march_row_kernel<3>), grid, dim3(256), 0, 0, buf[0], buf[1], buf[3], H, W, WP, nseg, rps, CQ);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  const double bytes = 3.0 * B * 30 * H * W * 4;
  printf("whole-row marching copy (3 pieces per step and wave), %d segments of %d rows: %.3f ms  %.2f TB/s of algorithmic bytes\n", nseg, rps, ms, bytes / ms * 1e-9);
}

template <int DEPTH>
void run_march(float4* buf[5], int WP = 172, int sstep = 62, int halo = 1, int nseg = 11) {
  const int B = 64, CQ = 8, H = 736, W = 171;
  const int rps = (H + nseg - 1) / nseg;
  const int nstrip = (W + sstep - 1) / sstep;
  dim3 grid((nstrip * nseg + 3) / 4, CQ, B);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((march_copy_kernel<DEPTH>), grid, dim3(256), 0, 0, buf[0], buf[1], buf[3], H, W, WP, nstrip, nseg, rps, CQ, sstep, halo);
  (void)hipEventRecord(e0);
  const int reps = 10;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((march_copy_kernel<DEPTH>), grid, dim3(256), 0, 0, buf[0], buf[1], buf[3], H, W, WP, nstrip, nseg, rps, CQ, sstep, halo);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  const double bytes = 3.0 * B * 30 * H * W * 4;  // algorithmic: 30 channels, two tensors read, one written
  printf("marching copy, row pitch %d px, strips of %d columns + %d halo, %d strips, %d segments, %d rows ahead: %.3f ms  %.2f TB/s of algorithmic bytes\n", WP, sstep, halo,
         nstrip, nseg, DEPTH, ms, bytes / ms * 1e-9);
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
    (void)hipMemset(buf[i], 0, n4 * 16);
  }
  for (int blocks : {8192}) {
    run<1, 0>("1 read", buf, n4, blocks);
    run<1, 1>("1 read + 1 write (copy)", buf, n4, blocks);
    run<2, 1>("2 reads + 1 write", buf, n4, blocks);
    run<3, 1>("3 reads + 1 write", buf, n4, blocks);
    run<1, 2>("1 read + 2 writes", buf, n4, blocks);
  }
  run_march<3>(buf);
  run_march<3>(buf, 176, 64, 0);
  run_march<3>(buf, 176, 64, 0, 4);
  run_march<3>(buf, 176, 64, 0, 31);
  run_march<3>(buf, 176, 64, 0, 92);
  run_row(buf, 11);
  run_row(buf, 31);
  run_row(buf, 92);
  return 0;
}
