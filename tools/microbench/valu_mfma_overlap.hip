// Do f32 VALU work and f32-input MFMA work overlap on one SIMD of gfx950?
// Four kernels over the same number of loop iterations, one workgroup per CU:
//   valu : 32 independent v_pk_fma_f32 per iteration (128 issue cycles)
//   mfma : 4 independent v_mfma_f32_16x16x4_f32 per iteration (128 pipe cycles)
//   both : the two bodies interleaved in ONE wave
//   split: waves 0-3 run the VALU body, waves 4-7 the MFMA body (512 threads: every SIMD holds one wave of each kind)
// If the two units run concurrently, `split` (and ideally `both`) take about max(valu, mfma); if they share the datapath, the sum.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_mfma_overlap valu_mfma_overlap.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, bool HALF, bool PACKED /*VALU body: 32 v_pk_fma_f32 or 32 v_fma_f32 (4 issue cycles each either way)*/ /*MFMA body: 8 x v_mfma_f32_16x16x32_f16 (16 cycles each) instead of 4 x v_mfma_f32_16x16x4_f32*/>  // 0 valu, 1 mfma, 2 both in one wave, 3 split by wave (wave >> 2: consecutive waves of a workgroup go to consecutive SIMDs)
__global__ __launch_bounds__(1024) void body(float* out, int iters, float seed) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: the role branches are scalar
  f32x2 v[16];
  f32x4 c[4];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = (f32x2){seed + i, seed - i};
#pragma unroll
  for (int i = 0; i < 4; ++i) c[i] = (f32x4){seed, seed, seed, seed};
  const f32x2 m = {1.0001f, 0.9999f}, a = {1e-6f, -1e-6f};
  const float fa = seed * 0.5f, fb = seed * 0.25f;
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  const h8 ha = {(_Float16)fa, (_Float16)fb, 1, 2, 3, 4, 5, 6}, hb = {(_Float16)fb, 1, 1, 2, 2, 3, 3, 4};
  const bool do_valu = MODE == 0 || MODE == 2 || (MODE == 3 && ((wave >> 2) & 1) == 0);
  const bool do_mfma = MODE == 1 || MODE == 2 || (MODE == 3 && ((wave >> 2) & 1) == 1);
  for (int it = 0; it < iters; ++it) {
    if (MODE == 4) {  // one wave, instruction-level interleave: each MFMA followed by its share of the VALU body
#pragma unroll
      for (int g = 0; g < (HALF ? 8 : 4); ++g) {
        if (HALF) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c[g & 3]) : "v"(ha), "v"(hb));
        else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c[g]) : "v"(fa), "v"(fb));
#pragma unroll
        for (int i = 0; i < (HALF ? 4 : 8); ++i) {
          const int k = (g * (HALF ? 4 : 8) + i) & 15;
          if (PACKED) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(m), "v"(a));
          else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k].x) : "v"(m.x), "v"(a.x));
        }
      }
      continue;
    }
    if (do_valu) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          if (PACKED) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(m), "v"(a));
          else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i].x) : "v"(m.x), "v"(a.x));
        }
    }
    if (do_mfma) {
      if (HALF) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int i = 0; i < 4; ++i) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c[i]) : "v"(ha), "v"(hb));
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c[i]) : "v"(fa), "v"(fb));
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += v[i].x + v[i].y;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, bool HALF, bool PACKED>
float run(float* out, int threads, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((body<MODE, HALF, PACKED>), dim3(256), dim3(threads), 0, 0, out, iters, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((body<MODE, HALF, PACKED>), dim3(256), dim3(threads), 0, 0, out, iters, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 512 * sizeof(float));
  const int iters = 20000;
  // 128 issue cycles of each kind per iteration and wave
  for (int threads : {256, 512, 1024}) {  // 1, 2 or 4 waves per SIMD
    const double cyc = 1e-3 / iters * 2.4e9;  // cycles per iteration at a nominal 2.4 GHz
    auto line = [&](const char* what, float t0, float t1, float t2, float t3, float t4) {
      printf("%s  waves/SIMD=%d  valu %.3f ms (%.0f cyc/iter)  mfma %.3f ms (%.0f)  both-in-one-wave %.3f ms (%.0f)  split-by-wave %.3f ms (%.0f)  interleaved-in-one-wave %.3f ms (%.0f)\n", what, threads / 256, t0,
             t0 * cyc, t1, t1 * cyc, t2, t2 * cyc, t3, t3 * cyc, t4, t4 * cyc);
    };
    line("v_pk_fma_f32 + f32 MFMA", run<0, false, true>(out, threads, iters), run<1, false, true>(out, threads, iters), run<2, false, true>(out, threads, iters), run<3, false, true>(out, threads, iters), run<4, false, true>(out, threads, iters));
    line("v_fma_f32    + f32 MFMA", run<0, false, false>(out, threads, iters), run<1, false, false>(out, threads, iters), run<2, false, false>(out, threads, iters), run<3, false, false>(out, threads, iters), run<4, false, false>(out, threads, iters));
    line("v_pk_fma_f32 + f16 MFMA", run<0, true, true>(out, threads, iters), run<1, true, true>(out, threads, iters), run<2, true, true>(out, threads, iters), run<3, true, true>(out, threads, iters), run<4, true, true>(out, threads, iters));
    line("v_fma_f32    + f16 MFMA", run<0, true, false>(out, threads, iters), run<1, true, false>(out, threads, iters), run<2, true, false>(out, threads, iters), run<3, true, false>(out, threads, iters), run<4, true, false>(out, threads, iters));
  }
  // reading: with W waves per SIMD, valu and mfma each cost W * 128 cycles per iteration; `both` costs W * 256 if nothing overlaps and W * 128 if
  // everything does; `split` (W = 2: one wave of each kind) costs 256 if the units share the datapath and 128 if they run concurrently.
  return 0;
}
