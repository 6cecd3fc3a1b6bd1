// model_fwd.hip -- inference forward of orcAI's ResNetLSTM (architectures.py:162-241) for gfx950.
//
// Activation layout in HBM: planar fp32 [snippet][channel][HP][WP], zero-padded planes (see "Padded plane layout").
// The pointwise (1x1) convolutions, the LSTM projections/recurrence and the dense layers are f32-input
// MFMA contractions (v_mfma_f32_16x16x4_f32: exact f32, same rounding as an fmaf chain), with the
// OUTPUT-CHANNEL index on the MFMA row and the PIXEL (or batch) index on the MFMA column/lane, so a
// D fragment register holds 16 consecutive pixels of one channel -> planar stores need no transpose.
//
// Keras semantics restated (no Keras in this image; oracle/model_ref.py is the checker):
//   Conv2D/SeparableConv2D padding="same" stride 1 -> symmetric zero pad (k-1)/2
//   MaxPooling2D((3,2), strides 2, "same")       -> out = ceil(in/2), -inf pad, pad_before = total/2
//   Conv2D(1x1, strides 2, "same")               -> no pad, samples rows/cols 0,2,4,...
//   BatchNormalization (inference)               -> y = x*scale + shift, folded on the host
//   LSTM gate order i,f,c,o; Bidirectional concat [fwd, bwd]
#include <hip/hip_runtime.h>
#include <type_traits>

#include <cstdint>

#include "lds_dma.h"
#include "orcai_hip.h"
#include "zero_fill.h"

int g_orcai_lstm_split = 1;  // 1: the LSTM recurrences of the f32 path run on split-f16 MFMA at f32 accuracy; 0: on v_mfma_f32_16x16x4_f32 (orcai_lstm_split)

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// =========================================================================================
// Activation layout ("padded channel-quad planes"), fp32:
//     [snippet][CQ = ceil(C/4)][HP = H + 2R][WP = roundup4(W + R)][4]          R = k/2
//   - 4 consecutive channels are interleaved per pixel, so one pixel of one quad is a 16-byte vector: a wave
//     reading/writing 64 consecutive pixels moves 1 KiB contiguous per instruction (dwordx4 per lane), and the
//     4 channels of a quad are exactly one k-step of v_mfma_f32_16x16x4_f32;
//   - R zero rows above and below and >= R zero columns on the right of every plane (the right pad doubles as the
//     left pad of the next row): image pixel (y, x) of a quad sits at flat pixel (y + R)*WP + x.  The pads are
//     zero-filled once by the host and never written, so "same" padding needs no bounds checks and a k x k
//     window is k loads at flat offsets (dy - R)*WP; channels past C inside the last quad are kept at zero.
// =========================================================================================

// XCD-aware block order.  Workgroups are dealt round-robin to the 8 XCDs (each with a private L2), so linear block L runs on
// XCD L % 8.  Remapping L -> start(L % 8) + L / 8 gives every XCD one contiguous band of the (snippet, window) space: windows
// of vertically adjacent rows -- which re-read the same input rows -- then share an L2 instead of each pulling the rows over
// the fabric again.  Bijective for any grid size; placement affects speed only.
__device__ __forceinline__ void xcd_remap(int& bx, int& by) {
  const unsigned nbx = gridDim.x, total = nbx * gridDim.y;
  const unsigned L = blockIdx.y * nbx + blockIdx.x;
  const unsigned k = L & 7u, q = total >> 3, r = total & 7u;
  const unsigned Lp = k * q + (k < r ? k : r) + (L >> 3);
  by = (int)(Lp / nbx);
  bx = (int)(Lp - (unsigned)by * nbx);
}

// conv0: Conv2D(16, k x k, same) on a single-channel input + folded BN + ReLU   (architectures.py:164-168)
template <int KS>
__global__ __launch_bounds__(256) void conv0_kernel(const float* __restrict__ in, int64_t snippet_stride, int H, int W, int WP,
                                                     const float* __restrict__ w /*[KS*KS][16]*/, const float* __restrict__ scale,
                                                     const float* __restrict__ shift, float* __restrict__ out /*[B][4][HP][WP][4]*/, int relu,
                                                     const float* __restrict__ bn_mean = nullptr, const float* __restrict__ bn_var = nullptr,
                                                     const float* __restrict__ bn_gamma = nullptr, const float* __restrict__ bn_beta = nullptr, float bn_eps = 0.0f) {
  // bn_mean != nullptr (training forward, second pass): v = fma(acc, scale, shift) as always, then y = fma(v, gamma * inv, beta - mean * gamma * inv)
  // -- the two roundings of conv0_affine followed by bn_planes_apply, so y0 is bit for bit what those two launches wrote, without v0 in HBM
  constexpr int TH = 8, TW = 32, R = KS / 2, HH = TH + KS - 1, HW = TW + KS - 1, HP_ = HW + 1;
  __shared__ float halo[HH][HP_];
  __shared__ float bn2_s[2][16];
  if (bn_mean && threadIdx.x < 16) {
    const float sc = bn_gamma[threadIdx.x] * rsqrtf(bn_var[threadIdx.x] + bn_eps);
    bn2_s[0][threadIdx.x] = sc;
    bn2_s[1][threadIdx.x] = bn_beta[threadIdx.x] - bn_mean[threadIdx.x] * sc;
  }
  const int b = blockIdx.z, y0 = blockIdx.y * TH, x0 = blockIdx.x * TW;
  const float* src = in + (int64_t)b * snippet_stride;
  for (int i = threadIdx.x; i < HH * HW; i += 256) {
    const int r = i / HW, c = i % HW;
    const int y = y0 + r - R, x = x0 + c - R;
    halo[r][c] = (y >= 0 && y < H && x >= 0 && x < W) ? src[(int64_t)y * W + x] : 0.0f;
  }
  __syncthreads();
  const int py = threadIdx.x / TW, px = threadIdx.x % TW;
  float acc[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) acc[c] = 0.0f;
#pragma unroll
  for (int dy = 0; dy < KS; ++dy)
#pragma unroll
    for (int dx = 0; dx < KS; ++dx) {
      const float v = halo[py + dy][px + dx];
      const float* wt = w + (dy * KS + dx) * 16;
#pragma unroll
      for (int c = 0; c < 16; ++c) acc[c] = fmaf(v, wt[c], acc[c]);
    }
  const int y = y0 + py, x = x0 + px;
  if (y < H && x < W) {
    const int64_t plane = (int64_t)(H + 2 * R) * WP;  // pixels per quad plane
    float4* o = reinterpret_cast<float4*>(out) + (int64_t)b * 4 * plane + (int64_t)(y + R) * WP + x;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float4 v;
      v.x = fmaf(acc[4 * q + 0], scale[4 * q + 0], shift[4 * q + 0]);
      v.y = fmaf(acc[4 * q + 1], scale[4 * q + 1], shift[4 * q + 1]);
      v.z = fmaf(acc[4 * q + 2], scale[4 * q + 2], shift[4 * q + 2]);
      v.w = fmaf(acc[4 * q + 3], scale[4 * q + 3], shift[4 * q + 3]);
      if (bn_mean) {
        v.x = fmaf(v.x, bn2_s[0][4 * q + 0], bn2_s[1][4 * q + 0]);
        v.y = fmaf(v.y, bn2_s[0][4 * q + 1], bn2_s[1][4 * q + 1]);
        v.z = fmaf(v.z, bn2_s[0][4 * q + 2], bn2_s[1][4 * q + 2]);
        v.w = fmaf(v.w, bn2_s[0][4 * q + 3], bn2_s[1][4 * q + 3]);
      }
      if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      o[(int64_t)q * plane] = v;
    }
  }
}

// The training forward's FIRST pass over the entry conv: v = fma(conv, scale, shift) is formed exactly as conv0_kernel forms it and only its
// batch statistics leave the kernel (per channel sum and sum of squares: f32 per thread -> wave (DPP) -> workgroup (LDS) -> one f64 atomic
// per value and workgroup into one of 32 accumulator copies, the layout orcai_bn_finish_sharded reads).  The second pass (conv0_kernel with
// the BatchNorm stage) recomputes the conv from the 1-channel input -- 32 MB per batch of 64 -- and writes y0 = relu(bn0(v0)) directly:
// v0 (0.52 GB per batch) is never written, re-read for its statistics or re-read by the apply pass.
template <int KS>
__global__ __launch_bounds__(256) void conv0_stats_kernel(const float* __restrict__ in, int64_t snippet_stride, int H, int W, int B,
                                                           const float* __restrict__ w /*[KS*KS][16]*/, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, double* __restrict__ shards /*[32][4][8]*/) {
  constexpr int TH = 8, TW = 32, R = KS / 2, HH = TH + KS - 1, HW = TW + KS - 1, HP_ = HW + 1;
  __shared__ float halo[HH][HP_];
  __shared__ float red[4][32];
  const int tx = (W + TW - 1) / TW, ty = (H + TH - 1) / TH;
  const int ntiles = tx * ty * B;
  const int py = threadIdx.x / TW, px = threadIdx.x % TW;
  float s1[16], s2[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) { s1[c] = 0.0f; s2[c] = 0.0f; }
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {  // a workgroup walks many tiles: one reduction per workgroup, not per tile
    const int b = tile / (tx * ty), rem = tile - b * (tx * ty);
    const int y0 = (rem / tx) * TH, x0 = (rem - (rem / tx) * tx) * TW;
    const float* src = in + (int64_t)b * snippet_stride;
    __syncthreads();  // the previous tile's halo reads are done
    for (int i = threadIdx.x; i < HH * HW; i += 256) {
      const int r = i / HW, c = i % HW;
      const int y = y0 + r - R, x = x0 + c - R;
      halo[r][c] = (y >= 0 && y < H && x >= 0 && x < W) ? src[(int64_t)y * W + x] : 0.0f;
    }
    __syncthreads();
    float acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = 0.0f;
#pragma unroll
    for (int dy = 0; dy < KS; ++dy)
#pragma unroll
      for (int dx = 0; dx < KS; ++dx) {
        const float v = halo[py + dy][px + dx];
        const float* wt = w + (dy * KS + dx) * 16;
#pragma unroll
        for (int c = 0; c < 16; ++c) acc[c] = fmaf(v, wt[c], acc[c]);
      }
    const bool live = (y0 + py) < H && (x0 + px) < W;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const float v = live ? fmaf(acc[c], scale[c], shift[c]) : 0.0f;
      s1[c] += v;
      s2[c] = fmaf(v, v, s2[c]);
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    float a = s1[c], q = s2[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      a += __shfl_xor(a, o, 64);
      q += __shfl_xor(q, o, 64);
    }
    if (lane == 0) { red[wave][c] = a; red[wave][16 + c] = q; }
  }
  __syncthreads();
  if (threadIdx.x < 32) {
    const double tot = ((double)red[0][threadIdx.x] + (double)red[1][threadIdx.x]) + ((double)red[2][threadIdx.x] + (double)red[3][threadIdx.x]);
    const int isq = threadIdx.x >> 4, c = threadIdx.x & 15;
    atomicAdd(&shards[((int64_t)(blockIdx.x & 31) * 4 + (c >> 2)) * 8 + isq * 4 + (c & 3)], tot);
  }
}

// =========================================================================================
// sepconv: [ReLU] -> depthwise k x k (same) -> pointwise 1x1 + bias -> folded BN -> [ReLU]
//          (architectures.py:174-189, :198-206)
//
// Register-tile formulation, no LDS, no barriers: one WAVE owns 64 consecutive flat pixels of the padded planes
// (lane = pixel); the inner 64 - 2R are valid outputs, the outer R on each side only feed the horizontal taps
// through lane shifts.  Per channel quad (= one MFMA k-step) the wave loads the k rows of its window as dwordx4
// (1 KiB contiguous per instruction), forms the 4 depthwise outputs d[0..3] (lane = pixel, one VGPR per channel)
// and turns them into the four B fragments of v_mfma_f32_16x16x4_f32 -- B[k = channel][col = pixel], one
// 16-pixel column tile per 16-lane row -- by a 4 x 4 transpose of 16-lane rows: two v_permlane32_swap + two
// v_permlane16_swap.  The pointwise weights are the A operand (row = output channel), so the 4 accumulator
// registers of a lane are 4 consecutive output channels of one pixel = one dwordx4 store of the output layout.
// The next quad's rows are loaded before the current quad's arithmetic (register double buffer).
// =========================================================================================
__device__ __forceinline__ void swap32(float& a, float& b) {  // a's lanes 32..63 <-> b's lanes 0..31
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
}
__device__ __forceinline__ void swap16(float& a, float& b) {  // a's odd 16-lane rows <-> b's even 16-lane rows
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
}
// value of lane (l + SH) for SH in [-3, 3] (lanes shifted in from outside the wave are don't-care)
// One DPP wave shift (wave_shr:1 / wave_shl:1, a VALU move) per lane of distance, instead of an LDS-pipe ds_bpermute.
template <int SH>
__device__ __forceinline__ float lane_shift(float v) {
  if constexpr (SH == 0) return v;
  else if constexpr (SH < 0)
    return lane_shift<SH + 1>(__uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x138 /*wave_shr:1*/, 0xf, 0xf, true)));
  else
    return lane_shift<SH - 1>(__uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x130 /*wave_shl:1*/, 0xf, 0xf, true)));
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// max(x, 0) as ONE v_max_f32: fmaxf() is compiled to two (a canonicalising v_max x, x first), and the depthwise stage of the
// separable convolutions is bound by VALU issue.  Activations are finite, so NaN handling is irrelevant.
__device__ __forceinline__ float max2(float a, float b) {  // one v_max_f32, no canonicalisation
  float r;
  asm("v_max_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float relu1(float x) {
  float r;
  asm("v_max_f32_e32 %0, 0, %1" : "=v"(r) : "v"(x));
  return r;
}

// Depthwise stage for one channel quad, lane = pixel.  rows[dy] = the quad's dwordx4 of window row dy; wq = the quad's taps
// [k*k][4] (channel innermost, wave-uniform -> scalar loads); relu_lo = 0 (ReLU on load) or -inf.
// By linearity the horizontal taps are applied to per-column partial sums: p_dx = sum_dy w[dy][dx] * a_dy (lane-local,
// packed two channels per v_pk_fma_f32), d = sum_dx shift(p_dx, dx - R): k - 1 lane shifts per channel instead of k*(k-1).
template <int KS, bool RELU>
__device__ __forceinline__ void dw_quad_impl(const float4 (&rows)[KS], const float* __restrict__ wq, float (&d)[4]) {
  constexpr int R = KS / 2;
  f32x2 p01[KS], p23[KS];
#pragma unroll
  for (int dx = 0; dx < KS; ++dx) { p01[dx] = (f32x2){0.f, 0.f}; p23[dx] = (f32x2){0.f, 0.f}; }
#pragma unroll
  for (int dy = 0; dy < KS; ++dy) {
    const f32x2 a01 = {RELU ? relu1(rows[dy].x) : rows[dy].x, RELU ? relu1(rows[dy].y) : rows[dy].y};
    const f32x2 a23 = {RELU ? relu1(rows[dy].z) : rows[dy].z, RELU ? relu1(rows[dy].w) : rows[dy].w};
#pragma unroll
    for (int dx = 0; dx < KS; ++dx) {
      const float* w = wq + (dy * KS + dx) * 4;
      const f32x2 w01 = {w[0], w[1]}, w23 = {w[2], w[3]};
      p01[dx] = a01 * w01 + p01[dx];
      p23[dx] = a23 * w23 + p23[dx];
    }
  }
  float acc[4] = {p01[R].x, p01[R].y, p23[R].x, p23[R].y};
#pragma unroll
  for (int dx = 0; dx < KS; ++dx) {
    if (dx == R) continue;
    const float v[4] = {p01[dx].x, p01[dx].y, p23[dx].x, p23[dx].y};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float sv;
      if (dx - R == -3) sv = lane_shift<-3>(v[j]);
      else if (dx - R == -2) sv = lane_shift<-2>(v[j]);
      else if (dx - R == -1) sv = lane_shift<-1>(v[j]);
      else if (dx - R == 1) sv = lane_shift<1>(v[j]);
      else if (dx - R == 2) sv = lane_shift<2>(v[j]);
      else sv = lane_shift<3>(v[j]);
      acc[j] += sv;
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) d[j] = acc[j];
}

// relu_lo: 0 -> ReLU on load, -inf -> none (wave-uniform branch: the 12 v_max per quad are not issued when there is no ReLU)
template <int KS>
__device__ __forceinline__ void dw_quad(const float4 (&rows)[KS], const float* __restrict__ wq, float relu_lo, float (&d)[4]) {
  if (relu_lo == 0.0f) dw_quad_impl<KS, true>(rows, wq, d);
  else dw_quad_impl<KS, false>(rows, wq, d);
}

template <int KS, int MT>
__global__ __launch_bounds__(256) void sepconv_kernel(const float* __restrict__ in /*[B][CQin][HP][WP][4]*/, int Cin, int H, int W, int WP, int relu_in,
                                                       const float* __restrict__ dw /*[CQin][KS*KS][4]*/, const float* __restrict__ pw /*[Cin][Cout]*/,
                                                       const float* __restrict__ scale, const float* __restrict__ shift, int Cout, int relu_out,
                                                       int out_layout, float* __restrict__ out, int tasks, uint32_t magic_WP, int lo, int RP, int H2, int WP2,
                                                       float* __restrict__ u_out /*optional [B][CQin][HP][WP][4]: the depthwise output*/) {
  constexpr int KK = KS * KS;
  const int R = RP;  // rows of zero padding of the planes (>= KS/2, the tap radius)
  const int VAL = 64 - 2 * lo;  // valid output lanes are [lo, 64 - lo); lo >= R
  const int lane = threadIdx.x & 63;
  int bx, b;
  xcd_remap(bx, b);
  const int task = bx * 4 + (threadIdx.x >> 6);
  if (task >= tasks) return;  // whole wave; the kernel has no barriers
  const int lk = lane >> 4, lj = lane & 15;
  const int plane = (H + 2 * R) * WP;  // pixels per quad plane
  const int CQ = (Cin + 3) >> 2, CQo = (Cout + 3) >> 2;
  const float4* src = reinterpret_cast<const float4*>(in) + (int64_t)b * CQ * plane;
  const int qbase = R * WP + task * VAL - lo;  // flat padded-plane pixel of lane 0
  const int q = qbase + lane;

  int ridx[KS];  // row-window load indices (clamped: only lanes whose outputs are discarded can leave the plane)
#pragma unroll
  for (int dy = 0; dy < KS; ++dy) {
    const int i = q + (dy - KS / 2) * WP;
    ridx[dy] = i < 0 ? 0 : (i >= plane ? plane - 1 : i);
  }

  f32x4 acc[MT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // register pipeline two quads deep: the rows of quads cq+1 and cq+2 are in flight while quad cq is being processed
  float4 nxt[KS], nx2[KS];
#pragma unroll
  for (int dy = 0; dy < KS; ++dy) nxt[dy] = src[ridx[dy]];
  if (CQ > 1) {
#pragma unroll
    for (int dy = 0; dy < KS; ++dy) nx2[dy] = src[plane + ridx[dy]];
  }
  const float relu_lo = relu_in ? 0.0f : -INFINITY;

  for (int cq = 0; cq < CQ; ++cq) {
    float4 cur[KS];
#pragma unroll
    for (int dy = 0; dy < KS; ++dy) { cur[dy] = nxt[dy]; nxt[dy] = nx2[dy]; }
    if (cq + 2 < CQ) {
      const float4* pn = src + (int64_t)(cq + 2) * plane;
#pragma unroll
      for (int dy = 0; dy < KS; ++dy) nx2[dy] = pn[ridx[dy]];
    }
    float afrag[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int ci = cq * 4 + lk, co = m * 16 + lj;
      const bool ok = ci < Cin && co < Cout;
      const float av = pw[ok ? ci * Cout + co : 0];
      afrag[m] = ok ? av : 0.0f;
    }
    float d[4];
    dw_quad<KS>(cur, dw + cq * 4 * KK, relu_lo, d);  // taps of the quad: [KK][4], wave-uniform -> scalar loads
    if (u_out) {  // training forward: keep u = depthwise output (left operand of the pointwise weight gradient)
      const int urow = (int)__umulhi((uint32_t)q, magic_WP);
      const int ux = q - urow * WP;
      if (lane >= lo && lane < 64 - lo && ux < W && urow < R + H)
        reinterpret_cast<float4*>(u_out)[((int64_t)b * CQ + cq) * plane + q] = make_float4(d[0], d[1], d[2], d[3]);
    }
    // d[j] = depthwise output of channel 4cq+j, lane = pixel.  4x4 transpose of 16-lane rows:
    // afterwards d[t] row g = (channel 4cq+g, pixels 16t..16t+15) = B fragment of column tile t.
    swap32(d[0], d[2]);
    swap32(d[1], d[3]);
    swap16(d[0], d[1]);
    swap16(d[2], d[3]);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m][t] = mfma16(afrag[m], d[t], acc[m][t]);
  }

  // ---- epilogue: D[row = 4*lk + r -> cout][col = lj -> pixel 16t + lj of the window]
  float sc_r[MT][4], sh_r[MT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = m * 16 + lk * 4 + r;
      sc_r[m][r] = co < Cout ? scale[co] : 0.0f;
      sh_r[m][r] = co < Cout ? shift[co] : 0.0f;
    }
  float4* outq = reinterpret_cast<float4*>(out) + (int64_t)b * CQo * plane;
  if (out_layout == 2) {
    // x-pooled output for the MaxPooling2D((3,2)) that follows (architectures.py:190): element (y, j) = max over the
    // column pair (2j, 2j+1) -- the pair's second column is ignored when it is past the image ("same" pads with -inf).
    // The window starts at an even flat pixel, so pairs are lanes (2k, 2k+1) of one column tile: one shfl_xor.
    const int Wx = (W + 1) >> 1, WPx = (Wx + 3) & ~3;
    float4* outx = reinterpret_cast<float4*>(out) + (int64_t)b * CQo * H * WPx;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int wl = 16 * t + lj;
      const int flat = qbase + wl;
      const int row = (int)__umulhi((uint32_t)flat, magic_WP);
      const int x = flat - row * WP;
      const bool live = wl >= lo && wl < 64 - lo && x < W && row < R + H && (x & 1) == 0;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = fmaf(acc[m][t][r], sc_r[m][r], sh_r[m][r]);
          if (relu_out) v[r] = fmaxf(v[r], 0.0f);
          const float other = __shfl_xor(v[r], 1, 64);
          v[r] = (x + 1 < W) ? fmaxf(v[r], other) : v[r];
        }
        const int oq = m * 4 + lk;
        if (live && oq < CQo) outx[((int64_t)oq * H + (row - R)) * WPx + (x >> 1)] = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
    return;
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int wl = 16 * t + lj;  // lane index of this pixel inside the window
    const int flat = qbase + wl;
    const int row = (int)__umulhi((uint32_t)flat, magic_WP);  // padded row
    const int x = flat - row * WP;
    const bool live = wl >= lo && wl < 64 - lo && x < W && row < R + H;  // row >= R always
    if (!live) continue;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = fmaf(acc[m][t][r], sc_r[m][r], sh_r[m][r]);  // channels past Cout: 0*0+0
        if (relu_out) v[r] = fmaxf(v[r], 0.0f);
      }
      const int oq = m * 4 + lk;  // output quad
      if (oq >= CQo) continue;
      if (out_layout == 0) {
        outq[(int64_t)oq * plane + flat] = make_float4(v[0], v[1], v[2], v[3]);
      } else if (out_layout == 3) {  // scatter-add to pixel (2y, 2x) of planes [B][CQo][H2 + 2R][WP2][4]: backward of a stride-2 1x1 conv
        const int64_t plane2 = (int64_t)(H2 + 2 * R) * WP2;
        float4* o2 = reinterpret_cast<float4*>(out) + ((int64_t)b * CQo + oq) * plane2 + (int64_t)(2 * (row - R) + R) * WP2 + 2 * x;
        float4 t4 = *o2;
        t4.x += v[0]; t4.y += v[1]; t4.z += v[2]; t4.w += v[3];
        *o2 = t4;
      } else {  // Keras Reshape((-1, W*C)) of NHWC: feature = x*Cout + co   (architectures.py:208)
        float* o = out + ((int64_t)b * H + (row - R)) * ((int64_t)W * Cout) + (int64_t)x * Cout + oq * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (oq * 4 + r < Cout) o[r] = v[r];
      }
    }
  }
}

// =========================================================================================
// conv0_sep: the entry convolution fused into the first separable convolution (inference only):
//   Conv2D(16, 3x3, same) + BN + ReLU  ->  [ReLU] -> depthwise 3x3 -> pointwise + bias -> BN -> [ReLU]   (architectures.py:164-179)
// The 16-channel entry activation (15 GB per hour of audio) is never written to nor read back from HBM: a wave computes it
// for its 64-pixel window at the three rows the depthwise stage needs, straight from the 1-channel spectrogram, one channel
// quad at a time, and hands the quad to the same depthwise / transpose / MFMA pipeline as sepconv_kernel.  The recomputation
// (three rows) is VALU work on a kernel that was bound by its loads; results are bit-identical to conv0_kernel followed by
// sepconv_kernel (same fma chains in the same order).
//   * input: 15 dwords per lane (5 rows x 3 columns) through a raw buffer resource over the snippet's H*W floats: rows outside
//     the snippet and columns outside the image (offset sentinel) are out of range and read as 0 = the "same" zero padding;
//   * the entry activation must itself be zero outside the image (it is the depthwise stage's padding): v_med3(x, 0, hi) with
//     hi = +inf inside / 0 outside is ReLU and mask in one instruction;
//   * the residual branch of block 1 (1x1 conv, stride 2) only ever samples pixels (2i, 2j): those are written to a compact
//     [B][4][ceil(H/2)][ceil(W/2)][4] tensor (1/4 of the activation) that orcai_pool_res_add reads with flag 2.
// =========================================================================================
__device__ __forceinline__ float relu_mask(float x, float hi) {  // min(max(x, 0), hi), hi in {0, +inf}
  float r;
  asm("v_med3_f32 %0, %1, 0, %2" : "=v"(r) : "v"(x), "v"(hi));
  return r;
}

// acc + in[HALF] * w for both halves of w / acc: v_pk_fma_f32 with its first operand broadcast from one half of a register
// pair.  Written as asm because the compiler, left to itself, gives every broadcast scalar a register pair of its own (the
// 15 inputs of conv0_sep_kernel then cost 30 VGPRs per set instead of 16) and unpacks a third of the fmas.
template <int HALF>
__device__ __forceinline__ f32x2 pk_fma_bcast(f32x2 in, f32x2 w, f32x2 acc) {
  f32x2 r;
  if (HALF) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(in), "s"(w), "v"(acc));
  else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(r) : "v"(in), "s"(w), "v"(acc));
  return r;
}

template <int MT>
__global__ __launch_bounds__(256) void conv0_sep_kernel(const float* __restrict__ in, int64_t snippet_stride, int H, int W, int WP,
                                                         const float* __restrict__ w0_ /*[9][16]*/, const float* __restrict__ sc0_, const float* __restrict__ sh0_,
                                                         const float* __restrict__ dw_ /*[4][9][4]*/, const float* __restrict__ pw /*[16][Cout]*/,
                                                         const float* __restrict__ scale, const float* __restrict__ shift, int Cout, int relu_out,
                                                         float* __restrict__ out /*[B][CQo][H+2][WP][4]*/, float* __restrict__ prev_sub, int tasks, uint32_t magic_WP,
                                                         int NW) {
  // A wave walks NW windows (task t0 + 4w: the four waves of a workgroup side by side) with the NEXT window's 15 input dwords
  // in flight while it computes the current one -- a one-window wave spent a quarter of its life waiting for its inputs.
  // Two register sets in a loop unrolled by two (no copies of in-flight registers); the epilogue's stores are unconditional
  // (dead lanes: zeros to a padding pixel), so that the compiler's wait counts are exact and the stores drain behind the
  // next window's arithmetic.
  constexpr int R = 1, lo = 1, VAL = 62, C0 = 16;
  __shared__ float pw_s[C0 * 16 * MT];  // [(ci * 16 + lj)][m]
  __shared__ float sc_s[MT * 16], sh_s[MT * 16];
  const int lane = threadIdx.x & 63;
  int bx, b;
  xcd_remap(bx, b);
  const int t0 = bx * 4 * NW + (threadIdx.x >> 6);
  const int nw = min(NW, (tasks - t0 + 3) >> 2);
  const int lk = lane >> 4, lj = lane & 15;
  const int plane = (H + 2 * R) * WP;
  const int CQo = (Cout + 3) >> 2;
  const int Ho = (H + 1) >> 1, Wo = (W + 1) >> 1;
  const float lo_out = relu_out ? 0.0f : -INFINITY;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in + (int64_t)b * snippet_stride), 0, H * W * 4, 0x00020000);
  // 32-bit byte offsets from wave-uniform bases: no 64-bit (quarter-rate) multiplies per store
  char* sub_base = reinterpret_cast<char*>(prev_sub) + (int64_t)b * (C0 / 4) * Ho * Wo * 16;
  char* outq = reinterpret_cast<char*>(reinterpret_cast<float4*>(out) + (int64_t)b * CQo * plane);

  // the 5 x 3 input neighbourhood of window t (clamped to the last window: the stream's tail prefetch)
  // input (row d, column j) lives in half (3d + j) & 1 of register pair (3d + j) >> 1: v_pk_fma_f32 takes its broadcast operand
  // from either half of an aligned pair (op_sel), so 15 inputs cost 16 registers -- one input per pair would cost 30
  auto load_inputs = [&](int t, f32x2 (&inp)[8]) {
    const int q = R * WP + min(t, tasks - 1) * VAL - lo + lane;
    const int prow = (int)__umulhi((uint32_t)q, magic_WP);
    const int x = q - prow * WP, iy = prow - R;
    constexpr uint32_t OOB = 0x80000000u;  // stays out of range after adding a few row pitches
    const uint32_t center = (uint32_t)((iy * W + x) * 4);
    const uint32_t off[3] = {(x >= 1 && x <= W) ? center - 4u : OOB, x < W ? center : OOB, x + 1 < W ? center + 4u : OOB};
#pragma unroll
    for (int d = 0; d < 5; ++d)
#pragma unroll
      for (int j = 0; j < 3; ++j)
        inp[(3 * d + j) >> 1][(3 * d + j) & 1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, off[j] + (uint32_t)((d - 2) * W * 4), 0, 0));
  };

  auto window = [&](int t, const f32x2 (&inp)[8]) {
    const int qbase = R * WP + t * VAL - lo;
    const int q = qbase + lane;
    const int prow = (int)__umulhi((uint32_t)q, magic_WP);
    const int x = q - prow * WP, iy = prow - R;  // image coordinates of this lane's pixel (iy >= -1; x >= W: padding column)
    float hi[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) hi[r] = (x < W && iy + r - 1 >= 0 && iy + r - 1 < H) ? INFINITY : 0.0f;
    const bool sub_lane = prev_sub && lane >= lo && lane < 64 - lo && x < W && iy >= 0 && iy < H && ((x | iy) & 1) == 0;
    const uint32_t sub_off = (uint32_t)((iy >> 1) * Wo + (x >> 1)) * 16u;

    f32x4 acc[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) acc[m][tt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // the weights are re-read through the scalar cache every window: hoisted out of the window loop, the ~230 scalars do not
    // fit the SGPR file and come back as v_readlane / unpacked fmas
    int opaque_zero = 0;
    asm volatile("" : "+s"(opaque_zero));
    const float* w0 = static_cast<const float*>(__builtin_assume_aligned(w0_ + opaque_zero, 16));
    const float* sc0 = static_cast<const float*>(__builtin_assume_aligned(sc0_ + opaque_zero, 16));
    const float* sh0 = static_cast<const float*>(__builtin_assume_aligned(sh0_ + opaque_zero, 16));
    const float* dw = static_cast<const float*>(__builtin_assume_aligned(dw_ + opaque_zero, 16));

#pragma unroll
    for (int cq = 0; cq < C0 / 4; ++cq) {
      float afrag[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) afrag[m] = pw_s[((cq * 4 + lk) * 16 + lj) * MT + m];
      // entry convolution of channels 4cq..4cq+3 at rows iy-1, iy, iy+1: the fma chain of conv0_kernel (taps in dy, dx order)
      const f32x2 s01 = {sc0[cq * 4 + 0], sc0[cq * 4 + 1]}, s23 = {sc0[cq * 4 + 2], sc0[cq * 4 + 3]};
      const f32x2 h01 = {sh0[cq * 4 + 0], sh0[cq * 4 + 1]}, h23 = {sh0[cq * 4 + 2], sh0[cq * 4 + 3]};
      float4 c0[3];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        f32x2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            const float* wt = w0 + (dy * 3 + dx) * C0 + cq * 4;
            const f32x2 w01 = {wt[0], wt[1]}, w23 = {wt[2], wt[3]};
            const int ii = 3 * (r + dy) + dx;
            a01 = ((ii & 1) ? pk_fma_bcast<1>(inp[ii >> 1], w01, a01) : pk_fma_bcast<0>(inp[ii >> 1], w01, a01));
            a23 = ((ii & 1) ? pk_fma_bcast<1>(inp[ii >> 1], w23, a23) : pk_fma_bcast<0>(inp[ii >> 1], w23, a23));
          }
        a01 = a01 * s01 + h01;
        a23 = a23 * s23 + h23;
        c0[r] = make_float4(relu_mask(a01.x, hi[r]), relu_mask(a01.y, hi[r]), relu_mask(a23.x, hi[r]), relu_mask(a23.y, hi[r]));
      }
      if (sub_lane) *reinterpret_cast<float4*>(sub_base + (sub_off + (uint32_t)(cq * Ho * Wo) * 16u)) = c0[1];
      float d[4];
      dw_quad_impl<3, false>(c0, dw + cq * 36, d);  // the entry activation is already >= 0: the separable conv's ReLU is the identity
      swap32(d[0], d[2]);
      swap32(d[1], d[3]);
      swap16(d[0], d[1]);
      swap16(d[2], d[3]);
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m][tt] = mfma16(afrag[m], d[tt], acc[m][tt]);
    }

    // ---- epilogue: D[row = 4*lk + r -> cout][col = lj -> pixel 16tt + lj of the window]; always 4*MT stores
    const uint32_t lk_off = (uint32_t)(lk * plane + qbase + lj);  // quad lk, pixel lj of the window
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      const int wl = 16 * tt + lj;
      const int flat = qbase + wl;
      const int row = (int)__umulhi((uint32_t)flat, magic_WP);
      const int xx = flat - row * WP;
      const bool live = wl >= lo && wl < 64 - lo && xx < W && row < R + H;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const float4 sc = reinterpret_cast<const float4*>(sc_s)[m * 4 + lk], sh = reinterpret_cast<const float4*>(sh_s)[m * 4 + lk];
        const bool ok = live && m * 4 + lk < CQo;
        const float4 val = make_float4(max2(fmaf(acc[m][tt][0], sc.x, sh.x), lo_out), max2(fmaf(acc[m][tt][1], sc.y, sh.y), lo_out),
                                       max2(fmaf(acc[m][tt][2], sc.z, sh.z), lo_out), max2(fmaf(acc[m][tt][3], sc.w, sh.w), lo_out));
        *reinterpret_cast<float4*>(outq + (ok ? lk_off + (uint32_t)(m * 4 * plane + 16 * tt) : 0u) * 16u) = ok ? val : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };

  // The first window's inputs are requested before the LDS fill: the fill's own loads are younger, so its wait retires these
  // too and the loop is entered with nothing outstanding (the state the steady-state wait counts assume).
  f32x2 ia[8], ib[8];
  load_inputs(t0, ia);
  __builtin_amdgcn_sched_barrier(0);
  for (int i = threadIdx.x; i < C0 * 16 * MT; i += 256) {
    const int m = i % MT, lj_ = (i / MT) % 16, ci = i / (16 * MT), co = m * 16 + lj_;
    pw_s[i] = co < Cout ? pw[ci * Cout + co] : 0.0f;
  }
  if (threadIdx.x < MT * 16) {
    const int co = threadIdx.x;
    sc_s[co] = co < Cout ? scale[co] : 0.0f;
    sh_s[co] = co < Cout ? shift[co] : 0.0f;
  }
  __syncthreads();  // the only barrier

  if (t0 >= tasks) return;
#pragma unroll 1
  for (int w = 0; w < nw; w += 2) {
    load_inputs(t0 + 4 * (w + 1), ib);
    __builtin_amdgcn_sched_barrier(0);
    window(t0 + 4 * w, ia);
    __builtin_amdgcn_sched_barrier(0);
    if (w + 1 >= nw) break;
    load_inputs(t0 + 4 * (w + 2), ia);
    __builtin_amdgcn_sched_barrier(0);
    window(t0 + 4 * (w + 1), ib);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// =========================================================================================
// conv0_sep_tile: conv0_sep_kernel on 2-D strip tiles with the entry activation shared through LDS.  conv0_sep_kernel computes the
// 16-channel entry activation three times (once per row of every window's depthwise stencil); here a workgroup of TRW waves owns
// TRW - 2 image rows x 64 columns, wave w computes the entry activation of ONE row (tile row w = image row r0 - 1 + w, all 16
// channels: 1 / 3 of the per-window entry arithmetic, (TRW) / (TRW - 2) rows per output row) from 9 input dwords per lane and
// writes it to LDS as [row][quad][lane][4]; after one barrier the first TRW - 2 waves run the depthwise / transpose / MFMA
// pipeline of one output row each, reading their three rows per quad with ds_read_b128.  No register set of in-flight inputs, no
// per-window recomputation: 64 VGPRs.  Same fma chains in the same order as conv0_kernel + sepconv_kernel: bit-identical.
// =========================================================================================
template <int MT, int TRW>
__global__ __launch_bounds__(64 * TRW) void conv0_sep_tile_kernel(const float* __restrict__ in, int64_t snippet_stride, int H, int W, int WP,
                                                                const float* __restrict__ w0_ /*[9][16]*/, const float* __restrict__ sc0_,
                                                                const float* __restrict__ sh0_, const float* __restrict__ dw_ /*[4][9][4]*/,
                                                                const float* __restrict__ pw /*[16][Cout]*/, const float* __restrict__ scale,
                                                                const float* __restrict__ shift, int Cout, int relu_out, float* __restrict__ out,
                                                                float* __restrict__ prev_sub, int nstrip) {
  constexpr int R = 1, lo = 1, VAL = 62, C0 = 16, TR = TRW - 2;
  __shared__ __attribute__((aligned(16))) float act_s[TRW][C0 / 4][256];  // entry activation: [tile row][quad][lane][4]
  __shared__ float pw_s[C0 * 16 * MT];                                     // [(ci * 16 + lj)][m]
  __shared__ float sc_s[MT * 16], sh_s[MT * 16];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int bx, b;
  xcd_remap(bx, b);
  const int rg = bx / nstrip, strip = bx - rg * nstrip;
  const int r0 = rg * TR, c0 = strip * VAL;
  const int lk = lane >> 4, lj = lane & 15;
  const int plane = (H + 2 * R) * WP;
  const int CQo = (Cout + 3) >> 2;
  const int Ho = (H + 1) >> 1, Wo = (W + 1) >> 1;
  const float lo_out = relu_out ? 0.0f : -INFINITY;

  // ---- phase 1: the entry activation of image row e = r0 - 1 + wave at columns x = c0 - 1 + lane.  Inputs through a raw buffer
  // resource over the snippet's H * W floats: rows outside the snippet and columns outside the image (offset sentinel) read as 0.
  const int e = r0 - 1 + wave, x = c0 - lo + lane;
  {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in + (int64_t)b * snippet_stride), 0, H * W * 4, 0x00020000);
    constexpr uint32_t OOB = 0x80000000u;  // stays out of range after adding a row pitch
    const uint32_t center = (uint32_t)((e * W + x) * 4);
    const bool e_ok = e >= -1 && e <= H;  // beyond that even the neighbouring rows are outside (and center may wrap into range)
    const uint32_t off[3] = {(e_ok && x >= 1 && x <= W) ? center - 4u : OOB, (e_ok && x >= 0 && x < W) ? center : OOB, (e_ok && x >= -1 && x + 1 < W) ? center + 4u : OOB};
    f32x2 inp[5];  // input (row d, column j) in half (3d + j) & 1 of pair (3d + j) >> 1 (see conv0_sep_kernel)
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
      for (int j = 0; j < 3; ++j)
        inp[(3 * d + j) >> 1][(3 * d + j) & 1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, off[j] + (uint32_t)((d - 1) * W * 4), 0, 0));
    inp[4][1] = 0.0f;

    for (int i = threadIdx.x; i < C0 * 16 * MT; i += 64 * TRW) {
      const int m = i % MT, lj_ = (i / MT) % 16, ci = i / (16 * MT), co = m * 16 + lj_;
      pw_s[i] = co < Cout ? pw[ci * Cout + co] : 0.0f;
    }
    if (threadIdx.x < MT * 16) {
      const int co = threadIdx.x;
      sc_s[co] = co < Cout ? scale[co] : 0.0f;
      sh_s[co] = co < Cout ? shift[co] : 0.0f;
    }
    const float hi = (x >= 0 && x < W && e >= 0 && e < H) ? INFINITY : 0.0f;  // the activation is zero outside the image: the depthwise padding
    // the residual branch's (2i, 2j) subsample: written by the row's owner (tile rows 1 .. TR, lanes lo .. 63 - lo)
    const bool sub_lane = prev_sub && wave >= 1 && wave <= TR && lane >= lo && lane < 64 - lo && x < W && e < H && ((x | e) & 1) == 0;
    float4* sub = reinterpret_cast<float4*>(prev_sub) + (int64_t)b * (C0 / 4) * Ho * Wo + (e >> 1) * Wo + (x >> 1);
    const float* w0 = static_cast<const float*>(__builtin_assume_aligned(w0_, 16));
    const float* sc0 = static_cast<const float*>(__builtin_assume_aligned(sc0_, 16));
    const float* sh0 = static_cast<const float*>(__builtin_assume_aligned(sh0_, 16));
#pragma unroll
    for (int cq = 0; cq < C0 / 4; ++cq) {
      const f32x2 s01 = {sc0[cq * 4 + 0], sc0[cq * 4 + 1]}, s23 = {sc0[cq * 4 + 2], sc0[cq * 4 + 3]};
      const f32x2 h01 = {sh0[cq * 4 + 0], sh0[cq * 4 + 1]}, h23 = {sh0[cq * 4 + 2], sh0[cq * 4 + 3]};
      f32x2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const float* wt = w0 + (dy * 3 + dx) * C0 + cq * 4;
          const f32x2 w01 = {wt[0], wt[1]}, w23 = {wt[2], wt[3]};
          const int ii = 3 * dy + dx;
          a01 = ((ii & 1) ? pk_fma_bcast<1>(inp[ii >> 1], w01, a01) : pk_fma_bcast<0>(inp[ii >> 1], w01, a01));
          a23 = ((ii & 1) ? pk_fma_bcast<1>(inp[ii >> 1], w23, a23) : pk_fma_bcast<0>(inp[ii >> 1], w23, a23));
        }
      a01 = a01 * s01 + h01;
      a23 = a23 * s23 + h23;
      const float4 c = make_float4(relu_mask(a01.x, hi), relu_mask(a01.y, hi), relu_mask(a23.x, hi), relu_mask(a23.y, hi));
      *reinterpret_cast<float4*>(&act_s[wave][cq][lane * 4]) = c;
      if (sub_lane) sub[(int64_t)cq * Ho * Wo] = c;
    }
  }
  __syncthreads();
  const int y = r0 + wave;  // this wave's output row
  if (wave >= TR || y >= H) return;

  // ---- phase 2: depthwise 3x3 over tile rows wave .. wave + 2, pointwise on the MFMA
  f32x4 acc[MT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) acc[m][tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float* dw = static_cast<const float*>(__builtin_assume_aligned(dw_, 16));
#pragma unroll
  for (int cq = 0; cq < C0 / 4; ++cq) {
    float4 c0r[3];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) c0r[dy] = *reinterpret_cast<const float4*>(&act_s[wave + dy][cq][lane * 4]);
    float afrag[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) afrag[m] = pw_s[((cq * 4 + lk) * 16 + lj) * MT + m];
    float d[4];
    dw_quad_impl<3, false>(c0r, dw + cq * 36, d);  // the entry activation is already >= 0: the separable conv's ReLU is the identity
    swap32(d[0], d[2]);
    swap32(d[1], d[3]);
    swap16(d[0], d[1]);
    swap16(d[2], d[3]);
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m][tt] = mfma16(afrag[m], d[tt], acc[m][tt]);
  }
  float4* outb = reinterpret_cast<float4*>(out) + (int64_t)b * CQo * plane + (R + y) * WP;
#pragma unroll
  for (int tt = 0; tt < 4; ++tt) {
    const int wl = 16 * tt + lj;
    const int xx = c0 - lo + wl;
    const bool live = wl >= lo && wl < 64 - lo && xx < W;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const float4 sc = reinterpret_cast<const float4*>(sc_s)[m * 4 + lk], sh = reinterpret_cast<const float4*>(sh_s)[m * 4 + lk];
      const int oq = m * 4 + lk;
      if (live && oq < CQo)
        outb[(int64_t)oq * plane + xx] = make_float4(max2(fmaf(acc[m][tt][0], sc.x, sh.x), lo_out), max2(fmaf(acc[m][tt][1], sc.y, sh.y), lo_out),
                                                    max2(fmaf(acc[m][tt][2], sc.z, sh.z), lo_out), max2(fmaf(acc[m][tt][3], sc.w, sh.w), lo_out));
    }
  }
}

// =========================================================================================
// sepconv_tile: the arithmetic of sepconv_kernel<3, MT> with the window rows shared through LDS, for planes
// several windows wide (block 1).  A workgroup of TR waves owns a 2-D tile of TR image rows x 64 columns (one row per wave); per
// input quad the tile's TR + 2 rows are fetched ONCE per workgroup by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave-instruction,
// lane-linear in LDS, no VGPR destination) into one of two slots, and every wave reads its three rows back with ds_read_b128.
// L1 sees TR + 2 row requests per quad and tile instead of the 3 TR of independent windows -- the L1 fill path is the busiest
// unit of the one-window kernels -- and with no row registers the kernel fits 64 VGPRs = 8 waves per SIMD.
// One raw s_barrier per quad; the DMA of the next quad is issued right behind it (asm-issued, so the compiler's wait bookkeeping
// neither sees nor drains it; the wait in front of the barrier is a counted vmcnt that leaves the depthwise-output store of the
// training forward outstanding).  Ring depths of 3 and 4 slots measured equal to 2: the other 7 waves of the SIMD cover the latency.
// Same fma chains in the same order as the other two kernels: bit-identical results.
// =========================================================================================
using orcai_lds::glds16;
using orcai_lds::wait_vm_barrier;

// EPI (epilogue extras, all on plane output): 0 none; 1 BatchNorm batch statistics of the output (sums / sums of squares -> shards: the training
// forward); 2 BatchNorm BACKWARD sums of the output taken as the gradient dy of a BatchNorm whose input `ref` has the output's layout: sum g and
// sum g * xhat with g = dy * [relu gate], xhat = (ref - mean) * inv -> shards (the input-gradient pass of a block's second separable conv
// produces dy_a: the separate read pass over (dy_a, v_a) is gone); 3 out = ref > 0 ? out : 0 (ReLU backward folded into the
// input-gradient pass of a block's first separable conv).
struct EpiRef {
  const float* ref = nullptr;                                                        // EPI 2, 3: [B][CQo][HP][WP][4]
  const float *mean = nullptr, *var = nullptr, *gamma = nullptr, *beta = nullptr;  // EPI 2
  float eps = 0.0f;
  int relu = 0;
};

// BNIN: the input planes hold the PRE-normalisation tensor v of a BatchNorm (+ ReLU): y = max(fma(v, gamma * inv, beta - mean * gamma * inv), 0)
// is formed where a row leaves LDS -- the same two roundings as orcai_bn_planes_apply, so the conv sees bit for bit the tensor that pass
// would have materialised -- and forced to zero outside the image (the planes' pads hold v = 0, but "same" padding pads y).
struct InBn {
  const float *mean = nullptr, *var = nullptr, *gamma = nullptr, *beta = nullptr;
  float eps = 0.0f;
};

template <int MT, int CQ, bool XP, bool RELU, int TR, bool UOUT, int EPI = 0, bool BNIN = false>
__global__ __launch_bounds__(64 * TR) void sepconv_tile_kernel(const float* __restrict__ in /*[B][CQ][HP][WP][4]*/, int Cin, int H, int W, int WP,
                                                             const float* __restrict__ dw /*[CQ][9][4]*/, const float* __restrict__ pw /*[Cin][Cout]*/,
                                                             const float* __restrict__ scale, const float* __restrict__ shift, int Cout, int relu_out,
                                                             float* __restrict__ out, int nstrip, float* __restrict__ u_out /*UOUT: [B][CQ][HP][WP][4]*/,
                                                             double* __restrict__ shards = nullptr /*EPI 1, 2: [32][ceil(Cout/4)][8] sums / sums of squares of the output*/,
                                                             EpiRef er = EpiRef{}, InBn ib = InBn{}) {
  constexpr int KK = 9, R = 1, lo = XP ? 2 : 1, VAL = 64 - 2 * lo;
  constexpr bool STATS = EPI == 1 || EPI == 2;
  static_assert(TR == 4 || TR == 8, "rows (= waves) per workgroup");
  static_assert(!BNIN || !RELU, "BNIN carries its own ReLU");
  __shared__ __attribute__((aligned(16))) float inbn_s[BNIN ? 2 : 1][BNIN ? CQ * 4 : 4];  // BNIN: folded scale, shift per input channel
  static_assert(EPI == 0 || (!XP && MT == 2), "epilogue extras: plane output, two output tiles");
  __shared__ float stat_s[STATS ? TR : 1][4][16];  // STATS: per wave and 16-lane row, the row's 8 sums and 8 sums of squares
  __shared__ __attribute__((aligned(16))) float bn_s[EPI == 2 ? 4 : 1][EPI == 2 ? MT * 16 : 4];  // EPI 2: mean, inv, gamma, beta per output channel
  static_assert(!(XP && UOUT), "the training forward writes planes");
  __shared__ __attribute__((aligned(16))) float rows_s[2][TR + 2][256];  // [slot][tile row][lane][4]
  __shared__ float pw_s[CQ * 4 * 16 * MT];                               // [(ci * 16 + lj)][m]: a lane's MT A-fragment values are contiguous
  __shared__ float sc_s[MT * 16], sh_s[MT * 16];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int bx, b;
  xcd_remap(bx, b);
  const int rg = bx / nstrip, strip = bx - rg * nstrip;  // strips of a row group are neighbours in launch order: halo rows and columns hit L2
  const int r0 = rg * TR, c0 = strip * VAL;
  const int lk = lane >> 4, lj = lane & 15;
  const int plane = (H + 2 * R) * WP;
  const int CQo = (Cout + 3) >> 2;
  const int CQr = (Cin + 3) >> 2;  // real input quads (<= CQ; the launcher picks CQ = 4 or 8)
  const char* src = reinterpret_cast<const char*>(in) + (int64_t)b * CQr * plane * 16;
  const int Wx = (W + 1) >> 1, WPx = (Wx + 3) & ~3;
  float4* outb = reinterpret_cast<float4*>(out) + (XP ? (int64_t)b * CQo * H * WPx : (int64_t)b * CQo * plane);

  // tile row j (0 .. TR+1) is padded-plane row r0 + j; this wave fetches row `wave`, waves 0 and 1 also rows TR and TR + 1.
  // Offsets are clamped into the quad plane: only lanes / rows whose outputs are discarded can leave it.
  auto goff = [&](int j) {
    const int i = (r0 + j) * WP + c0 - lo + lane;
    return (uint32_t)(i < 0 ? 0 : (i >= plane ? plane - 1 : i)) * 16u;
  };
  const uint32_t off0 = goff(wave), off1 = goff(TR + (wave & 1));
  const uint32_t lds0 = (uint32_t)(uintptr_t)&rows_s[0][0][0];
  const bool two = wave < 2;
  auto issue = [&](int e) {
    const char* base = src + (int64_t)e * plane * 16;
    const uint32_t slot = lds0 + (uint32_t)((e & 1) * (TR + 2) * 1024);
    glds16(base + off0, slot + (uint32_t)wave * 1024u);
    if (two) glds16(base + off1, slot + (uint32_t)(TR + wave) * 1024u);
  };
  issue(0);

  for (int i = threadIdx.x; i < CQ * 4 * 16 * MT; i += 64 * TR) {
    const int m = i % MT, lj_ = (i / MT) % 16, ci = i / (16 * MT), co = m * 16 + lj_;
    pw_s[i] = (ci < Cin && co < Cout) ? pw[ci * Cout + co] : 0.0f;
  }
  if (threadIdx.x < MT * 16) {
    const int co = threadIdx.x;
    sc_s[co] = co < Cout ? scale[co] : 0.0f;
    sh_s[co] = co < Cout ? shift[co] : 0.0f;
    if (EPI == 2) {
      const int cc = co < Cout ? co : 0;
      bn_s[0][co] = er.mean[cc];
      bn_s[1][co] = rsqrtf(er.var[cc] + er.eps);
      bn_s[2][co] = er.gamma[cc];
      bn_s[3][co] = er.beta[cc];
    }
  }
  if (BNIN && threadIdx.x < CQ * 4) {
    const int ci = threadIdx.x, cc = ci < Cin ? ci : 0;
    const float sc = ib.gamma[cc] * rsqrtf(ib.var[cc] + ib.eps);  // exactly bn_planes_apply_kernel's arithmetic
    inbn_s[0][ci] = ci < Cin ? sc : 0.0f;
    inbn_s[1][ci] = ci < Cin ? ib.beta[cc] - ib.mean[cc] * sc : 0.0f;
  }
  __syncthreads();
  const float lo_out = relu_out ? 0.0f : -INFINITY;
  const int row = r0 + wave;  // this wave's image row
  const int xl = c0 - lo + lane;
  const bool u_live = UOUT && lane >= lo && lane < 64 - lo && xl < W && row < H;

  f32x4 acc[MT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int cq = 0; cq < CQ; ++cq) {
    if (cq < CQr) {  // workgroup-uniform
      // outstanding, oldest first: the DMA(s) of quad cq, then (UOUT, cq > 0, a wave on an image row: lane `lo` is always live, so
      // the store was issued) the depthwise-output store of quad cq - 1, which may stay in flight
      if (UOUT && cq > 0 && row < H) wait_vm_barrier<1>(); else wait_vm_barrier<0>();
      if (cq + 1 < CQr) issue(cq + 1);  // into the slot every wave finished reading before this barrier
      float4 rows[3];
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) rows[dy] = *reinterpret_cast<const float4*>(&rows_s[cq & 1][wave + dy][lane * 4]);
      if (BNIN) {
        const float4 s4 = reinterpret_cast<const float4*>(inbn_s[0])[cq], t4 = reinterpret_cast<const float4*>(inbn_s[1])[cq];
        const bool colok = xl >= 0 && xl < W;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          const bool ok = colok && (row + dy - 1) >= 0 && (row + dy - 1) < H;  // image row of tile row wave + dy
          rows[dy].x = ok ? fmaxf(fmaf(rows[dy].x, s4.x, t4.x), 0.0f) : 0.0f;
          rows[dy].y = ok ? fmaxf(fmaf(rows[dy].y, s4.y, t4.y), 0.0f) : 0.0f;
          rows[dy].z = ok ? fmaxf(fmaf(rows[dy].z, s4.z, t4.z), 0.0f) : 0.0f;
          rows[dy].w = ok ? fmaxf(fmaf(rows[dy].w, s4.w, t4.w), 0.0f) : 0.0f;
        }
      }
      float afrag[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) afrag[m] = pw_s[((cq * 4 + lk) * 16 + lj) * MT + m];
      float d[4];
      dw_quad_impl<3, RELU>(rows, dw + cq * 4 * KK, d);
      if (UOUT) {
        if (u_live) reinterpret_cast<float4*>(u_out)[((int64_t)b * CQr + cq) * plane + (R + row) * WP + xl] = make_float4(d[0], d[1], d[2], d[3]);
      }
      swap32(d[0], d[2]);
      swap32(d[1], d[3]);
      swap16(d[0], d[1]);
      swap16(d[2], d[3]);
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m][tt] = mfma16(afrag[m], d[tt], acc[m][tt]);
    }
  }
  // ---- epilogue: D[row = 4*lk + r -> cout][col = lj -> tile column 16*tt + lj]
  // Waves past the last image row: without the statistics epilogue they are done; with it they stay for its workgroup barrier
  // (a barrier behind a divergent return is undefined in HIP) and contribute nothing: `live` is false for all their lanes.
  const bool row_ok = row < H;
  if (!STATS && !row_ok) return;
  float st[STATS ? 16 : 1];  // STATS: [sum | sum of squares][m][r] over this lane's stored pixels
  if (STATS) {
#pragma unroll
    for (int j = 0; j < 16; ++j) st[j] = 0.0f;
  }
#pragma unroll
  for (int tt = 0; tt < 4; ++tt) {
    const int wl = 16 * tt + lj;
    const int x = c0 - lo + wl;
    const bool live = row_ok && wl >= lo && wl < 64 - lo && x < W && (!XP || (x & 1) == 0);
    const bool pair_ok = x + 1 < W;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const float4 sc = reinterpret_cast<const float4*>(sc_s)[m * 4 + lk], sh = reinterpret_cast<const float4*>(sh_s)[m * 4 + lk];
      float v[4] = {fmaf(acc[m][tt][0], sc.x, sh.x), fmaf(acc[m][tt][1], sc.y, sh.y), fmaf(acc[m][tt][2], sc.z, sh.z), fmaf(acc[m][tt][3], sc.w, sh.w)};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = max2(v[r], lo_out);
        if (XP) {  // max over the column pair (2j, 2j+1); the second column is ignored when it is past the image
          const float other = __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v[r]), 0xB1 /*quad_perm:[1,0,3,2]*/, 0xf, 0xf, true));
          v[r] = max2(v[r], pair_ok ? other : v[r]);
        }
        if (EPI == 1) {
          const float lv = live ? v[r] : 0.0f;
          st[(MT == 2 ? m : 0) * 4 + r] += lv;
          st[8 + (MT == 2 ? m : 0) * 4 + r] = fmaf(lv, lv, st[8 + (MT == 2 ? m : 0) * 4 + r]);
        }
      }
      const int oq = m * 4 + lk;
      const bool st_ok = live && oq < CQo;
      const int idx = XP ? ((oq * H + row) * WPx + (x >> 1)) : (oq * plane + (R + row) * WP + x);
      if (EPI == 2) {
        float4 rv4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (st_ok) rv4 = (reinterpret_cast<const float4*>(er.ref) + (int64_t)b * CQo * plane)[idx];
        const float4 mu4 = reinterpret_cast<const float4*>(bn_s[0])[m * 4 + lk], in4 = reinterpret_cast<const float4*>(bn_s[1])[m * 4 + lk];
        const float4 g4 = reinterpret_cast<const float4*>(bn_s[2])[m * 4 + lk], b4 = reinterpret_cast<const float4*>(bn_s[3])[m * 4 + lk];
        const float rvv[4] = {rv4.x, rv4.y, rv4.z, rv4.w}, mu[4] = {mu4.x, mu4.y, mu4.z, mu4.w}, iv[4] = {in4.x, in4.y, in4.z, in4.w};
        const float gm[4] = {g4.x, g4.y, g4.z, g4.w}, bt[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float xh = (rvv[r] - mu[r]) * iv[r];
          const bool gate = !er.relu || fmaf(xh, gm[r], bt[r]) > 0.0f;
          const float gg = (st_ok && gate) ? v[r] : 0.0f;
          st[m * 4 + r] += gg;
          st[8 + m * 4 + r] = fmaf(gg, xh, st[8 + m * 4 + r]);
        }
      }
      if (st_ok) outb[idx] = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
  if (STATS) {
    // BatchNorm batch statistics of the tensor just written (the training forward's separate read pass over it): the 16 lanes of a row
    // hold the same 4 channel quads' values for 16 pixels -> inclusive DPP row scan (lane 15 of the row ends with the row's total),
    // rows of the 8 waves through LDS, one f64 atomic per value and workgroup into one of 32 accumulator copies
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      float a = st[j];
      a += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(a), 0x111 /*row_shr:1*/, 0xf, 0xf, true));
      a += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(a), 0x112 /*row_shr:2*/, 0xf, 0xf, true));
      a += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(a), 0x114 /*row_shr:4*/, 0xf, 0xf, true));
      a += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(a), 0x118 /*row_shr:8*/, 0xf, 0xf, true));
      st[j] = a;
    }
    if (lj == 15) {
#pragma unroll
      for (int j = 0; j < 16; j += 4) *reinterpret_cast<float4*>(&stat_s[wave][lk][j]) = make_float4(st[j], st[j + 1], st[j + 2], st[j + 3]);
    }
    __syncthreads();  // all TR waves (those past the last image row wrote zeros)
    if (wave == 0) {
      const int g = lane >> 4, j = lane & 15;  // channel quad (j >> 2 & 1) * 4 + g, element (j >> 3) * 4 + (j & 3) of its 8 doubles
      float tot = 0.0f;
#pragma unroll
      for (int w2 = 0; w2 < TR; ++w2) tot += stat_s[w2][g][j];
      const int cq = ((j >> 2) & 1) * 4 + g;
      if (cq < CQo) atomicAdd(&shards[(((bx + b * 7) & 31) * CQo + cq) * 8 + (j >> 3) * 4 + (j & 3)], (double)tot);
    }
  }
}

// =========================================================================================
// sepconv_pool_march: a residual block's SECOND separable convolution (+ folded BatchNorm) with the block's tail in its epilogue
// (architectures.py:172-196): MaxPooling2D((3, 2), strides 2, "same") of the conv output + Conv2D(C, 1, strides 2)(prev) + add, written as
// the next block's padded input planes.  The x-pooled conv output (7.6 MB per snippet in block 1, written by sepconv_tile_kernel<.., XP> and
// read straight back by pool_res_add_x_kernel) never reaches HBM.
//   * The conv part is sepconv_tile_kernel<2, CQ, XP = true>: 8 waves = 8 conv rows x 64 lanes (60 valid columns), the tile's 10 input rows
//     of a quad fetched once by LDS-DMA into one of two slots, one raw barrier per quad -- the same fma / MFMA chains, bit for bit.
//   * A workgroup MARCHES down its column strip over NT tiles.  A pooling window needs conv rows 2i, 2i + 1, 2i + 2: row 2i + 2 is shared
//     with window i + 1, and for the last window of a tile it is the NEXT tile's first row -- so the elementwise maximum of the tile's rows
//     6 and 7 is carried (in LDS) to the next tile instead of recomputing a conv row per tile.  A segment of NT tiles yields 4 NT - 1 pooled
//     rows from 8 NT - 1 conv rows (one wave idles in the last tile; one conv row per segment is computed by two segments).
//   * Tail of a tile: every wave writes its x-pooled row to an LDS exchange area [row][quad][30 pooled pixels][4] (two of the nine rows
//     alias the input slot of the tile's last quad, which is free until the next tile's second DMA), one barrier, then wave w finishes
//     pooled row (w >> 1) - 1 (-1 = the carried window) for output tile w & 1: max of three rows, residual 1x1 conv of prev at the pooled
//     pixel on the MFMA (prev read as 16-byte pixels, 4x4 lane-row transpose as everywhere), + bias, add, one 16-byte store per lane.
//     The next tile's first DMA is issued before the tail and lands while it runs.
// Arithmetic order = pool_res_add_x_kernel's (max first, then mx + (acc + bias)): bit-identical to the two-launch path.
// LDS: 20 KB slots + 7 x 3840 B exchange + 4 KB pointwise weights = 51.7 KB -> three workgroups per compute unit (measured: the conv part
// loses 1.5 % at three instead of four; profiles/r04_ab_lds_pad.log), registers capped at 80 for six waves per SIMD.
// =========================================================================================
template <int CQ, bool RELU>
__global__ __launch_bounds__(512, 6) void sepconv_pool_march_kernel(const float* __restrict__ in /*[B][CQr][H+2][WP][4]*/, int Cin, int H, int W, int WP,
                                                                   const float* __restrict__ dw /*[CQ][9][4]*/, const float* __restrict__ pw /*[Cin][Cout]*/,
                                                                   const float* __restrict__ scale, const float* __restrict__ shift, int Cout, int relu_out,
                                                                   const float* __restrict__ prev, int Cp, int prev_compact, const float* __restrict__ wr /*[Cp][Cout]*/,
                                                                   const float* __restrict__ br, float* __restrict__ out /*[B][CQo][Ho+2][WPo][4]*/, int Ho, int Wo, int WPo,
                                                                   int nstrip, int NT) {
  constexpr int KK = 9, R = 1, lo = 2, VAL = 60, TR = 8, MT = 2, NPX = VAL / 2, XROW = 8 * NPX * 4;  // XROW floats per exchange row: [8 quads][NPX][4]
  constexpr int SLOT = (TR + 2) * 256;  // floats per input slot: [tile row][lane][4]
  // one array, so that every pointer below is an LDS pointer to the compiler: [2 input slots][7 exchange rows: rows 2 .. 7 of the tile, then the carry
  // max(row 6, row 7) of the previous tile]; exchange rows 0 and 1 alias the input slot of the tile's last quad
  __shared__ __attribute__((aligned(16))) float lds_f[2 * SLOT + 7 * XROW];
  __shared__ float pw_s[CQ * 4 * 16 * MT];
  __shared__ float sc_s[MT * 16], sh_s[MT * 16];
  __shared__ float wr_s[16 * MT * 16];  // residual conv weights [ci][co] (zero outside Cp x Cout), read in the tail (kept in registers they spill)
  // LDS: 53 760 B for 8 input quads = 42 allocation granules of 1 280 B -- a third of a compute unit's 128 granules exactly; one granule more and only
  // TWO workgroups are resident (the first version, 54 144 B, ran 20 % slower for it)
  static_assert(2 * XROW <= (TR + 2) * 256, "two exchange rows alias one input slot");
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int bx, b;
  xcd_remap(bx, b);
  const int seg = bx / nstrip, strip = bx - seg * nstrip;
  const int NPS = 4 * NT - 1;  // pooled rows per segment
  const int p0 = seg * NPS, pend = (p0 + NPS < Ho) ? p0 + NPS : Ho;
  const int rb = 2 * p0, c0 = strip * VAL;
  // conv rows the segment's windows touch (H even: the image's last window has two rows; when the image ends on a tile boundary one more tile --
  // all of its rows below the image, its results discarded -- closes that window through the carry: 1 tile in 93 for orcai-V1's block 1)
  const int ntile = (2 * (pend - p0) + 1 + TR - 1) / TR;
  const int lk = lane >> 4, lj = lane & 15;
  const int plane = (H + 2 * R) * WP;
  const int CQo = (Cout + 3) >> 2, CQp = (Cp + 3) >> 2;
  constexpr int CQr = CQ;  // the launcher instantiates the exact number of input quads
  const char* src = reinterpret_cast<const char*>(in) + (int64_t)b * CQr * plane * 16;
  const uint32_t lds0 = (uint32_t)(uintptr_t)&lds_f[0];
  const bool two = wave < 2;

  // DMA of tile t, quad cq into slot sl: tile row j (0 .. TR + 1) is padded-plane row rb + t TR + j; this wave fetches row `wave`, waves 0 and 1
  // also rows TR and TR + 1.  Offsets are clamped into the quad plane: only lanes / rows whose outputs are discarded can leave it.
  int64_t plane16 = (int64_t)plane * 16;
  auto issue = [&](int t, int cq, int sl, int lane) {
    const int r0 = rb + t * TR;
    auto goff = [&](int j) {
      const int i = (r0 + j) * WP + c0 - lo + lane;
      return (uint32_t)(i < 0 ? 0 : (i >= plane ? plane - 1 : i)) * 16u;
    };
    const char* base = src + (int64_t)cq * plane16;
    const uint32_t slot = lds0 + (uint32_t)(sl * (TR + 2) * 1024);
    glds16(base + goff(wave), slot + (uint32_t)wave * 1024u);
    if (two) glds16(base + goff(TR + (wave & 1)), slot + (uint32_t)(TR + wave) * 1024u);
  };
  issue(0, 0, 0, lane);

  for (int i = threadIdx.x; i < CQ * 4 * 16 * MT; i += 64 * TR) {
    const int m = i % MT, lj_ = (i / MT) % 16, ci = i / (16 * MT), co = m * 16 + lj_;
    pw_s[i] = (ci < Cin && co < Cout) ? pw[ci * Cout + co] : 0.0f;
  }
  if (threadIdx.x < MT * 16) {
    const int co = threadIdx.x;
    sc_s[co] = co < Cout ? scale[co] : 0.0f;
    sh_s[co] = co < Cout ? shift[co] : 0.0f;
  }
  {
    const int ci = threadIdx.x >> 5, co = threadIdx.x & 31;  // 16 x 32 = 512 threads
    const bool ok = ci < Cp && co < Cout;
    const float av = wr[ok ? ci * Cout + co : 0];
    wr_s[threadIdx.x] = ok ? av : 0.0f;
  }
  // this wave's tail item: output tile tm of pooled row 4 t + tk.  The carried window (tk = -1, which also refreshes the carry) goes to waves 6 and 7:
  // waves 0 and 1 already fetch two rows per quad
  const int tm = wave & 1, tk = (wave >> 1) == 3 ? -1 : (wave >> 1);
  __syncthreads();
  const float lo_out = relu_out ? 0.0f : -INFINITY;
  const int plane_p = prev_compact ? Ho * Wo : plane, plane_o = (Ho + 2 * R) * WPo;
  const float4* prevb = reinterpret_cast<const float4*>(prev) + (int64_t)b * CQp * plane_p;
  float4* outb = reinterpret_cast<float4*>(out) + (int64_t)b * CQo * plane_o;
  bool vm_clean = false;

  for (int t = 0; t < ntile; ++t) {
    const int r0 = rb + t * TR;
    const int par = (t * CQr) & 1;  // slot of this tile's quad cq = (cq & 1) ^ par (a running count over tiles and quads)
    // the depthwise taps are re-read through the scalar cache every tile: hoisted out of the tile loop, the 36 CQ scalars do not fit the SGPR file
    // and come back as v_readlane traffic and spilled vector registers
    // ... and everything lane-dependent inside the tile loop is derived from an opaque copy of the lane id: hoisted out of the loop, two dozen
    // loop-invariant registers are spilled and come back through scratch memory (vector-memory operations in front of the counted waits)
    // (the lane id itself is re-derived by v_mbcnt: kept live across the loop it is the one register that still spilled, and its reload at the
    // top of a tile -- a vector-memory operation -- made the wave wait for its output stores after all)
    int lane_t;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_t));
    const int tlk = lane_t >> 4, tlj = lane_t & 15;
    int opaque_zero = 0;
    asm volatile("" : "+s"(opaque_zero));
    plane16 = (int64_t)plane * 16 + opaque_zero;  // (and the per-quad plane bases are formed where they are used)
    const float* dwt = static_cast<const float*>(__builtin_assume_aligned(dw + opaque_zero, 16));
    const int pr = p0 + 4 * t + tk;  // this wave's pooled row in the tail (tk = -1: the window the previous tile left open, closed by this tile's first row)
    const bool item = (tk >= 0 || t > 0) && pr < pend;  // wave-uniform
    // residual operand of the tail: prev at the pooled pixel (2 pr, 2 j), lane = pooled pixel of the strip (float4 index inside one quad plane);
    // derived where it is used, from a fresh lane id each time -- one more register alive across the quad loop spills
    auto prev_pixel = [&]() {
      int l;
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
      const int jq = (c0 >> 1) + (l < NPX ? l : NPX - 1);
      const int jqc = jq < Wo ? jq : Wo - 1;
      return prev_compact ? pr * Wo + jqc : (2 * pr + R) * WP + 2 * jqc;
    };
    f32x4 acc[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) acc[m][tt] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int cq = 0; cq < CQ; ++cq) {
      {
        // this quad's DMAs have landed; every wave has finished reading the other slot (and, at cq = 0, the previous tile's tail).  A wave that finished a tail
        // item has already waited for loads it issued AFTER this tile's first DMA (in-order return): it must not wait for its output stores here
        if (cq == 0 && vm_clean) asm volatile("s_barrier" ::: "memory"); else wait_vm_barrier<0>();
        const int sl = (cq & 1) ^ par;
        if (cq + 1 < CQr) issue(t, cq + 1, sl ^ 1, lane_t);
        else if (t + 1 < ntile) issue(t + 1, 0, sl ^ 1, lane_t);  // lands during the tail

        float4 rows[3];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) rows[dy] = *reinterpret_cast<const float4*>(&lds_f[sl * SLOT + (wave + dy) * 256 + lane_t * 4]);
        float afrag[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) afrag[m] = pw_s[((cq * 4 + tlk) * 16 + tlj) * MT + m];
        float d[4];
        dw_quad_impl<3, RELU>(rows, dwt + cq * 4 * KK, d);
        swap32(d[0], d[2]);
        swap32(d[1], d[3]);
        swap16(d[0], d[1]);
        swap16(d[2], d[3]);
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[m][tt] = mfma16(afrag[m], d[tt], acc[m][tt]);
      }
    }
    // ---- the tail's operands that come from HBM: prev at the pooled pixel (2 pr, 2 j), lane = pooled pixel of the strip, 4 channels per load
    // (always four quads: quads past Cp re-read quad 0 against zero weights)
    float4 pv[4];
    auto request_prev = [&]() {
      const int srcpix = prev_pixel();
#pragma unroll
      for (int cq = 0; cq < 4; ++cq) pv[cq] = prevb[(int64_t)(cq < CQp ? cq : 0) * plane_p + srcpix];
    };
    // residual 1x1 conv of prev at the wave's 30 pooled pixels for output tile tm: pv (lane = pixel) -> B fragments by the 4x4 lane-row transpose,
    // the weights' A fragments from LDS, one MFMA chain per 16-pixel tile in pool_res_add_x_kernel's order
    f32x4 racc[2];
    auto residual = [&]() {
      float wa[4];
#pragma unroll
      for (int cq = 0; cq < 4; ++cq) wa[cq] = wr_s[(cq * 4 + tlk) * 32 + tm * 16 + tlj];
      racc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
      racc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int cq = 0; cq < 4; ++cq) {
        float d[4] = {pv[cq].x, pv[cq].y, pv[cq].z, pv[cq].w};
        swap32(d[0], d[2]);
        swap32(d[1], d[3]);
        swap16(d[0], d[1]);
        swap16(d[2], d[3]);
        racc[0] = mfma16(wa[cq], d[0], racc[0]);
        racc[1] = mfma16(wa[cq], d[1], racc[1]);
      }
    };

    // ---- folded BatchNorm (+ ReLU) and the column-pair maximum, exactly sepconv_tile_kernel<.., XP>'s epilogue; the values stay in `acc`
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      const int x = c0 - lo + 16 * tt + tlj;
      const bool pair_ok = x + 1 < W;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const float4 sc = reinterpret_cast<const float4*>(sc_s)[m * 4 + tlk], sh = reinterpret_cast<const float4*>(sh_s)[m * 4 + tlk];
        float v[4] = {fmaf(acc[m][tt][0], sc.x, sh.x), fmaf(acc[m][tt][1], sc.y, sh.y), fmaf(acc[m][tt][2], sc.z, sh.z), fmaf(acc[m][tt][3], sc.w, sh.w)};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = max2(v[r], lo_out);
          const float other = __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v[r]), 0xB1 /*quad_perm:[1,0,3,2]*/, 0xf, 0xf, true));
          acc[m][tt][r] = max2(v[r], pair_ok ? other : v[r]);
        }
      }
    }
    // E0: every wave has consumed the last quad's rows -- its slot is free until the DMA after next (issued behind the next tile's first barrier)
    asm volatile("s_barrier" ::: "memory");
    const int sfree = ((CQr - 1) & 1) ^ par;
    auto xrow = [&](int j) -> float* { return &lds_f[j < 2 ? sfree * SLOT + j * XROW : 2 * SLOT + (j - 2) * XROW]; };
    {
      float* xme = xrow(wave);
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        const int wl = 16 * tt + tlj;
        if ((tlj & 1) == 0 && wl >= lo && wl < 64 - lo) {
#pragma unroll
          for (int m = 0; m < MT; ++m)
            *reinterpret_cast<float4*>(&xme[((m * 4 + tlk) * NPX + ((wl - lo) >> 1)) * 4]) = make_float4(acc[m][tt][0], acc[m][tt][1], acc[m][tt][2], acc[m][tt][3]);
        }
      }
    }
    if (item) {  // between the exchange writes and their barrier (the accumulators are dead): what a wave would otherwise spend waiting for the others
      request_prev();
      residual();
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // E1: the tile's eight x-pooled rows are in LDS

    // ---- tail: pooled row `prow` of output tile tm = max of the exchange rows pa, pb, pc, + (residual + bias) -> out
    float* carry = &lds_f[2 * SLOT + 6 * XROW];
    auto finish = [&](int prow, const float* pa, const float* pb, const float* pc) {
      const int oq = tm * 4 + tlk;
      float brr[4];  // residual bias of the lane's four output channels (four cached 4-byte loads: no LDS left for them)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = oq * 4 + r;
        const float bv = br[co < Cout ? co : 0];
        brr[r] = co < Cout ? bv : 0.0f;
      }
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        const int px = 16 * t2 + tlj, pxc = px < NPX ? px : NPX - 1, j = (c0 >> 1) + px;
        const int idx = (oq * NPX + pxc) * 4;
        const float4 a = *reinterpret_cast<const float4*>(&pa[idx]), bq = *reinterpret_cast<const float4*>(&pb[idx]), c = *reinterpret_cast<const float4*>(&pc[idx]);
        const float mv[4] = {fmaxf(fmaxf(a.x, bq.x), c.x), fmaxf(fmaxf(a.y, bq.y), c.y), fmaxf(fmaxf(a.z, bq.z), c.z), fmaxf(fmaxf(a.w, bq.w), c.w)};
        if (px < NPX && j < Wo && oq < CQo) {
          float o[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (oq * 4 + r < Cout) ? mv[r] + (racc[t2][r] + brr[r]) : 0.0f;
          outb[(int64_t)oq * plane_o + (prow + R) * WPo + j] = make_float4(o[0], o[1], o[2], o[3]);
        }
      }
    };
    if (item) {
      const float* pa = tk < 0 ? carry : xrow(2 * tk);
      const float* pb = tk < 0 ? (r0 < H ? xrow(0) : pa) : xrow(2 * tk + 1);
      const float* pc = (tk >= 0 && r0 + 2 * tk + 2 < H) ? xrow(2 * tk + 2) : pa;
      finish(pr, pa, pb, pc);
    }
    if (tk < 0 && t + 1 < ntile) {  // waves 6 and 7: the new carry, max(row 6, row 7), for their half of the channels (the old carry was read above: same wave, in order)
      const float* p6 = xrow(6);
      const float* p7 = xrow(7);
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        const int px = 16 * t2 + tlj, pxc = px < NPX ? px : NPX - 1;
        const int idx = ((tm * 4 + tlk) * NPX + pxc) * 4;
        const float4 a = *reinterpret_cast<const float4*>(&p6[idx]), bq = *reinterpret_cast<const float4*>(&p7[idx]);
        *reinterpret_cast<float4*>(&carry[idx]) = make_float4(fmaxf(a.x, bq.x), fmaxf(a.y, bq.y), fmaxf(a.z, bq.z), fmaxf(a.w, bq.w));
      }
    }
    vm_clean = item;  // this wave waited for loads younger than the next tile's first DMA: that DMA has landed, only stores are in flight
  }
}

// =========================================================================================
// sepconv_ftile: the LDS-shared rows of sepconv_tile for ANY plane width.  The NWV waves of a workgroup own NWV consecutive 64-pixel
// windows of the flat padded plane (the mapping of sepconv_kernel: no strip waste on narrow planes); the rows above and below are
// the same flat range shifted by -WP / +WP, so the three rows of all NWV windows are ONE contiguous range of
// (NWV - 1) * VAL + 64 + 2 * WP pixels, fetched once per quad by LDS-DMA in 1-KiB chunks (chunk c by wave c % NWV) -- 14 chunks for
// 8 windows of a 176-pixel-wide plane, 8 chunks at width 44, against 24 row loads of independent windows.  Input quads are a run-time
// count (no dummy quads).  Two LDS slots, one raw barrier per quad, waits as in sepconv_tile.  Bit-identical to sepconv_kernel<3, MT>.
// =========================================================================================
template <int MT, bool XP, bool RELU, bool UOUT, int NWV, int EPI = 0, bool BNIN = false>
__global__ __launch_bounds__(64 * NWV) void sepconv_ftile_kernel(const float* __restrict__ in /*[B][CQ][HP][WP][4]*/, int Cin, int H, int W, int WP,
                                                               const float* __restrict__ dw /*[CQ][9][4]*/, const float* __restrict__ pw /*[Cin][Cout]*/,
                                                               const float* __restrict__ scale, const float* __restrict__ shift, int Cout, int relu_out,
                                                               float* __restrict__ out, int tasks, uint32_t magic_WP, int nchunk,
                                                               float* __restrict__ u_out /*UOUT: [B][CQ][HP][WP][4]*/,
                                                               double* __restrict__ shards = nullptr /*EPI 1, 2: [32][ceil(Cout/4)][8]*/, EpiRef er = EpiRef{},
                                                               InBn ib = InBn{}) {
  constexpr int KK = 9, R = 1, lo = XP ? 2 : 1, VAL = 64 - 2 * lo;
  constexpr bool STATS = EPI == 1 || EPI == 2;
  static_assert(!BNIN || !RELU, "BNIN carries its own ReLU");
  __shared__ __attribute__((aligned(16))) float inbn_s[BNIN ? 2 : 1][BNIN ? 64 : 4];  // BNIN: folded scale, shift per input channel (<= 64)
  static_assert(!(XP && UOUT), "the training forward writes planes");
  static_assert(EPI == 0 || !XP, "epilogue extras: plane output");
  __shared__ float stat_s[STATS ? NWV : 1][4][STATS ? 8 * MT : 1];  // STATS: per wave and 16-lane row, the row's sums and sums of squares
  __shared__ __attribute__((aligned(16))) float bn_s[EPI == 2 ? 4 : 1][EPI == 2 ? MT * 16 : 4];  // EPI 2: mean, inv, gamma, beta per output channel
  extern __shared__ __attribute__((aligned(16))) float smem_ft[];
  const int CQr = (Cin + 3) >> 2, CQo = (Cout + 3) >> 2;
  float* rows_s = smem_ft;                      // [2][nchunk][64][4]
  float* pw_s = smem_ft + 2 * nchunk * 256;     // [(ci * 16 + lj)][m]
  float* sc_s = pw_s + CQr * 64 * MT;           // [MT * 16]
  float* sh_s = sc_s + MT * 16;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int bx, b;
  xcd_remap(bx, b);
  const int lk = lane >> 4, lj = lane & 15;
  const int plane = (H + 2 * R) * WP;
  const char* src = reinterpret_cast<const char*>(in) + (int64_t)b * CQr * plane * 16;
  const int Wx = (W + 1) >> 1, WPx = (Wx + 3) & ~3;
  float4* outb = reinterpret_cast<float4*>(out) + (XP ? (int64_t)b * CQo * H * WPx : (int64_t)b * CQo * plane);

  // flat pixel of LDS position 0: lane 0 of the workgroup's first window, one row up (R * WP + first * VAL - lo - WP with R = 1)
  const int s0 = bx * NWV * VAL - lo;
  auto goff = [&](int c) {  // clamped into the quad plane: only lanes whose outputs are discarded can leave it
    const int i = s0 + 64 * c + lane;
    return (uint32_t)(i < 0 ? 0 : (i >= plane ? plane - 1 : i)) * 16u;
  };
  const uint32_t off0 = goff(wave), off1 = goff(wave + NWV), off2 = goff(wave + 2 * NWV);
  const uint32_t lds0 = (uint32_t)(uintptr_t)rows_s;
  const int mine = wave < nchunk ? (wave + NWV < nchunk ? (wave + 2 * NWV < nchunk ? 3 : 2) : 1) : 0;  // chunks this wave fetches (wave-uniform)
  auto issue = [&](int e) {
    const char* base = src + (int64_t)e * plane * 16;
    const uint32_t slot = lds0 + (uint32_t)((e & 1) * nchunk * 1024);
    if (mine > 0) glds16(base + off0, slot + (uint32_t)wave * 1024u);
    if (mine > 1) glds16(base + off1, slot + (uint32_t)(wave + NWV) * 1024u);
    if (mine > 2) glds16(base + off2, slot + (uint32_t)(wave + 2 * NWV) * 1024u);
  };
  issue(0);

  for (int i = threadIdx.x; i < CQr * 64 * MT; i += 64 * NWV) {
    const int m = i % MT, lj_ = (i / MT) % 16, ci = i / (16 * MT), co = m * 16 + lj_;
    pw_s[i] = (ci < Cin && co < Cout) ? pw[ci * Cout + co] : 0.0f;
  }
  if (threadIdx.x < MT * 16) {
    const int co = threadIdx.x;
    sc_s[co] = co < Cout ? scale[co] : 0.0f;
    sh_s[co] = co < Cout ? shift[co] : 0.0f;
    if (EPI == 2) {
      const int cc = co < Cout ? co : 0;
      bn_s[0][co] = er.mean[cc];
      bn_s[1][co] = rsqrtf(er.var[cc] + er.eps);
      bn_s[2][co] = er.gamma[cc];
      bn_s[3][co] = er.beta[cc];
    }
  }
  if (BNIN && threadIdx.x < 64) {
    const int ci = threadIdx.x, cc = ci < Cin ? ci : 0;
    const float sc = ib.gamma[cc] * rsqrtf(ib.var[cc] + ib.eps);  // exactly bn_planes_apply_kernel's arithmetic
    inbn_s[0][ci] = ci < Cin ? sc : 0.0f;
    inbn_s[1][ci] = ci < Cin ? ib.beta[cc] - ib.mean[cc] * sc : 0.0f;
  }
  __syncthreads();
  const float lo_out = relu_out ? 0.0f : -INFINITY;
  const int task = bx * NWV + wave;
  const bool wave_live = task < tasks;
  const int qbase = R * WP + task * VAL - lo;  // flat padded-plane pixel of lane 0 of this wave's window
  const int q = qbase + lane;
  bool u_live = false;
  if (UOUT) {
    const int urow = (int)__umulhi((uint32_t)q, magic_WP);
    const int ux = q - urow * WP;
    u_live = wave_live && lane >= lo && lane < 64 - lo && ux < W && urow < R + H;
  }
  // (wave_live: the window starts inside the image rows, so its lane `lo` ... not necessarily on an image column: the store may be
  // skipped by a whole wave whose live lanes all sit in the padding columns -> such waves wait for vmcnt(0), see below)
  const bool u_any = UOUT && __builtin_amdgcn_readfirstlane((int)(__ballot(u_live) != 0ull)) != 0;
  const float* rbase = rows_s + (wave * VAL + lane) * 4;
  int bn_row = 0;
  bool bn_colok = false;
  if (BNIN) {  // this lane's pixel of the flat padded plane: row / column once per window
    const int qq = q < 0 ? 0 : q;
    bn_row = (int)__umulhi((uint32_t)qq, magic_WP);
    bn_colok = q >= 0 && (qq - bn_row * WP) < W;
  }

  f32x4 acc[MT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int cq = 0; cq < CQr; ++cq) {
    // outstanding, oldest first: this wave's DMAs of quad cq, then (UOUT, cq > 0, a wave that stores) the depthwise-output store of
    // quad cq - 1, which may stay in flight
    if (u_any && cq > 0) wait_vm_barrier<1>(); else wait_vm_barrier<0>();
    if (cq + 1 < CQr) issue(cq + 1);  // into the slot every wave finished reading before this barrier
    const float* rs = rbase + (cq & 1) * nchunk * 256;
    float4 rows[3];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) rows[dy] = *reinterpret_cast<const float4*>(rs + dy * WP * 4);
    if (BNIN) {
      const float4 s4 = reinterpret_cast<const float4*>(inbn_s[0])[cq], t4 = reinterpret_cast<const float4*>(inbn_s[1])[cq];
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const bool ok = bn_colok && (bn_row + dy - 1) >= R && (bn_row + dy - 1) < R + H;  // padded-plane row of this lane's pixel, one up / same / one down
        rows[dy].x = ok ? fmaxf(fmaf(rows[dy].x, s4.x, t4.x), 0.0f) : 0.0f;
        rows[dy].y = ok ? fmaxf(fmaf(rows[dy].y, s4.y, t4.y), 0.0f) : 0.0f;
        rows[dy].z = ok ? fmaxf(fmaf(rows[dy].z, s4.z, t4.z), 0.0f) : 0.0f;
        rows[dy].w = ok ? fmaxf(fmaf(rows[dy].w, s4.w, t4.w), 0.0f) : 0.0f;
      }
    }
    float afrag[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) afrag[m] = pw_s[((cq * 4 + lk) * 16 + lj) * MT + m];
    float d[4];
    dw_quad_impl<3, RELU>(rows, dw + cq * 4 * KK, d);
    if (UOUT) {
      if (u_live) reinterpret_cast<float4*>(u_out)[((int64_t)b * CQr + cq) * plane + q] = make_float4(d[0], d[1], d[2], d[3]);
    }
    swap32(d[0], d[2]);
    swap32(d[1], d[3]);
    swap16(d[0], d[1]);
    swap16(d[2], d[3]);
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m][tt] = mfma16(afrag[m], d[tt], acc[m][tt]);
  }
  // ---- epilogue (see sepconv_kernel): D[row = 4*lk + r -> cout][col = lj -> pixel 16*tt + lj of the window]
  if (!STATS && !wave_live) return;  // with the statistics epilogue every wave stays for its workgroup barrier; `live` is false for all lanes of a wave without a window
  float st[STATS ? 8 * MT : 1];  // STATS: [sum | sum of squares][m][r] over this lane's stored pixels
  if (STATS) {
#pragma unroll
    for (int j = 0; j < 8 * MT; ++j) st[j] = 0.0f;
  }
#pragma unroll
  for (int tt = 0; tt < 4; ++tt) {
    const int wl = 16 * tt + lj;
    const int flat = qbase + wl;
    const int row = (int)__umulhi((uint32_t)flat, magic_WP);
    const int x = flat - row * WP;
    const bool live = wave_live && wl >= lo && wl < 64 - lo && x < W && row < R + H && (!XP || (x & 1) == 0);
    const bool pair_ok = x + 1 < W;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const float4 sc = reinterpret_cast<const float4*>(sc_s)[m * 4 + lk], sh = reinterpret_cast<const float4*>(sh_s)[m * 4 + lk];
      float v[4] = {fmaf(acc[m][tt][0], sc.x, sh.x), fmaf(acc[m][tt][1], sc.y, sh.y), fmaf(acc[m][tt][2], sc.z, sh.z), fmaf(acc[m][tt][3], sc.w, sh.w)};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = max2(v[r], lo_out);
        if (XP) {  // max over the column pair (2j, 2j+1); the second column is ignored when it is past the image
          const float other = __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v[r]), 0xB1 /*quad_perm:[1,0,3,2]*/, 0xf, 0xf, true));
          v[r] = max2(v[r], pair_ok ? other : v[r]);
        }
        if (EPI == 1) {
          const float lv = live ? v[r] : 0.0f;
          st[m * 4 + r] += lv;
          st[4 * MT + m * 4 + r] = fmaf(lv, lv, st[4 * MT + m * 4 + r]);
        }
      }
      const int oq = m * 4 + lk;
      const bool st_ok = live && oq < CQo;
      const int idx = XP ? ((oq * H + (row - R)) * WPx + (x >> 1)) : (oq * plane + flat);
      if (EPI == 2 || EPI == 3) {
        float4 rv4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (st_ok) rv4 = (reinterpret_cast<const float4*>(er.ref) + (int64_t)b * CQo * plane)[idx];
        const float rvv[4] = {rv4.x, rv4.y, rv4.z, rv4.w};
        if (EPI == 3) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = rvv[r] > 0.0f ? v[r] : 0.0f;
        } else {
          const float4 mu4 = reinterpret_cast<const float4*>(bn_s[0])[m * 4 + lk], in4 = reinterpret_cast<const float4*>(bn_s[1])[m * 4 + lk];
          const float4 g4 = reinterpret_cast<const float4*>(bn_s[2])[m * 4 + lk], b4 = reinterpret_cast<const float4*>(bn_s[3])[m * 4 + lk];
          const float mu[4] = {mu4.x, mu4.y, mu4.z, mu4.w}, iv[4] = {in4.x, in4.y, in4.z, in4.w}, gm[4] = {g4.x, g4.y, g4.z, g4.w}, bt[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float xh = (rvv[r] - mu[r]) * iv[r];
            const bool gate = !er.relu || fmaf(xh, gm[r], bt[r]) > 0.0f;
            const float gg = (st_ok && gate) ? v[r] : 0.0f;
            st[m * 4 + r] += gg;
            st[4 * MT + m * 4 + r] = fmaf(gg, xh, st[4 * MT + m * 4 + r]);
          }
        }
      }
      if (st_ok) outb[idx] = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
  if (STATS) {  // BatchNorm batch statistics of the tensor just written: the scheme of sepconv_tile_kernel, for any number of output tiles
#pragma unroll
    for (int j = 0; j < 8 * MT; ++j) {
      float a = st[j];
      a += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(a), 0x111 /*row_shr:1*/, 0xf, 0xf, true));
      a += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(a), 0x112 /*row_shr:2*/, 0xf, 0xf, true));
      a += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(a), 0x114 /*row_shr:4*/, 0xf, 0xf, true));
      a += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(a), 0x118 /*row_shr:8*/, 0xf, 0xf, true));
      st[j] = a;
    }
    if (lj == 15) {
#pragma unroll
      for (int j = 0; j < 8 * MT; ++j) stat_s[wave][lk][j] = st[j];
    }
    __syncthreads();  // all NWV waves (those without a window wrote zeros)
    if (wave == 0) {
      const int g = lane >> 4;
      for (int j = lane & 15; j < 8 * MT; j += 16) {  // j = [sum | sum of squares] * 4 MT + m * 4 + r -> channel quad m * 4 + g, element r
        float tot = 0.0f;
#pragma unroll
        for (int w2 = 0; w2 < NWV; ++w2) tot += stat_s[w2][g][j];
        const int isq = j / (4 * MT), m = (j / 4) % MT, r = j & 3;
        const int cq = m * 4 + g;
        if (cq < CQo) atomicAdd(&shards[(((bx + b * 7) & 31) * CQo + cq) * 8 + isq * 4 + r], (double)tot);
      }
    }
  }
}

// =========================================================================================
// pool_res_add: MaxPooling2D((3,2),2,same)(s) + Conv2D(1x1, strides 2)(prev) + bias   (architectures.py:190-196)
// Same register-tile scheme as sepconv: one wave owns 64 consecutive flat pixels of the padded OUTPUT plane
// (lane = pooled pixel).  The strided 1x1 residual convolution is an MFMA contraction: per input quad the lane
// loads the dwordx4 of source pixel (2i, 2j), the 4 VGPRs are transposed into B fragments (permlane swaps) and
// multiplied by the residual weights (A operand, row = output channel).  In the epilogue a lane holds 4
// consecutive output channels of its pixel and adds the max over the 3 x 2 pooling window, read as dwordx4.
// =========================================================================================
template <int MT>
__global__ __launch_bounds__(256) void pool_res_add_kernel(const float* __restrict__ s, const float* __restrict__ prev /*[B][CQp][HP][WP][4]*/,
                                                            int C, int Cp, int H, int W, int WP, int R, int Ho, int Wo, int WPo, int pad_top,
                                                            int pad_left, const float* __restrict__ wr /*[Cp][C]*/, const float* __restrict__ br,
                                                            float* __restrict__ out /*[B][CQ][Ho+2R][WPo][4]*/, int xpooled, int tasks, uint32_t magic_WPo,
                                                            const float* __restrict__ bn_mean, const float* __restrict__ bn_var,
                                                            const float* __restrict__ bn_gamma, const float* __restrict__ bn_beta, float bn_eps) {
  // bn_mean != NULL (training forward): s holds the PRE-BatchNorm tensor v and the pooling runs on BN(v) = fma(v, sc, sh) without
  // materialising it: fma with sc >= 0 is monotone, so max BN(v) = BN(max v) bit for bit; for sc < 0 it is BN(min v).
  const int lane = threadIdx.x & 63;
  int bx, b;
  xcd_remap(bx, b);
  const int task = bx * 4 + (threadIdx.x >> 6);
  if (task >= tasks) return;
  const int lk = lane >> 4, lj = lane & 15;
  const int CQ = (C + 3) >> 2, CQp = (Cp + 3) >> 2;
  const bool prev_compact = (xpooled & 2) != 0;  // prev = [B][CQp][Ho][Wo][4]: only the pixels (2i, 2j) the strided 1x1 conv samples
  xpooled &= 1;
  const int plane = prev_compact ? Ho * Wo : (H + 2 * R) * WP, plane_o = (Ho + 2 * R) * WPo;
  const int qbase = R * WPo + task * 64;
  const int q = qbase + lane;
  const int prow = (int)__umulhi((uint32_t)q, magic_WPo);
  const int pj = q - prow * WPo, pi = prow - R;
  const bool pvalid = pj < Wo && pi < Ho;
  const int srcpix = pvalid ? (prev_compact ? pi * Wo + pj : (2 * pi + R) * WP + 2 * pj) : 0;
  const float4* pp = reinterpret_cast<const float4*>(prev) + (int64_t)b * CQp * plane + srcpix;

  f32x4 acc[MT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float4 nxt = pp[0];
  for (int cq = 0; cq < CQp; ++cq) {
    const float4 cur = nxt;
    if (cq + 1 < CQp) nxt = pp[(int64_t)(cq + 1) * plane];
    float afrag[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int ci = cq * 4 + lk, co = m * 16 + lj;
      const bool ok = ci < Cp && co < C;
      const float av = wr[ok ? ci * C + co : 0];
      afrag[m] = ok ? av : 0.0f;
    }
    float d[4] = {cur.x, cur.y, cur.z, cur.w};
    swap32(d[0], d[2]);
    swap32(d[1], d[3]);
    swap16(d[0], d[1]);
    swap16(d[2], d[3]);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m][t] = mfma16(afrag[m], d[t], acc[m][t]);
  }

  float br_r[MT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = m * 16 + lk * 4 + r;
      br_r[m][r] = co < C ? br[co] : 0.0f;
    }
  // training forward: the folded BatchNorm of this lane's output channels, once per window (formed inside the tile macro it was four times the loads,
  // v_rsq and multiplies per channel: the channels of a lane do not change with the tile)
  float bsc[MT][4], bsh[MT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = (m * 4 + lk) * 4 + r, cc = c < C ? c : 0;
      const float sc = bn_mean ? bn_gamma[cc] * rsqrtf(bn_var[cc] + bn_eps) : 1.0f;  // as bn_planes_apply_kernel
      bsc[m][r] = sc;
      bsh[m][r] = bn_mean ? bn_beta[cc] - bn_mean[cc] * sc : 0.0f;
    }
  const int WPx = (Wo + 3) & ~3;
  // Pooling operands of one 16-pixel tile: 3 (x-pooled) or 3 x 2 values per output quad, ALL requested before the first is used, with
  // clamped coordinates (a duplicated row / column leaves a maximum and a minimum unchanged) instead of a bounds branch around every
  // load; for up to two output tiles the next tile's operands are requested before the current tile is reduced.
  constexpr int NV = 6;
  constexpr bool AHEAD = MT <= 2;
  float4 va[MT][NV], vb[MT][NV];
  // (macros, not lambdas taking the arrays by reference: those kept va / vb in scratch memory)
#define ORCAI_POOL_LOAD_TILE(t, v)                                                                              \
  {                                                                                                             \
    const int flat = qbase + 16 * (t) + lj;                                                                     \
    const int row = (int)__umulhi((uint32_t)flat, magic_WPo);                                                   \
    int j = flat - row * WPo, i = row - R;                                                                      \
    j = j < Wo ? j : Wo - 1;                                                                                    \
    i = i < Ho ? i : Ho - 1;                                                                                    \
    const int ys = 2 * i - pad_top, xs = 2 * j - pad_left;                                                      \
    _Pragma("unroll") for (int m = 0; m < MT; ++m) {                                                            \
      const int oq = m * 4 + lk < CQ ? m * 4 + lk : 0;                                                          \
      if (xpooled) {                                                                                            \
        const float4* sp = reinterpret_cast<const float4*>(s) + ((int64_t)b * CQ + oq) * (int64_t)H * WPx;      \
        _Pragma("unroll") for (int dy = 0; dy < 3; ++dy) {                                                      \
          int y = ys + dy;                                                                                      \
          y = y < 0 ? 0 : (y >= H ? H - 1 : y);                                                                 \
          v[m][2 * dy] = sp[(int64_t)y * WPx + j];                                                              \
          v[m][2 * dy + 1] = v[m][2 * dy];                                                                      \
        }                                                                                                       \
      } else {                                                                                                  \
        const float4* sp = reinterpret_cast<const float4*>(s) + ((int64_t)b * CQ + oq) * plane;                 \
        _Pragma("unroll") for (int dy = 0; dy < 3; ++dy) _Pragma("unroll") for (int dx = 0; dx < 2; ++dx) {     \
          int y = ys + dy, x = xs + dx;                                                                         \
          y = y < 0 ? 0 : (y >= H ? H - 1 : y);                                                                 \
          x = x < 0 ? 0 : (x >= W ? W - 1 : x);                                                                 \
          v[m][dy * 2 + dx] = sp[(int64_t)(y + R) * WP + x];                                                    \
        }                                                                                                       \
      }                                                                                                         \
    }                                                                                                           \
  }
#define ORCAI_POOL_REDUCE_TILE(t, v)                                                                            \
  {                                                                                                             \
    const int flat = qbase + 16 * (t) + lj;                                                                     \
    const int row = (int)__umulhi((uint32_t)flat, magic_WPo);                                                   \
    const int j = flat - row * WPo, i = row - R;                                                                \
    if (j < Wo && i < Ho) {                                                                                     \
      _Pragma("unroll") for (int m = 0; m < MT; ++m) {                                                          \
        const int oq = m * 4 + lk;                                                                              \
        if (oq < CQ) {                                                                                          \
        float mx[4] = {v[m][0].x, v[m][0].y, v[m][0].z, v[m][0].w};                                             \
        float mn[4] = {v[m][0].x, v[m][0].y, v[m][0].z, v[m][0].w};                                             \
        _Pragma("unroll") for (int e = 1; e < NV; ++e) {                                                        \
          mx[0] = fmaxf(mx[0], v[m][e].x); mx[1] = fmaxf(mx[1], v[m][e].y);                                     \
          mx[2] = fmaxf(mx[2], v[m][e].z); mx[3] = fmaxf(mx[3], v[m][e].w);                                     \
          if (bn_mean) {                                                                                        \
            mn[0] = fminf(mn[0], v[m][e].x); mn[1] = fminf(mn[1], v[m][e].y);                                   \
            mn[2] = fminf(mn[2], v[m][e].z); mn[3] = fminf(mn[3], v[m][e].w);                                   \
          }                                                                                                     \
        }                                                                                                       \
        if (bn_mean) {                                                                                          \
          _Pragma("unroll") for (int r = 0; r < 4; ++r)                                                         \
            mx[r] = fmaf(bsc[m][r] >= 0.0f ? mx[r] : mn[r], bsc[m][r], bsh[m][r]);                              \
        }                                                                                                       \
        float o[4];                                                                                             \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) o[r] = (oq * 4 + r < C) ? mx[r] + (acc[m][t][r] + br_r[m][r]) : 0.0f; \
        reinterpret_cast<float4*>(out)[((int64_t)b * CQ + oq) * plane_o + flat] = make_float4(o[0], o[1], o[2], o[3]);        \
        }                                                                                                       \
      }                                                                                                         \
    }                                                                                                           \
  }
  if (AHEAD) {
    ORCAI_POOL_LOAD_TILE(0, va)
    ORCAI_POOL_LOAD_TILE(1, vb)
    ORCAI_POOL_REDUCE_TILE(0, va)
    ORCAI_POOL_LOAD_TILE(2, va)
    ORCAI_POOL_REDUCE_TILE(1, vb)
    ORCAI_POOL_LOAD_TILE(3, vb)
    ORCAI_POOL_REDUCE_TILE(2, va)
    ORCAI_POOL_REDUCE_TILE(3, vb)
  } else {
    ORCAI_POOL_LOAD_TILE(0, va)
    ORCAI_POOL_REDUCE_TILE(0, va)
    ORCAI_POOL_LOAD_TILE(1, va)
    ORCAI_POOL_REDUCE_TILE(1, va)
    ORCAI_POOL_LOAD_TILE(2, va)
    ORCAI_POOL_REDUCE_TILE(2, va)
    ORCAI_POOL_LOAD_TILE(3, va)
    ORCAI_POOL_REDUCE_TILE(3, va)
  }
#undef ORCAI_POOL_LOAD_TILE
#undef ORCAI_POOL_REDUCE_TILE
}

// pool_res_add for the inference path (s x-pooled, no BatchNorm on the fly): the same arithmetic as pool_res_add_kernel, with
// the pooling loads made unconditional (row index clamped into the image: a duplicated row leaves a maximum unchanged) so that
// they can be issued ahead of their use -- the loads of two 16-pixel tiles are in flight while the residual 1x1 convolution's
// MFMAs run, instead of 24 load -> wait -> max round trips per wave.
// VERT: the wave's four 16-pixel tiles are STACKED -- output rows 4 g .. 4 g + 3 of one 16-column tile -- instead of 64 consecutive flat pixels.
// Consecutive output rows share one input row (2 i + 2 closes window i and opens window i + 1); with flat windows the two uses belong to
// different tiles of neighbouring waves, are requested at about the same time and BOTH miss (PMC: 1.64 GB fetched per block-1 launch for 0.99 GB
// of x-pooled input, the 3 / 2 of an unshared row; the kernel moves 5.9 TB/s of actual traffic).  Stacked, the shared row is one register set:
// 9 row loads per wave and quad instead of 12.  Same maxima, same sums: bit-identical.  Used where 16-column tiles waste few lanes (Wo >= 40).
template <int MT, bool VERT = false>
__global__ __launch_bounds__(256, MT <= 2 ? 4 : 2) void pool_res_add_x_kernel(const float* __restrict__ s /*[B][CQ][H][WPx][4]*/, const float* __restrict__ prev, int C, int Cp, int H,
                                                              int W, int WP, int R, int Ho, int Wo, int WPo, int pad_top, const float* __restrict__ wr /*[Cp][C]*/,
                                                              const float* __restrict__ br, float* __restrict__ out /*[B][CQ][Ho+2R][WPo][4]*/, int prev_compact,
                                                              int tasks, uint32_t magic_WPo, int ntc = 0 /*VERT: 16-column tiles per output row*/) {
  const int lane = threadIdx.x & 63;
  int bx, b;
  xcd_remap(bx, b);
  const int task = bx * 4 + (threadIdx.x >> 6);
  if (task >= tasks) return;
  const int lk = lane >> 4, lj = lane & 15;
  const int CQ = (C + 3) >> 2, CQp = (Cp + 3) >> 2;
  const int plane = prev_compact ? Ho * Wo : (H + 2 * R) * WP, plane_o = (Ho + 2 * R) * WPo;
  const int WPx = (Wo + 3) & ~3;
  const int qbase = R * WPo + task * 64;
  const int vrg = VERT ? task / ntc : 0, vct = VERT ? task - vrg * ntc : 0;  // VERT: row group (4 output rows), column tile

  // pooling operands: output pixel 16t + lj of the window, output quad m*4 + lk, rows 2i - pad_top + {0, 1, 2}
  constexpr int NROW = VERT ? 9 : 12;  // row sets: VERT shares the row between stacked tiles (tile t uses sets 2t, 2t + 1, 2t + 2)
  int soff[NROW], oidx[4];  // float4 index inside one quad plane of s; output pixel index (or -1)
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    int i, jj, flat;
    if (VERT) {
      i = vrg * 4 + t;
      jj = vct * 16 + lj;
      flat = (i + R) * WPo + jj;
    } else {
      flat = qbase + 16 * t + lj;
      const int row = (int)__umulhi((uint32_t)flat, magic_WPo);
      jj = flat - row * WPo;
      i = row - R;
    }
    const bool valid = jj < Wo && i < Ho;
    oidx[t] = valid ? flat : -1;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      if (VERT && t > 0 && dy == 0) continue;  // = set 2t of the tile above
      int y = 2 * i - pad_top + dy;
      y = y < 0 ? 0 : (y >= H ? H - 1 : y);
      // VERT: a lane past the last column / row still reads a pixel of the image (its column clamped): the set is shared with valid tiles
      const int jc = jj < Wo ? jj : Wo - 1;
      soff[VERT ? 2 * t + dy : 3 * t + dy] = VERT ? y * WPx + jc : (valid ? y * WPx + jj : 0);
    }
  }
  // uniform base of the snippet's planes + 32-bit byte offset per lane: one address register per load
  const char* sbase = reinterpret_cast<const char*>(reinterpret_cast<const float4*>(s) + (int64_t)b * CQ * H * WPx);
  uint32_t qoff[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int oq = m * 4 + lk;
    qoff[m] = (uint32_t)((oq < CQ ? oq : 0) * H * WPx);
  }
  float4 v[NROW][MT];
  auto load_tile = [&](int t) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        if (VERT && t > 0 && dy == 0) continue;
        const int e = VERT ? 2 * t + dy : 3 * t + dy;
        v[e][m] = *reinterpret_cast<const float4*>(sbase + (qoff[m] + (uint32_t)soff[e]) * 16u);
      }
  };
  float br_r[MT][4];  // residual bias of this lane's output channels
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = m * 16 + lk * 4 + r;
      const float bv = br[co < C ? co : 0];
      br_r[m][r] = co < C ? bv : 0.0f;
    }
  constexpr int DEPTH = MT <= 2 ? 2 : 1;  // 16-pixel tiles of pooling operands in flight (12 * MT VGPRs each)
  load_tile(0);
  if (DEPTH > 1) load_tile(1);
  __builtin_amdgcn_sched_barrier(0);

  // residual branch: Conv2D(C, 1, strides 2)(prev) at this window's 64 output pixels (lane = pixel: lane 16 t + lj is pixel lj of tile t)
  int pj, pi;
  if (VERT) {
    pi = vrg * 4 + lk;
    pj = vct * 16 + lj;
  } else {
    const int q = qbase + lane;
    const int prow = (int)__umulhi((uint32_t)q, magic_WPo);
    pj = q - prow * WPo;
    pi = prow - R;
  }
  const bool pvalid = pj < Wo && pi < Ho;
  const int srcpix = pvalid ? (prev_compact ? pi * Wo + pj : (2 * pi + R) * WP + 2 * pj) : 0;
  const float4* pp = reinterpret_cast<const float4*>(prev) + (int64_t)b * CQp * plane + srcpix;
  f32x4 acc[MT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float4 nxt = pp[0];
  for (int cq = 0; cq < CQp; ++cq) {
    const float4 cur = nxt;
    if (cq + 1 < CQp) nxt = pp[(int64_t)(cq + 1) * plane];
    float afrag[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int ci = cq * 4 + lk, co = m * 16 + lj;
      const bool ok = ci < Cp && co < C;
      const float av = wr[ok ? ci * C + co : 0];
      afrag[m] = ok ? av : 0.0f;
    }
    float d[4] = {cur.x, cur.y, cur.z, cur.w};
    swap32(d[0], d[2]);
    swap32(d[1], d[3]);
    swap16(d[0], d[1]);
    swap16(d[2], d[3]);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m][t] = mfma16(afrag[m], d[t], acc[m][t]);
  }

#pragma unroll
  for (int t = 0; t < 4; ++t) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int oq = m * 4 + lk;
      const int e0 = VERT ? 2 * t : 3 * t;
      float mx[4] = {v[e0][m].x, v[e0][m].y, v[e0][m].z, v[e0][m].w};
#pragma unroll
      for (int dy = 1; dy < 3; ++dy) {
        mx[0] = fmaxf(mx[0], v[e0 + dy][m].x); mx[1] = fmaxf(mx[1], v[e0 + dy][m].y);
        mx[2] = fmaxf(mx[2], v[e0 + dy][m].z); mx[3] = fmaxf(mx[3], v[e0 + dy][m].w);
      }
      if (oidx[t] >= 0 && oq < CQ) {
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (oq * 4 + r < C) ? mx[r] + (acc[m][t][r] + br_r[m][r]) : 0.0f;
        reinterpret_cast<float4*>(out)[((int64_t)b * CQ + oq) * plane_o + oidx[t]] = make_float4(o[0], o[1], o[2], o[3]);
      }
    }
    if (t + DEPTH < 4) {
      load_tile(t + DEPTH);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// =========================================================================================
// gemm: C[M][N] = act(A[M][K] * Bm[K][N] + bias[N]) [* scale[N] + shift[N]]     (LSTM input projections, Dense-128)
// 128 x 128 block tile, BK = 16, 4 waves as 2 x 2, wave tile 64 x 64 (4 x 4 MFMA 16x16x4 tiles).
// MFMA row = M index, MFMA column = N index.
// =========================================================================================
constexpr int GP = 128 + 16;  // LDS pitch: rows k and k+1 16 banks apart

constexpr int GBK = 32;

// 128 x 128 x 32 tiles; the next tile's global loads (16 bytes per lane where alignment allows) are in flight in registers
// while the MFMAs of the current tile run; A is transposed into LDS as [k][m].
__global__ __launch_bounds__(256) void gemm_kernel(const float* __restrict__ A, const float* __restrict__ Bm, const float* __restrict__ bias,
                                                    const float* __restrict__ scale, const float* __restrict__ shift, float* __restrict__ C,
                                                    int M, int N, int K, int act, int vec) {
  __shared__ __attribute__((aligned(16))) float As[GBK][GP];
  __shared__ __attribute__((aligned(16))) float Bs[GBK][GP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lk = lane >> 4, lj = lane & 15;
  const int64_t m0 = (int64_t)blockIdx.y * 128;
  const int n0 = blockIdx.x * 128;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float4 ra[4], rb[4];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int f = tid + 256 * g;
      {  // A: row m = f & 127, k group (f >> 7) * 4
        const int64_t m = m0 + (f & 127);
        const int k = k0 + (f >> 7) * 4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (m < M) {
          const float* p = A + m * K + k;
          if (vec && k + 3 < K) {
            const float4 t = *reinterpret_cast<const float4*>(p);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (k + e < K) ? p[e] : 0.0f;
          }
        }
        ra[g] = make_float4(v[0], v[1], v[2], v[3]);
      }
      {  // B: k row f >> 5, n group (f & 31) * 4
        const int k = k0 + (f >> 5), n = n0 + (f & 31) * 4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (k < K) {
          const float* p = Bm + (int64_t)k * N + n;
          if (vec && n + 3 < N) {
            const float4 t = *reinterpret_cast<const float4*>(p);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (n + e < N) ? p[e] : 0.0f;
          }
        }
        rb[g] = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < K; k0 += GBK) {
    __syncthreads();  // previous tile consumed
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int f = tid + 256 * g;
      const int m = f & 127, k = (f >> 7) * 4;
      As[k][m] = ra[g].x; As[k + 1][m] = ra[g].y; As[k + 2][m] = ra[g].z; As[k + 3][m] = ra[g].w;
      *reinterpret_cast<float4*>(&Bs[f >> 5][(f & 31) * 4]) = rb[g];
    }
    __syncthreads();
    if (k0 + GBK < K) fetch(k0 + GBK);
#pragma unroll
    for (int kk = 0; kk < GBK / 4; ++kk) {
      float a[4], bq[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = As[kk * 4 + lk][wm * 64 + i * 16 + lj];
#pragma unroll
      for (int j = 0; j < 4; ++j) bq[j] = Bs[kk * 4 + lk][wn * 64 + j * 16 + lj];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(a[i], bq[j], acc[i][j]);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t m = m0 + wm * 64 + i * 16 + lk * 4 + r;
      if (m >= M) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn * 64 + j * 16 + lj;
        if (n >= N) continue;
        float v = acc[i][j][r] + (bias ? bias[n] : 0.0f);
        if (act == 1) v = fmaxf(v, 0.0f);
        if (scale) v = fmaf(v, scale[n], shift[n]);
        C[m * N + n] = v;
      }
    }
}

// =========================================================================================
// lstm_recurrent: one direction of one Bidirectional(LSTM(u)) layer for a tile of 16 snippets.
// xz = x*W + b is precomputed by gemm_kernel with PERMUTED gate columns: column p = 32*w + 16*nt + j holds
//   unit 8*w + (j&7), gate 2*nt + (j>>3)  (gate order i,f,g,o), so wave w owns all four gates of units
//   [8w, 8w+8).  The recurrent kernel U (u x 4u, same column permutation) stays in registers for all T steps
//   (u/4 k-steps x 2 column tiles = 64 VGPRs at u = 128); h lives in LDS, c in registers.
// =========================================================================================
// v_rcp_f32 (1 ulp) instead of an IEEE division: the quotient cost ten instructions per gate value (v_div_scale / v_div_fmas / v_div_fixup around the
// same v_rcp) in a step whose SIMDs issue 96 % of the time, next to a fast exponential that is less accurate than either
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return fmaf(-2.0f, __builtin_amdgcn_rcpf(__expf(2.0f * x) + 1.0f), 1.0f); }

// The gate stage of a recurrence step (the training kernels' lstm_gate_stage, train_head.hip, without the gate store): every lane activates its OWN eight
// pre-activations -- tile 0 column lj = gate i (lj < 8) or f, tile 1 = g or o, rows 4 lk + r -- then the half-rows trade what the other needs (four DPP row
// rotations) and each finishes two (row, unit) cells: 10 exponential / reciprocal pairs per lane and step instead of 20, the same functions of the same numbers.
__device__ __forceinline__ float dpp_xor8(float v) {  // lane l <- lane l ^ 8
  return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x128 /*row_ror:8*/, 0xf, 0xf, true));
}
template <class StoreState>
__device__ __forceinline__ void lstm_gate_stage(const float (&z0)[4], const float (&z1)[4], float (&cst)[4], int lj, StoreState store_state) {
  const bool low = lj < 8;
  const float m1 = low ? 2.0f : -1.0f;  // tile 1: tanh (g) on the low half-row, sigmoid (o) on the high one
  float a0[4], a1[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    a0[r] = sigmoidf_(z0[r]);
    const float rc = __builtin_amdgcn_rcpf(__expf(m1 * z1[r]) + 1.0f);
    a1[r] = low ? fmaf(-2.0f, rc, 1.0f) : rc;  // tanhf_ / sigmoidf_ to the bit
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const float got0 = dpp_xor8(low ? a0[2 + j] : a0[j]), got1 = dpp_xor8(low ? a1[2 + j] : a1[j]);
    const float gi = low ? a0[j] : got0, gf = low ? got0 : a0[2 + j];
    const float gg = low ? a1[j] : got1, go = low ? got1 : a1[2 + j];
    const float c = gf * cst[j] + gi * gg;
    const float h = go * tanhf_(c);
    cst[j] = c;
    store_state(low ? j : 2 + j, h);
  }
}

template <int U>
__global__ __launch_bounds__(U * 8) void lstm_kernel(const float* __restrict__ xz /*[B][T][2][4U] permuted*/, const float* __restrict__ Uw /*[2][U][4U] permuted*/,
                                                      int B, int T, float* __restrict__ out /*[B][T][2U]*/) {
  constexpr int KSTEPS = U / 4, HP = U + 2;
  __shared__ float hbuf[2][16][HP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lk = lane >> 4, lj = lane & 15;
  const int dir = blockIdx.y;
  const int b0 = blockIdx.x * 16;
  const float* Ud = Uw + (int64_t)dir * U * 4 * U;

  float ufrag[2][KSTEPS];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int kk = 0; kk < KSTEPS; ++kk) ufrag[nt][kk] = Ud[(int64_t)(kk * 4 + lk) * (4 * U) + wave * 32 + nt * 16 + lj];

  for (int i = tid; i < 2 * 16 * HP; i += U * 8) (&hbuf[0][0][0])[i] = 0.0f;
  float cst[4] = {0.f, 0.f, 0.f, 0.f};
  __syncthreads();

  const int unit = wave * 8 + (lj & 7);
  f32x4 xz_next[2];
  auto load_xz = [&](int tt) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int bb = b0 + lk * 4 + r;
        xz_next[nt][r] = (bb < B) ? xz[(((int64_t)bb * T + tt) * 2 + dir) * (4 * U) + wave * 32 + nt * 16 + lj] : 0.0f;
      }
  };
  load_xz(dir ? T - 1 : 0);
  for (int step = 0; step < T; ++step) {
    const int t = dir ? (T - 1 - step) : step;
    const int cur = step & 1;
    f32x4 acc[2] = {xz_next[0], xz_next[1]};
    if (step + 1 < T) load_xz(dir ? (T - 2 - step) : step + 1);  // the next step's input projection is in flight during this step's recurrence
#pragma unroll
    for (int kk = 0; kk < KSTEPS; ++kk) {
      const float a = hbuf[cur][lj][kk * 4 + lk];  // A[i = batch][k]
      acc[0] = mfma16(a, ufrag[0][kk], acc[0]);
      acc[1] = mfma16(a, ufrag[1][kk], acc[1]);
    }
    // lane j<8 holds (i, g), lane j+8 holds (f, o) of the same unit: swap across the pair
    {
      const float z0[4] = {acc[0][0], acc[0][1], acc[0][2], acc[0][3]}, z1[4] = {acc[1][0], acc[1][1], acc[1][2], acc[1][3]};
      lstm_gate_stage(z0, z1, cst, lj, [&](int r, float h) {
        const int row = lk * 4 + r;
        hbuf[cur ^ 1][row][unit] = h;
        const int bb = b0 + row;
        if (bb < B) out[((int64_t)bb * T + t) * (2 * U) + dir * U + unit] = h;
      });
    }
    __syncthreads();
  }
}

// The inference recurrence on split-f16 MFMA at f32 accuracy (train_head.hip, lstm_train_fwd_split_kernel, explains the scheme: every operand
// as hi + lo / 4096 in two normal f16 numbers, hi*hi in one accumulator, hi*lo + lo*hi in a second one scaled back by 2^-12): a step is bound
// by one compute unit's matrix rate, and the f32-input MFMA runs at 1/16 of the f16 rate.
typedef _Float16 mh16;
typedef mh16 mh16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split_f16_m(float x, mh16& hi, mh16& lo) {
  const mh16 h = fabsf(x) >= 6.103515625e-05f ? (mh16)x : (mh16)0.0f;
  hi = h;
  lo = (mh16)((x - (float)h) * 4096.0f);
}

template <int U>
__global__ __launch_bounds__(U * 8) void lstm_split_kernel(const float* __restrict__ xz /*[B][T][2][4U] permuted*/, const float* __restrict__ Uw /*[2][U][4U] permuted*/,
                                                            int B, int T, float* __restrict__ out /*[B][T][2U]*/) {
  constexpr int KB = U / 32, HPh = U + 8;
  constexpr float LO_INV = 1.0f / 4096.0f;
  __shared__ __attribute__((aligned(16))) mh16 hhi[2][16][HPh];
  __shared__ __attribute__((aligned(16))) mh16 hlo[2][16][HPh];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lk = lane >> 4, lj = lane & 15;
  const int dir = blockIdx.y;
  const int b0 = blockIdx.x * 16;
  const float* Ud = Uw + (int64_t)dir * U * 4 * U;
  mh16x8 uhi[2][KB], ulo[2][KB];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        mh16 hi, lo;
        split_f16_m(Ud[(int64_t)(kb * 32 + lk * 8 + e) * (4 * U) + wave * 32 + nt * 16 + lj], hi, lo);
        uhi[nt][kb][e] = hi;
        ulo[nt][kb][e] = lo;
      }
  for (int i = tid; i < 2 * 16 * HPh; i += U * 8) { (&hhi[0][0][0])[i] = (mh16)0.0f; (&hlo[0][0][0])[i] = (mh16)0.0f; }
  float cst[4] = {0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  const int unit = wave * 8 + (lj & 7);
  f32x4 xz_next[2];
  auto load_xz = [&](int tt) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int bb = b0 + lk * 4 + r;
        xz_next[nt][r] = (bb < B) ? xz[(((int64_t)bb * T + tt) * 2 + dir) * (4 * U) + wave * 32 + nt * 16 + lj] : 0.0f;
      }
  };
  load_xz(dir ? T - 1 : 0);
  for (int step = 0; step < T; ++step) {
    const int t = dir ? (T - 1 - step) : step;
    const int cur = step & 1;
    f32x4 acc[2] = {xz_next[0], xz_next[1]};
    f32x4 acl[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    if (step + 1 < T) load_xz(dir ? (T - 2 - step) : step + 1);
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const mh16x8 ah = *reinterpret_cast<const mh16x8*>(&hhi[cur][lj][kb * 32 + lk * 8]);
      const mh16x8 al = *reinterpret_cast<const mh16x8*>(&hlo[cur][lj][kb * 32 + lk * 8]);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, uhi[nt][kb], acc[nt], 0, 0, 0);
        acl[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, ulo[nt][kb], acl[nt], 0, 0, 0);
        acl[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, uhi[nt][kb], acl[nt], 0, 0, 0);
      }
    }
    {
      float z0[4], z1[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        z0[r] = fmaf(acl[0][r], LO_INV, acc[0][r]);
        z1[r] = fmaf(acl[1][r], LO_INV, acc[1][r]);
      }
      lstm_gate_stage(z0, z1, cst, lj, [&](int r, float h) {
        const int row = lk * 4 + r;
        mh16 hh, hl;
        split_f16_m(h, hh, hl);
        hhi[cur ^ 1][row][unit] = hh;
        hlo[cur ^ 1][row][unit] = hl;
        const int bb = b0 + row;
        if (bb < B) out[((int64_t)bb * T + t) * (2 * U) + dir * U + unit] = h;
      });
    }
    __syncthreads();
  }
}

// =========================================================================================
// dense_sigmoid: out[m][n] = sigmoid(x[m][:] . w[:][n] + b[n]),  N <= 8   (architectures.py:239)
// =========================================================================================
__global__ __launch_bounds__(256) void dense_sigmoid_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                             int64_t M, int K, int N, float* __restrict__ out) {
  const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  float acc[8];
#pragma unroll
  for (int n = 0; n < 8; ++n) acc[n] = 0.0f;
  const float* xr = x + m * K;
  for (int k = 0; k < K; ++k) {
    const float v = xr[k];
#pragma unroll
    for (int n = 0; n < 8; ++n)
      if (n < N) acc[n] = fmaf(v, w[k * N + n], acc[n]);
  }
#pragma unroll
  for (int n = 0; n < 8; ++n)
    if (n < N) out[m * N + n] = 1.0f / (1.0f + expf(-(acc[n] + bias[n])));
}

// The same with eight lanes per row (K a multiple of 4): lane j of a row reads the row's float4 number j, j + 8, ... -- a row's 128-byte segments,
// coalesced -- and its 4 N weights for them; the eight partial sums meet by three lane exchanges.  (One thread per row reads 4 bytes a row apart and
// walks K dependent steps: 59 us for the 2 944 x 128 input of a training step, 12 workgroups.)
__global__ __launch_bounds__(256) void dense_sigmoid_rows_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                                  int64_t M, int K, int N, float* __restrict__ out) {
  const int j = threadIdx.x & 7;
  const int64_t m = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3);
  const bool live = m < M;
  float acc[8];
#pragma unroll
  for (int n = 0; n < 8; ++n) acc[n] = 0.0f;
  const float4* xr = reinterpret_cast<const float4*>(x + (live ? m : 0) * K);
  for (int q = j; q < (K >> 2); q += 8) {
    const float4 v = xr[q];
    const float* wq = w + (int64_t)q * 4 * N;
#pragma unroll
    for (int n = 0; n < 8; ++n)
      if (n < N) acc[n] = fmaf(v.w, wq[3 * N + n], fmaf(v.z, wq[2 * N + n], fmaf(v.y, wq[N + n], fmaf(v.x, wq[n], acc[n]))));
  }
#pragma unroll
  for (int n = 0; n < 8; ++n) {
    acc[n] += __shfl_xor(acc[n], 1);
    acc[n] += __shfl_xor(acc[n], 2);
    acc[n] += __shfl_xor(acc[n], 4);
  }
  if (live && j < N) {
    float a = acc[0];
#pragma unroll
    for (int n = 1; n < 8; ++n) a = j == n ? acc[n] : a;
    out[m * N + j] = 1.0f / (1.0f + expf(-(a + bias[j])));
  }
}

// =========================================================================================
// overlap_average: predict.py:276-293.  Output step s is covered by snippets i with step*i <= s < step*i + P.
// float64 accumulate in snippet order, then divide by the count (bit-exact with the numpy loop).
// =========================================================================================
__global__ __launch_bounds__(256) void overlap_average_kernel(const float* __restrict__ pred /*[n][P][L]*/, int n, int P, int L, int step, int64_t S,
                                                               double* __restrict__ agg /*[S][L]*/, double* __restrict__ cnt /*[S]*/) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= S * L) return;
  const int64_t s = idx / L;
  const int l = (int)(idx % L);
  // snippets i with i*step <= s and s - i*step < P
  int64_t i_hi = s / step;
  if (i_hi > n - 1) i_hi = n - 1;
  int64_t i_lo = (s - P + step) / step;  // ceil((s - P + 1)/step)
  if (s - P + 1 <= 0) i_lo = 0;
  double sum = 0.0;
  int c = 0;
  for (int64_t i = i_lo; i <= i_hi; ++i) {
    const int64_t off = s - i * step;
    if (off >= 0 && off < P) {
      sum += (double)pred[(i * P + off) * L + l];
      ++c;
    }
  }
  if (c > 0) sum /= (double)c;
  agg[idx] = sum;
  if (l == 0) cnt[s] = (double)c;
}

inline uint32_t magic_for(uint32_t d) { return (uint32_t)((0x100000000ull + d - 1) / d); }  // __umulhi(n, magic) == n / d while n*d < 2^32

// ---------------------------------------------------------------- ResNet1DConv head (architectures.py:10-15, 107-115)
// ReduceFrequencyMean on the Keras Reshape layout: out[m][c] = mean over x of feat[m][x*C + c]
__global__ __launch_bounds__(256) void freq_mean_kernel(const float* __restrict__ feat, int64_t M, int W, int C, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= M * C) return;
  const int64_t m = i / C;
  const int c = (int)(i - m * C);
  const float* src = feat + m * (int64_t)W * C + c;
  float s = 0.0f;
  for (int x = 0; x < W; ++x) s += src[(int64_t)x * C];
  out[i] = s / (float)W;
}

// Conv1D over time, "same" padding ((K-1)/2 left, K/2 right), sigmoid: one workgroup per snippet, x[T][C] staged in LDS,
// thread (t, l) accumulates K*C products in the order (k, c).
__global__ __launch_bounds__(512) void conv1d_sigmoid_kernel(const float* __restrict__ x /*[B][T][C]*/, const float* __restrict__ w /*[K][C][L]*/,
                                                             const float* __restrict__ bias, int T, int C, int K, int L, float* __restrict__ out /*[B][T][L]*/) {
  extern __shared__ float xs[];  // [T][C]
  const int b = blockIdx.x;
  for (int i = threadIdx.x; i < T * C; i += blockDim.x) xs[i] = x[(int64_t)b * T * C + i];
  __syncthreads();
  const int left = (K - 1) / 2;
  for (int o = threadIdx.x; o < T * L; o += blockDim.x) {
    const int t = o / L, l = o - t * L;
    float acc = bias[l];
    for (int k = 0; k < K; ++k) {
      const int tt = t + k - left;
      if (tt < 0 || tt >= T) continue;
      const float* xr = xs + tt * C;
      const float* wr = w + (int64_t)k * C * L + l;
      for (int c = 0; c < C; ++c) acc = fmaf(xr[c], wr[(int64_t)c * L], acc);
    }
    out[((int64_t)b * T + t) * L + l] = 1.0f / (1.0f + __expf(-acc));
  }
}

// EPI 2 leaves 32 accumulator copies [32][CQ][8] (sum g | sum g * xhat per channel quad); the BatchNorm backward kernels read
// scratch2C = dbeta[4 CQ] | dgamma[4 CQ] (doubles): summed and compacted in place by one workgroup (all reads before the first write).
__global__ __launch_bounds__(256) void bn_bwd_sums_compact_kernel(double* __restrict__ shards, int CQ) {
  const int t = threadIdx.x;  // output element: t < 4 CQ -> dbeta[t], else dgamma[t - 4 CQ]
  double tot = 0.0;
  if (t < 8 * CQ) {
    const int which = t >= 4 * CQ, c = which ? t - 4 * CQ : t;
    for (int sh = 0; sh < 32; ++sh) tot += shards[((int64_t)sh * CQ + (c >> 2)) * 8 + which * 4 + (c & 3)];
  }
  __syncthreads();
  if (t < 8 * CQ) shards[t] = tot;
}

struct SepArgs {
  const float *in, *dw, *pw, *scale, *shift;
  float* out;
  int B, Cin, H, W, WP, RP, Cout, relu_in, relu_out, out_layout, H2, WP2;
  float* u_out = nullptr;
  double* shards = nullptr;  // strip tiles with the depthwise-output store: BatchNorm statistics of the output in the epilogue
  int epi = 0;               // 2 / 3: the epilogue extras of the input-gradient passes (EpiRef er), no depthwise-output store
  EpiRef er;
  InBn ib;                   // ib.mean != nullptr: BatchNorm + ReLU of the input applied on load (training forward with the statistics epilogue)
};

int g_pool_vert = 1;       // inference pooling kernel on stacked tiles where the plane is wide enough (orcai_pool_vertical: A/B)
int g_entry_windows = 4;   // windows per wave of conv0_sep_kernel
int g_entry_tile = 10;     // waves per workgroup of conv0_sep_tile_kernel (10 or 16); 0 = conv0_sep_kernel everywhere
int g_tile_mode = 1;  // k = 3 launches with plane / x-pooled output: 1 = sepconv_tile_kernel for wide planes with two output tiles (<= 8 input quads,
                      // >= 2 strips that cover the width with <= 15 % waste) and sepconv_ftile_kernel otherwise; 2 = sepconv_ftile_kernel for
                      // all of them; 0 = sepconv_kernel everywhere (the reference the bit-identity tests compare with)

template <int MT, int CQ>
int launch_sepconv_tile(hipStream_t st, const SepArgs& a, int nstrip) {
  constexpr int TR = 8;
  dim3 grid(nstrip * ((a.H + TR - 1) / TR), a.B);
#define ORCAI_TILE_LAUNCH(XP, RELU, UOUT)                                                                                                       \
  hipLaunchKernelGGL((sepconv_tile_kernel<MT, CQ, XP, RELU, TR, UOUT>), grid, dim3(64 * TR), 0, st, a.in, a.Cin, a.H, a.W, a.WP, a.dw, a.pw, a.scale, \
                     a.shift, a.Cout, a.relu_out, a.out, nstrip, a.u_out)
  if (a.epi == 2) {  // input-gradient pass with the BatchNorm backward sums of its output (no ReLU on load, no depthwise-output store)
    if constexpr (MT == 2)
      hipLaunchKernelGGL((sepconv_tile_kernel<MT, CQ, false, false, TR, false, 2>), grid, dim3(64 * TR), 0, st, a.in, a.Cin, a.H, a.W, a.WP, a.dw, a.pw, a.scale, a.shift,
                         a.Cout, a.relu_out, a.out, nstrip, a.u_out, a.shards, a.er);
  } else if (a.epi) {
    return -1;  // EPI 3 lives in the flat-tile kernel only
  } else if (a.out_layout == 2) {
    if (a.relu_in) ORCAI_TILE_LAUNCH(true, true, false); else ORCAI_TILE_LAUNCH(true, false, false);
  } else if (a.u_out && a.shards) {
    if constexpr (MT == 2) {
      if (a.ib.mean)
        hipLaunchKernelGGL((sepconv_tile_kernel<MT, CQ, false, false, TR, true, 1, true>), grid, dim3(64 * TR), 0, st, a.in, a.Cin, a.H, a.W, a.WP, a.dw, a.pw, a.scale, a.shift,
                           a.Cout, a.relu_out, a.out, nstrip, a.u_out, a.shards, EpiRef{}, a.ib);
      else if (a.relu_in)
        hipLaunchKernelGGL((sepconv_tile_kernel<MT, CQ, false, true, TR, true, 1>), grid, dim3(64 * TR), 0, st, a.in, a.Cin, a.H, a.W, a.WP, a.dw, a.pw, a.scale, a.shift,
                           a.Cout, a.relu_out, a.out, nstrip, a.u_out, a.shards);
      else
        hipLaunchKernelGGL((sepconv_tile_kernel<MT, CQ, false, false, TR, true, 1>), grid, dim3(64 * TR), 0, st, a.in, a.Cin, a.H, a.W, a.WP, a.dw, a.pw, a.scale, a.shift,
                           a.Cout, a.relu_out, a.out, nstrip, a.u_out, a.shards);
    }
  } else if (a.u_out) {
    if (a.relu_in) ORCAI_TILE_LAUNCH(false, true, true); else ORCAI_TILE_LAUNCH(false, false, true);
  } else {
    if (a.relu_in) ORCAI_TILE_LAUNCH(false, true, false); else ORCAI_TILE_LAUNCH(false, false, false);
  }
#undef ORCAI_TILE_LAUNCH
  return (int)hipGetLastError();
}

template <int MT>
int launch_sepconv_ftile(hipStream_t st, const SepArgs& a, int tasks) {
  constexpr int NWV = 8;
  const int lo = a.out_layout == 2 ? 2 : 1, VAL = 64 - 2 * lo;
  const int nchunk = ((NWV - 1) * VAL + 64 + 2 * a.WP + 63) / 64;
  const int CQr = (a.Cin + 3) / 4;
  const size_t lds = (size_t)(2 * nchunk * 256 + CQr * 64 * MT + 2 * MT * 16) * sizeof(float);
  if (nchunk > 3 * NWV || lds > 64 * 1024) return -1;  // planes wider than ~500 pixels: not this kernel's shape, the caller falls back
  dim3 grid((tasks + NWV - 1) / NWV, a.B);
#define ORCAI_FTILE_LAUNCH(XP, RELU, UOUT)                                                                                                       \
  hipLaunchKernelGGL((sepconv_ftile_kernel<MT, XP, RELU, UOUT, NWV>), grid, dim3(64 * NWV), lds, st, a.in, a.Cin, a.H, a.W, a.WP, a.dw, a.pw, a.scale, \
                     a.shift, a.Cout, a.relu_out, a.out, tasks, magic_for(a.WP), nchunk, a.u_out)
  if (a.epi == 2) {
    if (lds + sizeof(float) * (NWV * 4 * 8 * MT + 4 * MT * 16) > 64 * 1024) return -1;  // + the kernel's static statistics slots and BatchNorm constants
    hipLaunchKernelGGL((sepconv_ftile_kernel<MT, false, false, false, NWV, 2>), grid, dim3(64 * NWV), lds, st, a.in, a.Cin, a.H, a.W, a.WP, a.dw, a.pw, a.scale, a.shift,
                       a.Cout, a.relu_out, a.out, tasks, magic_for(a.WP), nchunk, a.u_out, a.shards, a.er);
  } else if (a.epi == 3) {
    hipLaunchKernelGGL((sepconv_ftile_kernel<MT, false, false, false, NWV, 3>), grid, dim3(64 * NWV), lds, st, a.in, a.Cin, a.H, a.W, a.WP, a.dw, a.pw, a.scale, a.shift,
                       a.Cout, a.relu_out, a.out, tasks, magic_for(a.WP), nchunk, a.u_out, a.shards, a.er);
  } else if (a.out_layout == 2) {
    if (a.relu_in) ORCAI_FTILE_LAUNCH(true, true, false); else ORCAI_FTILE_LAUNCH(true, false, false);
  } else if (a.u_out && a.shards) {
    if (lds + sizeof(float) * (NWV * 4 * 8 * MT + (a.ib.mean ? 128 : 0)) > 64 * 1024) return -1;  // + the kernel's static statistics slots (and folded input BatchNorm)
    if (a.ib.mean)
      hipLaunchKernelGGL((sepconv_ftile_kernel<MT, false, false, true, NWV, 1, true>), grid, dim3(64 * NWV), lds, st, a.in, a.Cin, a.H, a.W, a.WP, a.dw, a.pw, a.scale, a.shift,
                         a.Cout, a.relu_out, a.out, tasks, magic_for(a.WP), nchunk, a.u_out, a.shards, EpiRef{}, a.ib);
    else if (a.relu_in)
      hipLaunchKernelGGL((sepconv_ftile_kernel<MT, false, true, true, NWV, 1>), grid, dim3(64 * NWV), lds, st, a.in, a.Cin, a.H, a.W, a.WP, a.dw, a.pw, a.scale, a.shift,
                         a.Cout, a.relu_out, a.out, tasks, magic_for(a.WP), nchunk, a.u_out, a.shards);
    else
      hipLaunchKernelGGL((sepconv_ftile_kernel<MT, false, false, true, NWV, 1>), grid, dim3(64 * NWV), lds, st, a.in, a.Cin, a.H, a.W, a.WP, a.dw, a.pw, a.scale, a.shift,
                         a.Cout, a.relu_out, a.out, tasks, magic_for(a.WP), nchunk, a.u_out, a.shards);
  } else if (a.u_out) {
    if (a.relu_in) ORCAI_FTILE_LAUNCH(false, true, true); else ORCAI_FTILE_LAUNCH(false, false, true);
  } else {
    if (a.relu_in) ORCAI_FTILE_LAUNCH(false, true, false); else ORCAI_FTILE_LAUNCH(false, false, false);
  }
#undef ORCAI_FTILE_LAUNCH
  return (int)hipGetLastError();
}

template <int KS, int MT>
int launch_sepconv_impl(hipStream_t st, const SepArgs& a) {
  const int lo = (a.out_layout == 2) ? ((KS / 2 + 1) & ~1) : KS / 2;  // x-pooled output: windows start on an even pixel
  const int VAL = 64 - 2 * lo;
  const int tasks = (a.H * a.WP + VAL - 1) / VAL;  // 64-pixel windows covering the H image rows of a plane
  if ((int64_t)(a.H + 2 * a.RP) * a.WP >= (1ll << 29)) return ORCAI_E_UNSUPPORTED;
  if constexpr (KS == 3) {
    // k = 3, plane or x-pooled output (optionally with the depthwise-output store of the training forward): the LDS-shared-row kernels.
    // Wide planes with two output tiles (orcai-V1 block 1) take the 2-D strip tiles, which fetch 10 rows per 8 windows; everything else
    // the flat-range tiles (b1/sep_b: strip 3.3 ms, flat 3.45 ms per 611 snippets; the one-window kernels 3.8 ms).
    const int CQ = (a.Cin + 3) / 4, CQo = (a.Cout + 3) / 4;
    if (g_tile_mode && a.RP == 1 && (a.out_layout == 0 || (a.out_layout == 2 && !a.u_out)) && (int64_t)CQo * (a.H + 2) * a.WP < (1ll << 27) &&
        (int64_t)CQ * (a.H + 2) * a.WP < (1ll << 27)) {
      if constexpr (MT == 2) {
        const int VALt = a.out_layout == 2 ? 60 : 62, nstrip = (a.W + VALt - 1) / VALt;
        if (g_tile_mode == 1 && a.epi != 3 && CQ <= 8 && nstrip >= 2 && a.W * 100 >= nstrip * VALt * 85)
          return CQ <= 4 ? launch_sepconv_tile<2, 4>(st, a, nstrip) : launch_sepconv_tile<2, 8>(st, a, nstrip);
      }
      const int rc = launch_sepconv_ftile<MT>(st, a, tasks);
      if (rc >= 0) return rc;
    }
  }
  if (a.shards || a.epi) return ORCAI_E_UNSUPPORTED;  // the epilogue extras exist in the LDS-tile kernels only
  dim3 grid((tasks + 3) / 4, a.B);
  hipLaunchKernelGGL((sepconv_kernel<KS, MT>), grid, dim3(256), 0, st, a.in, a.Cin, a.H, a.W, a.WP, a.relu_in, a.dw, a.pw, a.scale, a.shift, a.Cout, a.relu_out,
                     a.out_layout, a.out, tasks, magic_for(a.WP), lo, a.RP, a.H2, a.WP2, a.u_out);
  return (int)hipGetLastError();
}

template <int KS>
int launch_sepconv(hipStream_t st, const SepArgs& a) {
  switch ((a.Cout + 15) / 16) {
    case 1: return launch_sepconv_impl<KS, 1>(st, a);
    case 2: return launch_sepconv_impl<KS, 2>(st, a);
    case 3: return launch_sepconv_impl<KS, 3>(st, a);
    case 4: return launch_sepconv_impl<KS, 4>(st, a);
    default: return ORCAI_E_UNSUPPORTED;
  }
}

}  // namespace

extern "C" {

int orcai_freq_mean(const float* feat, int64_t M, int W, int C, float* out, void* stream) {
  if (!feat || !out || M <= 0 || W <= 0 || C <= 0) return ORCAI_E_BADARG;
  hipLaunchKernelGGL(freq_mean_kernel, dim3((unsigned)((M * C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, feat, M, W, C, out);
  return (int)hipGetLastError();
}

int orcai_conv1d_sigmoid(const float* x, const float* w, const float* bias, int B, int T, int C, int K, int L, float* out, void* stream) {
  if (!x || !w || !bias || !out || B <= 0 || T <= 0 || C <= 0 || K <= 0 || L <= 0) return ORCAI_E_BADARG;
  const size_t lds = (size_t)T * C * sizeof(float);
  if (lds > 64 * 1024) return ORCAI_E_UNSUPPORTED;
  hipLaunchKernelGGL(conv1d_sigmoid_kernel, dim3(B), dim3(512), lds, (hipStream_t)stream, x, w, bias, T, C, K, L, out);
  return (int)hipGetLastError();
}

int orcai_padded_width(int W, int ksize) { return (W + ksize / 2 + 3) & ~3; }

int orcai_entry_tile(int waves) {
  const int prev = g_entry_tile;
  if (waves == 0 || waves == 10 || waves == 16) g_entry_tile = waves;
  return prev;
}

int orcai_entry_windows(int windows_per_wave) {
  const int prev = g_entry_windows;
  if (windows_per_wave >= 1 && windows_per_wave <= 64) g_entry_windows = windows_per_wave;
  return prev;
}

int orcai_sepconv_tile_mode(int mode) {
  const int prev = g_tile_mode;
  if (mode >= 0 && mode <= 2) g_tile_mode = mode;
  return prev;
}

int orcai_conv0_bn_relu(const float* in, int64_t snippet_stride, int B, int H, int W, int ksize, const float* w, const float* scale,
                        const float* shift, float* out, void* stream) {
  return orcai_conv0_affine(in, snippet_stride, B, H, W, ksize, w, scale, shift, 1, out, stream);
}

int orcai_conv0_affine(const float* in, int64_t snippet_stride, int B, int H, int W, int ksize, const float* w, const float* scale, const float* shift,
                       int relu, float* out, void* stream) {
  if (!in || !w || !scale || !shift || !out || B <= 0 || H <= 0 || W <= 0) return ORCAI_E_BADARG;
  dim3 grid((W + 31) / 32, (H + 7) / 8, B);
  hipStream_t st = (hipStream_t)stream;
  const int WP = orcai_padded_width(W, ksize);
  switch (ksize) {
    case 3: hipLaunchKernelGGL(conv0_kernel<3>, grid, dim3(256), 0, st, in, snippet_stride, H, W, WP, w, scale, shift, out, relu); break;
    case 5: hipLaunchKernelGGL(conv0_kernel<5>, grid, dim3(256), 0, st, in, snippet_stride, H, W, WP, w, scale, shift, out, relu); break;
    case 7: hipLaunchKernelGGL(conv0_kernel<7>, grid, dim3(256), 0, st, in, snippet_stride, H, W, WP, w, scale, shift, out, relu); break;
    default: return ORCAI_E_UNSUPPORTED;
  }
  return (int)hipGetLastError();
}

int orcai_conv0_stats(const float* in, int64_t snippet_stride, int B, int H, int W, int ksize, const float* w, const float* scale, const float* shift, double* shards,
                      void* stream) {
  if (!in || !w || !scale || !shift || !shards || B <= 0 || H <= 0 || W <= 0) return ORCAI_E_BADARG;
  const int64_t ntiles = (int64_t)((W + 31) / 32) * ((H + 7) / 8) * B;
  if (ntiles >= (1ll << 31)) return ORCAI_E_UNSUPPORTED;
  dim3 grid((unsigned)(ntiles < 2048 ? ntiles : 2048));  // eight workgroups per compute unit, ~17 tiles each at batch 64
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = orcai_zero::zero_async(shards, sizeof(double) * 8 * 4 * 32, st);
  if (e != hipSuccess) return (int)e;
  switch (ksize) {
    case 3: hipLaunchKernelGGL(conv0_stats_kernel<3>, grid, dim3(256), 0, st, in, snippet_stride, H, W, B, w, scale, shift, shards); break;
    case 5: hipLaunchKernelGGL(conv0_stats_kernel<5>, grid, dim3(256), 0, st, in, snippet_stride, H, W, B, w, scale, shift, shards); break;
    case 7: hipLaunchKernelGGL(conv0_stats_kernel<7>, grid, dim3(256), 0, st, in, snippet_stride, H, W, B, w, scale, shift, shards); break;
    default: return ORCAI_E_UNSUPPORTED;
  }
  return (int)hipGetLastError();
}

int orcai_conv0_affine_bn(const float* in, int64_t snippet_stride, int B, int H, int W, int ksize, const float* w, const float* scale, const float* shift,
                          const float* bn_mean, const float* bn_var, const float* bn_gamma, const float* bn_beta, float bn_eps, int relu, float* out, void* stream) {
  if (!in || !w || !scale || !shift || !out || !bn_mean || !bn_var || !bn_gamma || !bn_beta || B <= 0 || H <= 0 || W <= 0) return ORCAI_E_BADARG;
  dim3 grid((W + 31) / 32, (H + 7) / 8, B);
  hipStream_t st = (hipStream_t)stream;
  const int WP = orcai_padded_width(W, ksize);
  switch (ksize) {
    case 3: hipLaunchKernelGGL(conv0_kernel<3>, grid, dim3(256), 0, st, in, snippet_stride, H, W, WP, w, scale, shift, out, relu, bn_mean, bn_var, bn_gamma, bn_beta, bn_eps); break;
    case 5: hipLaunchKernelGGL(conv0_kernel<5>, grid, dim3(256), 0, st, in, snippet_stride, H, W, WP, w, scale, shift, out, relu, bn_mean, bn_var, bn_gamma, bn_beta, bn_eps); break;
    case 7: hipLaunchKernelGGL(conv0_kernel<7>, grid, dim3(256), 0, st, in, snippet_stride, H, W, WP, w, scale, shift, out, relu, bn_mean, bn_var, bn_gamma, bn_beta, bn_eps); break;
    default: return ORCAI_E_UNSUPPORTED;
  }
  return (int)hipGetLastError();
}

int orcai_conv0_sepconv(const float* in, int64_t snippet_stride, int B, int H, int W, const float* w0, const float* scale0, const float* shift0,
                        const float* dw, const float* pw, const float* scale, const float* shift, int Cout, int relu_out, float* out, float* prev_sub,
                        void* stream) {
  if (!in || !w0 || !scale0 || !shift0 || !dw || !pw || !scale || !shift || !out || B <= 0 || H <= 0 || W <= 0 || Cout <= 0) return ORCAI_E_BADARG;
  if (Cout > 64 || (int64_t)H * W >= (1ll << 26) || (((uintptr_t)w0 | (uintptr_t)dw | (uintptr_t)scale0 | (uintptr_t)shift0) & 15)) return ORCAI_E_UNSUPPORTED;
  const int WP = orcai_padded_width(W, 3);
  if ((int64_t)((Cout + 3) / 4) * (H + 2) * WP >= (1ll << 28)) return ORCAI_E_UNSUPPORTED;  // 32-bit byte offsets inside a snippet's planes
  hipStream_t st = (hipStream_t)stream;
  const int nstrip = (W + 61) / 62;
  if (g_entry_tile && Cout > 16 && Cout <= 32 && nstrip >= 2 && W * 100 >= nstrip * 62 * 85) {  // wide planes: entry rows shared through LDS
    if (g_entry_tile == 16) {
      dim3 grid(nstrip * ((H + 13) / 14), B);
      hipLaunchKernelGGL((conv0_sep_tile_kernel<2, 16>), grid, dim3(1024), 0, st, in, snippet_stride, H, W, WP, w0, scale0, shift0, dw, pw, scale, shift, Cout,
                         relu_out, out, prev_sub, nstrip);
    } else {
      dim3 grid(nstrip * ((H + 7) / 8), B);
      hipLaunchKernelGGL((conv0_sep_tile_kernel<2, 10>), grid, dim3(640), 0, st, in, snippet_stride, H, W, WP, w0, scale0, shift0, dw, pw, scale, shift, Cout,
                         relu_out, out, prev_sub, nstrip);
    }
    return (int)hipGetLastError();
  }
  const int tasks = (H * WP + 61) / 62;
  const int NW = g_entry_windows;
  dim3 grid((tasks + 4 * NW - 1) / (4 * NW), B);  // a workgroup = 4 waves side by side over 4*NW consecutive windows
#define ORCAI_C0S(MT) hipLaunchKernelGGL(conv0_sep_kernel<MT>, grid, dim3(256), 0, st, in, snippet_stride, H, W, WP, w0, scale0, shift0, dw, pw, scale, shift, \
                                         Cout, relu_out, out, prev_sub, tasks, magic_for(WP), NW)
  switch ((Cout + 15) / 16) {
    case 1: ORCAI_C0S(1); break;
    case 2: ORCAI_C0S(2); break;
    case 3: ORCAI_C0S(3); break;
    case 4: ORCAI_C0S(4); break;
  }
#undef ORCAI_C0S
  return (int)hipGetLastError();
}

int orcai_sepconv_bn(const float* in, int B, int Cin, int H, int W, int ksize, int relu_in, const float* dw, const float* pw, const float* scale,
                     const float* shift, int Cout, int relu_out, int out_layout, float* out, void* stream) {
  return orcai_sepconv_planes(in, B, Cin, H, W, ksize, ksize, relu_in, dw, pw, scale, shift, Cout, relu_out, out_layout, 0, 0, out, stream);
}

static int sepconv_planes_stats_impl(const float* in, int B, int Cin, int H, int W, int relu_in, const float* dw, const float* pw, const float* scale, const float* shift,
                                     int Cout, float* out, float* u_out, double* shards, const InBn& ib, void* stream) {
  if (!in || !dw || !pw || !scale || !shift || !out || !u_out || !shards || B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) return ORCAI_E_BADARG;
  if (((uintptr_t)in & 15) || B > 65535 || Cout > 64 || Cin > 64) return ORCAI_E_UNSUPPORTED;
  // the shapes launch_sepconv_impl<3, MT> hands to the LDS-tile kernels, checked BEFORE anything is touched; everything else:
  // ORCAI_E_UNSUPPORTED, and the caller runs orcai_sepconv_planes_u + orcai_bn_planes_stats
  const int CQ = (Cin + 3) / 4, CQo = (Cout + 3) / 4, WP = orcai_padded_width(W, 3), MTv = (Cout + 15) / 16;
  const int nchunk = (7 * 62 + 64 + 2 * WP + 63) / 64;
  const size_t lds = (size_t)(2 * nchunk * 256 + CQ * 64 * MTv + 2 * MTv * 16 + 8 * 4 * 8 * MTv + (ib.mean ? 128 : 0)) * sizeof(float);
  const int nstrip = (W + 61) / 62;
  const bool strip = g_tile_mode == 1 && MTv == 2 && CQ <= 8 && nstrip >= 2 && W * 100 >= nstrip * 62 * 85;
  if (g_tile_mode == 0 || (int64_t)CQo * (H + 2) * WP >= (1ll << 27) || (int64_t)CQ * (H + 2) * WP >= (1ll << 27) || (!strip && (nchunk > 24 || lds > 64 * 1024)))
    return ORCAI_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  {
    hipError_t e = orcai_zero::zero_async(shards, sizeof(double) * 8 * CQo * 32, st);
    if (e != hipSuccess) return (int)e;
  }
  SepArgs a{in, dw, pw, scale, shift, out, B, Cin, H, W, WP, 1, Cout, relu_in, 0, 0, 0, 0, u_out, shards};
  a.ib = ib;
  return launch_sepconv<3>(st, a);
}

int orcai_sepconv_planes_stats(const float* in, int B, int Cin, int H, int W, int relu_in, const float* dw, const float* pw, const float* scale, const float* shift,
                               int Cout, float* out, float* u_out, double* shards, void* stream) {
  return sepconv_planes_stats_impl(in, B, Cin, H, W, relu_in, dw, pw, scale, shift, Cout, out, u_out, shards, InBn{}, stream);
}

int orcai_sepconv_planes_stats_bn(const float* v_in, int B, int Cin, int H, int W, const float* in_mean, const float* in_var, const float* in_gamma, const float* in_beta,
                                  float in_eps, const float* dw, const float* pw, const float* scale, const float* shift, int Cout, float* out, float* u_out, double* shards,
                                  void* stream) {
  if (!in_mean || !in_var || !in_gamma || !in_beta) return ORCAI_E_BADARG;
  InBn ib;
  ib.mean = in_mean; ib.var = in_var; ib.gamma = in_gamma; ib.beta = in_beta; ib.eps = in_eps;
  return sepconv_planes_stats_impl(v_in, B, Cin, H, W, 0, dw, pw, scale, shift, Cout, out, u_out, shards, ib, stream);
}

int orcai_sepconv_planes_epi(const float* in, int B, int Cin, int H, int W, const float* dw, const float* pw, const float* scale, const float* shift, int Cout, float* out,
                             int epi, const float* ref, const float* mean, const float* var, const float* gamma, const float* beta, float eps, int relu, double* shards,
                             void* stream) {
  if (!in || !dw || !pw || !scale || !shift || !out || !ref || B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0 || (epi != 2 && epi != 3)) return ORCAI_E_BADARG;
  if (epi == 2 && (!mean || !var || !gamma || !beta || !shards)) return ORCAI_E_BADARG;
  if (((uintptr_t)in & 15) || ((uintptr_t)ref & 15) || B > 65535 || Cout > 64) return ORCAI_E_UNSUPPORTED;
  // the shapes launch_sepconv_impl<3, MT> hands to the LDS-tile kernels, checked BEFORE anything is touched (as orcai_sepconv_planes_stats)
  const int CQ = (Cin + 3) / 4, CQo = (Cout + 3) / 4, WP = orcai_padded_width(W, 3), MTv = (Cout + 15) / 16;
  const int nchunk = (7 * 62 + 64 + 2 * WP + 63) / 64;
  const size_t lds = (size_t)(2 * nchunk * 256 + CQ * 64 * MTv + 2 * MTv * 16 + 8 * 4 * 8 * MTv + 4 * MTv * 16) * sizeof(float);
  const int nstrip = (W + 61) / 62;
  const bool strip = epi == 2 && g_tile_mode == 1 && MTv == 2 && CQ <= 8 && nstrip >= 2 && W * 100 >= nstrip * 62 * 85;
  if (g_tile_mode == 0 || (int64_t)CQo * (H + 2) * WP >= (1ll << 27) || (int64_t)CQ * (H + 2) * WP >= (1ll << 27) || (!strip && (nchunk > 24 || lds > 64 * 1024)))
    return ORCAI_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (epi == 2) {
    hipError_t e = orcai_zero::zero_async(shards, sizeof(double) * 8 * CQo * 32, st);
    if (e != hipSuccess) return (int)e;
  }
  SepArgs a{in, dw, pw, scale, shift, out, B, Cin, H, W, WP, 1, Cout, 0, 0, 0, 0, 0, nullptr, epi == 2 ? shards : nullptr};
  a.epi = epi;
  a.er.ref = ref;
  a.er.mean = mean; a.er.var = var; a.er.gamma = gamma; a.er.beta = beta; a.er.eps = eps; a.er.relu = relu;
  const int rc = launch_sepconv<3>(st, a);
  if (rc != 0) return rc;
  if (epi == 2) hipLaunchKernelGGL(bn_bwd_sums_compact_kernel, dim3(1), dim3(256), 0, st, shards, CQo);
  return (int)hipGetLastError();
}

int orcai_sepconv_planes_u(const float* in, int B, int Cin, int H, int W, int ksize_planes, int ktap, int relu_in, const float* dw, const float* pw,
                         const float* scale, const float* shift, int Cout, int relu_out, int out_layout, int H2, int W2, float* out, float* u_out, void* stream) {
  if (!in || !dw || !pw || !scale || !shift || !out || B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) return ORCAI_E_BADARG;
  if (Cout > 64 || ((uintptr_t)in & 15) || ktap > ksize_planes) return ORCAI_E_UNSUPPORTED;
  if (out_layout == 3 && (H2 < 2 * H - 1 || W2 < 2 * W - 1)) return ORCAI_E_BADARG;
  SepArgs a{in, dw, pw, scale, shift, out, B, Cin, H, W, orcai_padded_width(W, ksize_planes), ksize_planes / 2, Cout, relu_in, relu_out, out_layout,
            H2, out_layout == 3 ? orcai_padded_width(W2, ksize_planes) : 0, u_out};
  hipStream_t st = (hipStream_t)stream;
  switch (ktap) {
    case 1: return launch_sepconv<1>(st, a);
    case 3: return launch_sepconv<3>(st, a);
    case 5: return launch_sepconv<5>(st, a);
    case 7: return launch_sepconv<7>(st, a);
    default: return ORCAI_E_UNSUPPORTED;
  }
}

int orcai_sepconv_planes(const float* in, int B, int Cin, int H, int W, int ksize_planes, int ktap, int relu_in, const float* dw, const float* pw,
                         const float* scale, const float* shift, int Cout, int relu_out, int out_layout, int H2, int W2, float* out, void* stream) {
  return orcai_sepconv_planes_u(in, B, Cin, H, W, ksize_planes, ktap, relu_in, dw, pw, scale, shift, Cout, relu_out, out_layout, H2, W2, out, nullptr, stream);
}

int orcai_pool_res_add_bn(const float* s, const float* prev, int B, int C, int Cp, int H, int W, int ksize, const float* wr, const float* br, float* out,
                          int xpooled, const float* bn_mean, const float* bn_var, const float* bn_gamma, const float* bn_beta, float bn_eps, void* stream) {
  if (!s || !prev || !wr || !br || !out || B <= 0 || C <= 0 || Cp <= 0 || H <= 0 || W <= 0) return ORCAI_E_BADARG;
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  int tot_h = (Ho - 1) * 2 + 3 - H, tot_w = (Wo - 1) * 2 + 2 - W;
  if (tot_h < 0) tot_h = 0;
  if (tot_w < 0) tot_w = 0;
  if ((xpooled & 1) && tot_w / 2 != 0) return ORCAI_E_UNSUPPORTED;
  if (xpooled & ~3) return ORCAI_E_BADARG;
  if (bn_mean && (xpooled || !bn_var || !bn_gamma || !bn_beta)) return ORCAI_E_BADARG;
  const int WP = orcai_padded_width(W, ksize), WPo = orcai_padded_width(Wo, ksize), R = ksize / 2;
  const int tasks = (Ho * WPo + 63) / 64;
  dim3 grid((tasks + 3) / 4, B);
  hipStream_t st = (hipStream_t)stream;
  const uint32_t mg = magic_for(WPo);
  // stacked tiles (a wave = 4 output rows x 16 columns, the row two pooling windows share loaded once) where 16-column tiles waste few lanes
  const int ntc = (Wo + 15) / 16, vtasks = ntc * ((Ho + 3) / 4);
  const bool vert = g_pool_vert && Wo >= 40;
  dim3 vgrid((vtasks + 3) / 4, B);
#define ORCAI_POOL_LAUNCH(MT)                                                                                                                        \
  if ((xpooled & 1) && !bn_mean && (int64_t)((C + 3) / 4) * H * (((Wo + 3) & ~3)) < (1ll << 28)) {                                                                   \
    if (vert)                                                                                                                                        \
      hipLaunchKernelGGL((pool_res_add_x_kernel<MT, true>), vgrid, dim3(256), 0, st, s, prev, C, Cp, H, W, WP, R, Ho, Wo, WPo, tot_h / 2, wr, br, out,            \
                         (xpooled >> 1) & 1, vtasks, mg, ntc);                                                                                       \
    else                                                                                                                                             \
      hipLaunchKernelGGL((pool_res_add_x_kernel<MT, false>), grid, dim3(256), 0, st, s, prev, C, Cp, H, W, WP, R, Ho, Wo, WPo, tot_h / 2, wr, br, out,            \
                         (xpooled >> 1) & 1, tasks, mg, 0);                                                                                          \
  } else                                                                                                                                               \
    hipLaunchKernelGGL(pool_res_add_kernel<MT>, grid, dim3(256), 0, st, s, prev, C, Cp, H, W, WP, R, Ho, Wo, WPo, tot_h / 2, tot_w / 2, wr, br, out, xpooled, tasks, mg, bn_mean, bn_var, bn_gamma, bn_beta, bn_eps)
  switch ((C + 15) / 16) {
    case 1: ORCAI_POOL_LAUNCH(1); break;
    case 2: ORCAI_POOL_LAUNCH(2); break;
    case 3: ORCAI_POOL_LAUNCH(3); break;
    case 4: ORCAI_POOL_LAUNCH(4); break;
    default: return ORCAI_E_UNSUPPORTED;
  }
#undef ORCAI_POOL_LAUNCH
  return (int)hipGetLastError();
}

int g_pool_fused_nt = 8;  // tiles per workgroup of sepconv_pool_march_kernel (orcai_pool_fused); 0 = the two-launch tail everywhere

int orcai_pool_fused(int nt) {
  const int prev = g_pool_fused_nt;
  if (nt >= 0) g_pool_fused_nt = nt > 64 ? 64 : nt;
  return prev;
}

int orcai_sepconv_pool_res(const float* in, const float* prev, int B, int Cin, int C, int Cp, int H, int W, int ksize, int relu_in, const float* dw, const float* pw,
                           const float* scale, const float* shift, int relu_out, const float* wr, const float* br, float* out, int prev_compact, void* stream) {
  if (!in || !prev || !dw || !pw || !scale || !shift || !wr || !br || !out || B <= 0 || Cin <= 0 || C <= 0 || Cp <= 0 || H <= 0 || W <= 0) return ORCAI_E_BADARG;
  // the marching kernel's shapes: k = 3, two output tiles (17 .. 32 channels), <= 8 input quads, <= 4 quads of prev, an even number of rows (pooling pads
  // at the bottom only), a plane wide enough for the 60-column strips of sepconv_tile_kernel; everything else: the caller's two launches
  const int nstrip = (W + 59) / 60, CQ = (Cin + 3) / 4, WP = orcai_padded_width(W, 3);
  if (g_pool_fused_nt <= 0 || g_tile_mode != 1 || ksize != 3 || (C + 15) / 16 != 2 || CQ < 5 || CQ > 8 || Cp > 16 || (H & 1) || nstrip < 2 || W * 100 < nstrip * 60 * 85 ||
      ((uintptr_t)in & 15) || ((uintptr_t)prev & 15) || ((uintptr_t)out & 15) || B > 65535 || (int64_t)8 * (H + 2) * WP >= (1ll << 27))
    return ORCAI_E_UNSUPPORTED;
  const int Ho = H / 2, Wo = (W + 1) / 2, WPo = orcai_padded_width(Wo, 3);
  const int NT = g_pool_fused_nt, nseg = (Ho + 4 * NT - 2) / (4 * NT - 1);
  dim3 grid(nstrip * nseg, B);
  hipStream_t st = (hipStream_t)stream;
#define ORCAI_PM_LAUNCH(CQT, RELU)                                                                                                                          \
  hipLaunchKernelGGL((sepconv_pool_march_kernel<CQT, RELU>), grid, dim3(512), 0, st, in, Cin, H, W, WP, dw, pw, scale, shift, C, relu_out, prev, Cp, prev_compact ? 1 : 0, \
                     wr, br, out, Ho, Wo, WPo, nstrip, NT)
  switch (CQ) {  // C in 17 .. 32 and Cin = C for a block's second conv: 5 .. 8 input quads
    case 5: if (relu_in) ORCAI_PM_LAUNCH(5, true); else ORCAI_PM_LAUNCH(5, false); break;
    case 6: if (relu_in) ORCAI_PM_LAUNCH(6, true); else ORCAI_PM_LAUNCH(6, false); break;
    case 7: if (relu_in) ORCAI_PM_LAUNCH(7, true); else ORCAI_PM_LAUNCH(7, false); break;
    case 8: if (relu_in) ORCAI_PM_LAUNCH(8, true); else ORCAI_PM_LAUNCH(8, false); break;
    default: return ORCAI_E_UNSUPPORTED;
  }
#undef ORCAI_PM_LAUNCH
  return (int)hipGetLastError();
}

int orcai_pool_vertical(int on) {
  const int prev = g_pool_vert;
  if (on >= 0) g_pool_vert = on ? 1 : 0;
  return prev;
}

int orcai_pool_res_add(const float* s, const float* prev, int B, int C, int Cp, int H, int W, int ksize, const float* wr, const float* br, float* out,
                       int xpooled, void* stream) {
  return orcai_pool_res_add_bn(s, prev, B, C, Cp, H, W, ksize, wr, br, out, xpooled, nullptr, nullptr, nullptr, nullptr, 0.0f, stream);
}

int orcai_gemm_bias_act(const float* A, const float* Bm, const float* bias, const float* scale, const float* shift, float* C, int64_t M, int N,
                        int K, int act, void* stream) {
  if (!A || !Bm || !C || M <= 0 || N <= 0 || K <= 0 || (scale && !shift)) return ORCAI_E_BADARG;
  dim3 grid((N + 127) / 128, (unsigned)((M + 127) / 128));
  const int vec = (((uintptr_t)A | (uintptr_t)Bm) % 16 == 0 && K % 4 == 0 && N % 4 == 0) ? 1 : 0;
  hipLaunchKernelGGL(gemm_kernel, grid, dim3(256), 0, (hipStream_t)stream, A, Bm, bias, scale, shift, C, (int)M, N, K, act, vec);
  return (int)hipGetLastError();
}

int orcai_lstm_split(int on) {
  const int prev = g_orcai_lstm_split;
  if (on >= 0) g_orcai_lstm_split = on ? 1 : 0;
  return prev;
}

int orcai_lstm_recurrent(const float* xz, const float* Uw, int B, int T, int units, float* out, void* stream) {
  if (!xz || !Uw || !out || B <= 0 || T <= 0) return ORCAI_E_BADARG;
  dim3 grid((B + 15) / 16, 2);
  hipStream_t st = (hipStream_t)stream;
  if (g_orcai_lstm_split) {
    switch (units) {
      case 128: hipLaunchKernelGGL(lstm_split_kernel<128>, grid, dim3(1024), 0, st, xz, Uw, B, T, out); break;
      case 64: hipLaunchKernelGGL(lstm_split_kernel<64>, grid, dim3(512), 0, st, xz, Uw, B, T, out); break;
      default: return ORCAI_E_UNSUPPORTED;
    }
    return (int)hipGetLastError();
  }
  switch (units) {
    case 128: hipLaunchKernelGGL(lstm_kernel<128>, grid, dim3(1024), 0, st, xz, Uw, B, T, out); break;
    case 64: hipLaunchKernelGGL(lstm_kernel<64>, grid, dim3(512), 0, st, xz, Uw, B, T, out); break;
    default: return ORCAI_E_UNSUPPORTED;
  }
  return (int)hipGetLastError();
}

int orcai_dense_sigmoid(const float* x, const float* w, const float* bias, int64_t M, int K, int N, float* out, void* stream) {
  if (!x || !w || !bias || !out || M <= 0 || K <= 0 || N <= 0) return ORCAI_E_BADARG;
  if (N > 8) return ORCAI_E_UNSUPPORTED;
  if ((K & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0)
    hipLaunchKernelGGL(dense_sigmoid_rows_kernel, dim3((unsigned)((M + 31) / 32)), dim3(256), 0, (hipStream_t)stream, x, w, bias, M, K, N, out);
  else
    hipLaunchKernelGGL(dense_sigmoid_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, w, bias, M, K, N, out);
  return (int)hipGetLastError();
}

int orcai_overlap_average(const float* pred, int n, int P, int L, int step, int64_t S, double* agg, double* cnt, void* stream) {
  if (!pred || !agg || !cnt || n < 0 || P <= 0 || L <= 0 || step <= 0 || S <= 0) return ORCAI_E_BADARG;
  hipLaunchKernelGGL(overlap_average_kernel, dim3((unsigned)((S * L + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pred, n, P, L, step, S, agg, cnt);
  return (int)hipGetLastError();
}

}  // extern "C"
