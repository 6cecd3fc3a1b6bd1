// half_fwd.hip -- the f16 path of orcAI's ResNetLSTM forward for gfx950 (BASELINE configs[4]: hyper-parameter variants on the
// "fp16 MFMA path"): activations as f16 channel-octet planes (half_planes.h), every contraction -- pointwise 1x1 convolutions,
// strided residual convolutions, LSTM input projections, Dense-128 -- on v_mfma_f32_16x16x32_f16 with f32 accumulation, BatchNorm /
// bias / activation epilogues in f32, depthwise taps in packed f16 (v_pk_fma_f16).  Master weights stay f32 (the packers convert).
// The 46-step LSTM recurrences stay on the exact-f32 kernels of model_fwd.hip: they are latency-bound chains, not a throughput item.
// Reference layers: architectures.py:162-241; reference strategy for this configuration: hpsearch.py:186-205.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "half_planes.h"
#include "lds_dma.h"
#include "orcai_hip.h"
#include "zero_fill.h"

namespace {

using namespace orcai_half;

// ---------------------------------------------------------------- entry conv: Conv2D(16, k x k, same) on the 1-channel f32 snippet
// out f16 planes of 16 channels (2 octets), y = [relu](conv * scale + shift)   (architectures.py:164-168)
// BN (training forward, orcai_h_conv0_affine_bn): the pre-normalisation tensor goes to `out` as before and y = relu(BatchNorm(out)) -- formed from the f16
// value just stored, with bn_planes_apply_h_kernel's arithmetic, bit for bit -- to y_out in the same pass: the separate apply pass (read + write of the 16
// channels) is gone.  The batch statistics come from the snippet (orcai_conv0_stats_march), so they exist before this launch.
template <int KS, bool BN = false>
__global__ __launch_bounds__(256) void conv0_h_kernel(const float* __restrict__ in, int64_t snippet_stride, int H, int W, int WP,
                                                       const float* __restrict__ w /*[KS*KS][16]*/, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, h16* __restrict__ out /*[B][2][HP][WP][8]*/, int relu,
                                                       const float* __restrict__ bn_mean = nullptr, const float* __restrict__ bn_var = nullptr,
                                                       const float* __restrict__ bn_gamma = nullptr, const float* __restrict__ bn_beta = nullptr, float bn_eps = 0.0f,
                                                       h16* __restrict__ y_out = nullptr) {
  constexpr int TH = 8, TW = 32, R = KS / 2, HH = TH + KS - 1, HW = TW + KS - 1, HP_ = HW + 1;
  __shared__ float halo[HH][HP_];
  __shared__ float bn2_s[2][16];  // BN: the folded scale / shift of the 16 channels, once per workgroup (per thread it is 16 v_rsq and 64 loads per PIXEL)
  if (BN && threadIdx.x < 16) {
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
    const int64_t plane = (int64_t)(H + 2 * R) * WP;
    h16x8* o = reinterpret_cast<h16x8*>(out) + (int64_t)b * 2 * plane + (int64_t)(y + R) * WP + x;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[e] = fmaf(acc[8 * q + e], scale[8 * q + e], shift[8 * q + e]);
        if (relu) v[e] = fmaxf(v[e], 0.0f);
      }
      const h16x8 stored = pack8(v);
      o[(int64_t)q * plane] = stored;
      if (BN) {
        float a[8], r[8];
        unpack8(stored, a);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int c = 8 * q + e;
          r[e] = fmaxf(fmaf(a[e], bn2_s[0][c], bn2_s[1][c]), 0.0f);
        }
        (reinterpret_cast<h16x8*>(y_out) + (int64_t)b * 2 * plane + (int64_t)(y + R) * WP + x)[(int64_t)q * plane] = pack8(r);
      }
    }
  }
}

// ---------------------------------------------------------------- depthwise stage of one octet, lane = pixel, packed f16
// rows[dy] = the octet's 16 bytes of window row dy; wq = the octet's taps [k*k][8] f16 (wave-uniform: scalar loads).
// Horizontal taps by linearity on per-column partial sums (k - 1 lane shifts per register), as dw_quad_impl of the f32 path.
template <int KS, bool RELU>
__device__ __forceinline__ h16x8 dw_octet(const h16x8 (&rows)[KS], const h16* __restrict__ wq) {
  constexpr int R = KS / 2;
  h16x8 p[KS];
#pragma unroll
  for (int dx = 0; dx < KS; ++dx) p[dx] = zero_h();
#pragma unroll
  for (int dy = 0; dy < KS; ++dy) {
    const h16x8 a = RELU ? relu_h(rows[dy]) : rows[dy];
#pragma unroll
    for (int dx = 0; dx < KS; ++dx) {
      const h16x8 wv = *reinterpret_cast<const h16x8*>(wq + (dy * KS + dx) * 8);
      p[dx] = a * wv + p[dx];
    }
  }
  h16x8 acc = p[R];
#pragma unroll
  for (int dx = 0; dx < KS; ++dx) {
    if (dx == R) continue;
    if (dx - R == -3) acc += lane_shift_h<-3>(p[dx]);
    else if (dx - R == -2) acc += lane_shift_h<-2>(p[dx]);
    else if (dx - R == -1) acc += lane_shift_h<-1>(p[dx]);
    else if (dx - R == 1) acc += lane_shift_h<1>(p[dx]);
    else if (dx - R == 2) acc += lane_shift_h<2>(p[dx]);
    else acc += lane_shift_h<3>(p[dx]);
  }
  return acc;
}

// =========================================================================================
// sepconv_h: [ReLU] -> depthwise k x k (same) -> pointwise 1x1 -> y = acc * scale + shift -> [ReLU]   (architectures.py:174-189, :198-206)
// One wave owns 64 consecutive flat pixels of the padded planes (lane = pixel, the inner 64 - 2 lo are outputs), exactly as
// sepconv_kernel of the f32 path; per K group of 4 input octets it forms the depthwise outputs (packed f16), optionally stores
// them (training keeps u for the pointwise weight gradient), transposes them into B fragments and issues 4 x MT MFMAs.
// out_layout 0: f16 octet planes; 1: f32 [B][H][W*Cout] (Keras Reshape: feature = x*Cout + c); 2: x-pooled f16
// [B][CO][H][roundup4(ceil(W/2))][8] (max over the column pair); 3: scatter-add into pixel (2y, 2x) of f16 planes (H2, WP2):
// the input gradient of a stride-2 1x1 convolution.
// =========================================================================================
template <int KS, int MT>
__global__ __launch_bounds__(256) void sepconv_h_kernel(const h16* __restrict__ in /*[B][COin][HP][WP][8]*/, int Cin, int H, int W, int WP, int relu_in,
                                                         const h16* __restrict__ dw /*[COin][KS*KS][8]*/, const h16* __restrict__ pwf /*[KG][MT][64][8]*/,
                                                         const float* __restrict__ scale, const float* __restrict__ shift, int Cout, int relu_out,
                                                         int out_layout, void* __restrict__ out, int tasks, uint32_t magic_WP, int lo, int RP, int H2, int WP2,
                                                         h16* __restrict__ u_out /*optional [B][COin][HP][WP][8]: the depthwise output*/) {
  constexpr int KK = KS * KS;
  __shared__ h16x8 pw_s[2 * MT * 64];  // up to 2 K groups (Cin <= 64)
  const int R = RP;
  const int VAL = 64 - 2 * lo;
  const int lane = threadIdx.x & 63;
  int bx, b;
  xcd_remap(bx, b);
  const int task = bx * 4 + (threadIdx.x >> 6);
  const int CO = (Cin + 7) >> 3, COo = (Cout + 7) >> 3, KG = (CO + 3) >> 2;
  for (int i = threadIdx.x; i < KG * MT * 64; i += 256) pw_s[i] = reinterpret_cast<const h16x8*>(pwf)[i];
  __syncthreads();  // the only barrier
  if (task >= tasks) return;
  const int lk = lane >> 4, lj = lane & 15;
  const int plane = (H + 2 * R) * WP;
  const h16x8* src = reinterpret_cast<const h16x8*>(in) + (int64_t)b * CO * plane;
  const int qbase = R * WP + task * VAL - lo;
  const int q = qbase + lane;

  int ridx[KS];
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

  // register pipeline two octets deep (as sepconv_kernel of the f32 path): the rows of octets o + 1 and o + 2 are in flight while
  // octet o is being processed
  h16x8 nxt[KS], nx2[KS];
#pragma unroll
  for (int dy = 0; dy < KS; ++dy) nxt[dy] = src[ridx[dy]];
  if (CO > 1) {
#pragma unroll
    for (int dy = 0; dy < KS; ++dy) nx2[dy] = src[plane + ridx[dy]];
  }
  const int urow = (int)__umulhi((uint32_t)q, magic_WP);
  const bool u_live = u_out && lane >= lo && lane < 64 - lo && (q - urow * WP) < W && urow < R + H;

  for (int kg = 0; kg < KG; ++kg) {
    u32x4 d[4];
#pragma unroll
    for (int oo = 0; oo < 4; ++oo) {
      const int o = kg * 4 + oo;
      if (o < CO) {  // wave-uniform
        h16x8 cur[KS];
#pragma unroll
        for (int dy = 0; dy < KS; ++dy) { cur[dy] = nxt[dy]; nxt[dy] = nx2[dy]; }
        if (o + 2 < CO) {
          const h16x8* pn = src + (int64_t)(o + 2) * plane;
#pragma unroll
          for (int dy = 0; dy < KS; ++dy) nx2[dy] = pn[ridx[dy]];
        }
        const h16x8 dd = relu_in ? dw_octet<KS, true>(cur, dw + (int64_t)o * KK * 8) : dw_octet<KS, false>(cur, dw + (int64_t)o * KK * 8);
        if (u_live) reinterpret_cast<h16x8*>(u_out)[((int64_t)b * CO + o) * plane + q] = dd;
        d[oo] = as_u(dd);
      } else {
        d[oo] = (u32x4){0u, 0u, 0u, 0u};
      }
    }
    octets_to_fragments(d);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const h16x8 a = pw_s[(kg * MT + m) * 64 + lane];
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[m][t] = mfma_h(a, as_h(d[t]), acc[m][t]);
    }
  }

  // ---- epilogue.  D[row = 4 lk + r -> cout 16 m + 4 lk + r][col = lj -> pixel 16 t + lj of the window]
  float sc_r[MT][4], sh_r[MT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = m * 16 + lk * 4 + r;
      sc_r[m][r] = co < Cout ? scale[co] : 0.0f;
      sh_r[m][r] = co < Cout ? shift[co] : 0.0f;
    }
  if (out_layout == 1) {  // Keras Reshape((-1, W*C)) of NHWC, f32: feature = x*Cout + co   (architectures.py:208)
    float* of = reinterpret_cast<float*>(out);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int wl = 16 * t + lj, flat = qbase + wl;
      const int row = (int)__umulhi((uint32_t)flat, magic_WP);
      const int x = flat - row * WP;
      if (!(wl >= lo && wl < 64 - lo && x < W && row < R + H)) continue;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = m * 16 + lk * 4 + r;
          float v = fmaf(acc[m][t][r], sc_r[m][r], sh_r[m][r]);
          if (relu_out) v = fmaxf(v, 0.0f);
          if (co < Cout) of[((int64_t)b * H + (row - R)) * ((int64_t)W * Cout) + (int64_t)x * Cout + co] = v;
        }
    }
    return;
  }
  const int Wx = (W + 1) >> 1, WPx = (Wx + 3) & ~3;
#pragma unroll
  for (int tp = 0; tp < 4; tp += 2) {
    // the pixel this lane will hold after the tile pair is folded into octets
    const int wl = 16 * (tp + (lk & 1)) + lj, flat = qbase + wl;
    const int row = (int)__umulhi((uint32_t)flat, magic_WP);
    const int x = flat - row * WP;
    const bool live = wl >= lo && wl < 64 - lo && x < W && row < R + H;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      float a[4], bq[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        a[r] = fmaf(acc[m][tp][r], sc_r[m][r], sh_r[m][r]);
        bq[r] = fmaf(acc[m][tp + 1][r], sc_r[m][r], sh_r[m][r]);
        if (relu_out) { a[r] = fmaxf(a[r], 0.0f); bq[r] = fmaxf(bq[r], 0.0f); }
        if (out_layout == 2) {  // max over the column pair (2j, 2j+1): the window starts on an even pixel, so pairs are lanes (2k, 2k+1)
          // whether the pair's second column exists is a property of the ORIGINAL lane's pixel (tile tp / tp + 1, column lj)
          const int f0 = qbase + 16 * tp + lj, f1 = f0 + 16;
          const int r0 = (int)__umulhi((uint32_t)f0, magic_WP), r1 = (int)__umulhi((uint32_t)f1, magic_WP);
          const float oa = __shfl_xor(a[r], 1, 64), ob = __shfl_xor(bq[r], 1, 64);
          if ((f0 - r0 * WP) + 1 < W) a[r] = fmaxf(a[r], oa);
          if ((f1 - r1 * WP) + 1 < W) bq[r] = fmaxf(bq[r], ob);
        }
      }
      float o8[8];
      tiles_to_octet(a, bq, o8);
      const int oq = 2 * m + (lk >> 1);
      if (!live || oq >= COo) continue;
      const h16x8 val = pack8(o8);
      if (out_layout == 0) {
        reinterpret_cast<h16x8*>(out)[((int64_t)b * COo + oq) * plane + flat] = val;
      } else if (out_layout == 2) {
        if ((x & 1) == 0) reinterpret_cast<h16x8*>(out)[(((int64_t)b * COo + oq) * H + (row - R)) * WPx + (x >> 1)] = val;
      } else {  // 3: scatter-add to pixel (2y, 2x) of planes [B][COo][H2 + 2R][WP2][8]
        const int64_t plane2 = (int64_t)(H2 + 2 * R) * WP2;
        h16x8* o2 = reinterpret_cast<h16x8*>(out) + ((int64_t)b * COo + oq) * plane2 + (int64_t)(2 * (row - R) + R) * WP2 + 2 * x;
        *o2 = *o2 + val;
      }
    }
  }
}

// =========================================================================================
// pool_res_add_h: MaxPooling2D((3,2), 2, "same")(s) + Conv2D(C, 1, strides 2)(prev) + bias   (architectures.py:190-196)
// One wave owns 64 consecutive flat pixels of the padded OUTPUT plane.  The strided 1x1 residual convolution is an MFMA contraction
// over prev's octets (lane = output pixel loads the 16 bytes of source pixel (2i, 2j) per octet); the D tiles are folded into octets
// (tiles_to_octet), so each lane then owns one output octet of one pixel: it reads its 3 (x-pooled s) or 3 x 2 (plane s) pooling
// operands as 16-byte vectors, takes the maximum in f16, adds the residual in f32 and stores 16 bytes.
// bn_mean != NULL (training forward): s holds the PRE-BatchNorm tensor and the pooling runs on BN(s) = fma(s, sc, sh) without
// materialising it: fma is monotone per channel, so the maximum (minimum for sc < 0) is transformed once.
// =========================================================================================
template <int MT>
__global__ __launch_bounds__(256) void pool_res_add_h_kernel(const h16* __restrict__ s, const h16* __restrict__ prev, int C, int Cp, int H, int W, int WP, int R,
                                                              int Ho, int Wo, int WPo, int pad_top, int pad_left, const h16* __restrict__ wrf /*[KGp][MT][64][8]*/,
                                                              const float* __restrict__ br, h16* __restrict__ out /*[B][CO][Ho+2R][WPo][8]*/, int xpooled,
                                                              int tasks, uint32_t magic_WPo, const float* __restrict__ bn_mean, const float* __restrict__ bn_var,
                                                              const float* __restrict__ bn_gamma, const float* __restrict__ bn_beta, float bn_eps) {
  const int lane = threadIdx.x & 63;
  int bx, b;
  xcd_remap(bx, b);
  const int task = bx * 4 + (threadIdx.x >> 6);
  if (task >= tasks) return;
  const int lk = lane >> 4, lj = lane & 15;
  const int CO = (C + 7) >> 3, COp = (Cp + 7) >> 3, KGp = (COp + 3) >> 2;
  const int plane = (H + 2 * R) * WP, plane_o = (Ho + 2 * R) * WPo;
  const int qbase = R * WPo + task * 64;
  const int q = qbase + lane;
  const int prow = (int)__umulhi((uint32_t)q, magic_WPo);
  const int pj = q - prow * WPo, pi = prow - R;
  const bool pvalid = pj < Wo && pi < Ho;
  const int srcpix = pvalid ? (2 * pi + R) * WP + 2 * pj : 0;
  const h16x8* pp = reinterpret_cast<const h16x8*>(prev) + (int64_t)b * COp * plane + srcpix;

  f32x4 acc[MT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int kg = 0; kg < KGp; ++kg) {
    u32x4 d[4];
#pragma unroll
    for (int oo = 0; oo < 4; ++oo) {
      const int o = kg * 4 + oo;
      d[oo] = (o < COp) ? as_u(pp[(int64_t)o * plane]) : (u32x4){0u, 0u, 0u, 0u};
      if (!pvalid) d[oo] = (u32x4){0u, 0u, 0u, 0u};
    }
    octets_to_fragments(d);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const h16x8 a = reinterpret_cast<const h16x8*>(wrf)[(kg * MT + m) * 64 + lane];
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[m][t] = mfma_h(a, as_h(d[t]), acc[m][t]);
    }
  }
  const int WPx = (Wo + 3) & ~3;
  // training forward: the folded BatchNorm of this lane's output octets, once per window (the octet of a lane does not change with the tile pair)
  float bsc[MT][8], bsh[MT][8];
  if (bn_mean) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = (2 * m + (lk >> 1)) * 8 + e, cc = c < C ? c : 0;
        const float sc = bn_gamma[cc] * rsqrtf(bn_var[cc] + bn_eps);
        bsc[m][e] = sc;
        bsh[m][e] = bn_beta[cc] - bn_mean[cc] * sc;
      }
  }
#pragma unroll
  for (int tp = 0; tp < 4; tp += 2) {
    const int flat = qbase + 16 * (tp + (lk & 1)) + lj;
    const int row = (int)__umulhi((uint32_t)flat, magic_WPo);
    const int j = flat - row * WPo, i = row - R;
    const bool valid = j < Wo && i < Ho;
    const int ys = 2 * i - pad_top, xs = 2 * j - pad_left;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      float a[4], bq[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = m * 16 + lk * 4 + r;
        const float bias = co < C ? br[co] : 0.0f;
        a[r] = acc[m][tp][r] + bias;
        bq[r] = acc[m][tp + 1][r] + bias;
      }
      float res[8];
      tiles_to_octet(a, bq, res);
      const int oq = 2 * m + (lk >> 1);
      if (!valid || oq >= CO) continue;
      h16x8 mx, mn;
      if (xpooled) {  // s is [B][CO][H][WPx][8], already reduced over the column pair
        const h16x8* sp = reinterpret_cast<const h16x8*>(s) + ((int64_t)b * CO + oq) * (int64_t)H * WPx;
        h16x8 v[3];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          int y = ys + dy;
          y = y < 0 ? 0 : (y >= H ? H - 1 : y);  // a duplicated row leaves the maximum unchanged
          v[dy] = sp[(int64_t)y * WPx + j];
        }
        mx = max_h(max_h(v[0], v[1]), v[2]);
        mn = mx;
      } else {
        // all six operands requested before the first is used: clamped coordinates (a duplicated row / column leaves a maximum and a
        // minimum unchanged) instead of a bounds branch around every load
        const h16x8* sp = reinterpret_cast<const h16x8*>(s) + ((int64_t)b * CO + oq) * plane;
        h16x8 v[6];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int dx = 0; dx < 2; ++dx) {
            int y = ys + dy, x = xs + dx;
            y = y < 0 ? 0 : (y >= H ? H - 1 : y);
            x = x < 0 ? 0 : (x >= W ? W - 1 : x);
            v[dy * 2 + dx] = sp[(int64_t)(y + R) * WP + x];
          }
        mx = v[0];
        mn = v[0];
#pragma unroll
        for (int e = 1; e < 6; ++e) {
          mx = max_h(mx, v[e]);
          mn = min_h(mn, v[e]);
        }
      }
      float pm[8], pn[8], o8[8];
      unpack8(mx, pm);
      unpack8(mn, pn);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = oq * 8 + e;
        float pooled = pm[e];
        if (bn_mean) pooled = fmaf(bsc[m][e] >= 0.0f ? pm[e] : pn[e], bsc[m][e], bsh[m][e]);
        o8[e] = c < C ? pooled + res[e] : 0.0f;
      }
      reinterpret_cast<h16x8*>(out)[((int64_t)b * CO + oq) * plane_o + flat] = pack8(o8);
    }
  }
}

// =========================================================================================
// gemm_h: C[M][N] = act(A[M][K] * W[K][N] + bias[N]) [* scale[N] + shift[N]],  A f32 (converted to f16 on load), W given TRANSPOSED
// and zero-padded as f16 Wt[N][Kp] (Kp = roundup32(K)) so that both MFMA operands are 16-byte runs along k, f32 accumulate, f32 out.
// (LSTM input projections: x W + b for both directions at once; Dense-128 + ReLU + folded BN; architectures.py:210-237.)
// 4 waves as 2 x 2, wave tile 32 x 32 (2 x 2 MFMA tiles), block tile 64 x 64; fragments straight from global memory (the operands of
// these products are L2-resident: A is re-read by N/64 column blocks, W by all row blocks).
// =========================================================================================
__global__ __launch_bounds__(256) void gemm_h_kernel(const float* __restrict__ A, const h16* __restrict__ Wt, const float* __restrict__ bias,
                                                      const float* __restrict__ scale, const float* __restrict__ shift, float* __restrict__ C, int64_t M, int N,
                                                      int K, int Kp, int act) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lk = lane >> 4, lj = lane & 15;
  const int64_t m0 = (int64_t)blockIdx.y * 64 + wm * 32;
  const int n0 = blockIdx.x * 64 + wn * 32;
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto load_a = [&](int i, int k0) -> h16x8 {
    const int64_t m = m0 + i * 16 + lj;
    const int k = k0 + 8 * lk;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (m < M) {
      const float* p = A + m * K + k;
      if (k + 7 < K) {
        const float4 u0 = *reinterpret_cast<const float4*>(p), u1 = *reinterpret_cast<const float4*>(p + 4);
        v[0] = u0.x; v[1] = u0.y; v[2] = u0.z; v[3] = u0.w; v[4] = u1.x; v[5] = u1.y; v[6] = u1.z; v[7] = u1.w;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (k + e < K) ? p[e] : 0.0f;
      }
    }
    return pack8(v);
  };
  auto load_b = [&](int j, int k0) -> h16x8 {
    const int n = n0 + j * 16 + lj;
    return n < N ? *reinterpret_cast<const h16x8*>(Wt + (int64_t)n * Kp + k0 + 8 * lk) : zero_h();
  };
  h16x8 a[2], bq[2], an[2], bn[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) { an[i] = load_a(i, 0); bn[i] = load_b(i, 0); }
  for (int k0 = 0; k0 < Kp; k0 += 32) {
#pragma unroll
    for (int i = 0; i < 2; ++i) { a[i] = an[i]; bq[i] = bn[i]; }
    if (k0 + 32 < Kp) {
#pragma unroll
      for (int i = 0; i < 2; ++i) { an[i] = load_a(i, k0 + 32); bn[i] = load_b(i, k0 + 32); }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = mfma_h(a[i], bq[j], acc[i][j]);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t m = m0 + i * 16 + lk * 4 + r;
      if (m >= M) continue;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = n0 + j * 16 + lj;
        if (n >= N) continue;
        float v = acc[i][j][r] + (bias ? bias[n] : 0.0f);
        if (act == 1) v = fmaxf(v, 0.0f);
        if (scale) v = fmaf(v, scale[n], shift[n]);
        C[m * N + n] = v;
      }
    }
}

// =========================================================================================
// sepconv_h_ftile: sepconv_h_kernel<3, MT> with the window rows shared through LDS -- the f16 twin of sepconv_ftile_kernel
// (model_fwd.hip).  A pixel of an octet plane is 16 bytes like a pixel of an f32 quad plane, so the scheme carries over unchanged:
// the 8 waves of a workgroup own 8 consecutive 64-pixel windows of the flat padded plane, the three rows of all of them are one
// contiguous range of 7 VAL + 64 + 2 WP pixels fetched once per octet by LDS-DMA in 1-KiB chunks (chunk c by wave c mod 8) into one of
// two slots, and every wave reads its three rows back with ds_read_b128.  One raw barrier per octet; counted vmcnt waits leave the
// depthwise-output store of the training forward in flight.  Same arithmetic in the same order as sepconv_h_kernel.
// =========================================================================================
// BNIN: the input planes hold the pre-normalisation tensor of a BatchNorm (+ ReLU): normalised where a row leaves LDS, zero outside the image (the
// planes' pads hold v = 0, but "same" padding pads y) -- y_a is never materialised by the training forward (orcai_h_sepconv_stats_bn).
template <int MT, bool XP, bool UOUT, bool STATS = false, bool BNIN = false>
__global__ __launch_bounds__(512) void sepconv_h_ftile_kernel(const h16* __restrict__ in /*[B][CO][HP][WP][8]*/, int Cin, int H, int W, int WP, int relu_in,
                                                              const h16* __restrict__ dw /*[CO][9][8]*/, const h16* __restrict__ pwf /*[KG][MT][64][8]*/,
                                                              const float* __restrict__ scale, const float* __restrict__ shift, int Cout, int relu_out,
                                                              void* __restrict__ out, int tasks, uint32_t magic_WP, int nchunk, h16* __restrict__ u_out,
                                                              double* __restrict__ shards = nullptr /*STATS: [32][ceil(Cout/8)][16]*/, InBnH ib = InBnH{}) {
  using orcai_lds::glds16;
  using orcai_lds::wait_vm_barrier;
  constexpr int NWV = 8, KK = 9, R = 1, lo = XP ? 2 : 1, VAL = 64 - 2 * lo;
  static_assert(!(XP && UOUT), "the training forward writes planes");
  static_assert(!STATS || !XP, "statistics epilogue: plane output");
  __shared__ float stat_s[STATS ? NWV : 1][4][STATS ? 8 * MT : 1];  // STATS: per wave and 16-lane row, the row's sums and sums of squares
  __shared__ __attribute__((aligned(16))) float inbn_s[BNIN ? 2 : 1][BNIN ? 64 : 4];  // BNIN: folded scale, shift per input channel (<= 64)
  extern __shared__ __attribute__((aligned(16))) h16x8 smem_hf[];
  const int CO = (Cin + 7) >> 3, COo = (Cout + 7) >> 3, KG = (CO + 3) >> 2;
  h16x8* rows_s = smem_hf;                    // [2][nchunk * 64] pixels
  h16x8* pw_s = smem_hf + 2 * nchunk * 64;    // [KG][MT][64]
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int bx, b;
  xcd_remap(bx, b);
  const int lk = lane >> 4, lj = lane & 15;
  const int plane = (H + 2 * R) * WP;
  const char* src = reinterpret_cast<const char*>(in) + (int64_t)b * CO * plane * 16;

  const int s0 = bx * NWV * VAL - lo;  // flat pixel of LDS position 0: lane 0 of the first window, one row up
  auto goff = [&](int c) {
    const int i = s0 + 64 * c + lane;
    return (uint32_t)(i < 0 ? 0 : (i >= plane ? plane - 1 : i)) * 16u;
  };
  const uint32_t off0 = goff(wave), off1 = goff(wave + NWV), off2 = goff(wave + 2 * NWV);
  const uint32_t lds0 = (uint32_t)(uintptr_t)rows_s;
  const int mine = wave < nchunk ? (wave + NWV < nchunk ? (wave + 2 * NWV < nchunk ? 3 : 2) : 1) : 0;
  auto issue = [&](int o) {
    const char* base = src + (int64_t)o * plane * 16;
    const uint32_t slot = lds0 + (uint32_t)((o & 1) * nchunk * 1024);
    if (mine > 0) glds16(base + off0, slot + (uint32_t)wave * 1024u);
    if (mine > 1) glds16(base + off1, slot + (uint32_t)(wave + NWV) * 1024u);
    if (mine > 2) glds16(base + off2, slot + (uint32_t)(wave + 2 * NWV) * 1024u);
  };
  issue(0);
  for (int i = threadIdx.x; i < KG * MT * 64; i += 64 * NWV) pw_s[i] = reinterpret_cast<const h16x8*>(pwf)[i];
  if (BNIN && threadIdx.x < 64) {
    const int ci = threadIdx.x, cc = ci < Cin ? ci : 0;
    const float sc = ib.gamma[cc] * rsqrtf(ib.var[cc] + ib.eps);  // exactly bn_planes_apply_h_kernel's arithmetic
    inbn_s[0][ci] = ci < Cin ? sc : 0.0f;
    inbn_s[1][ci] = ci < Cin ? ib.beta[cc] - ib.mean[cc] * sc : 0.0f;
  }
  __syncthreads();

  const int task = bx * NWV + wave;
  const bool wave_live = task < tasks;
  const int qbase = R * WP + task * VAL - lo;
  const int q = qbase + lane;
  int bn_row = 0;
  bool bn_colok = false;
  if (BNIN) {  // this lane's pixel of the flat padded plane: row / column once per window
    const int qq = q < 0 ? 0 : q;
    bn_row = (int)__umulhi((uint32_t)qq, magic_WP);
    bn_colok = q >= 0 && (qq - bn_row * WP) < W;
  }
  bool u_live = false;
  if (UOUT) {
    const int urow = (int)__umulhi((uint32_t)q, magic_WP);
    u_live = wave_live && lane >= lo && lane < 64 - lo && (q - urow * WP) < W && urow < R + H;
  }
  const bool u_any = UOUT && __builtin_amdgcn_readfirstlane((int)(__ballot(u_live) != 0ull)) != 0;
  const h16x8* rbase = rows_s + wave * VAL + lane;

  f32x4 acc[MT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int kg = 0; kg < KG; ++kg) {
    u32x4 d[4];
#pragma unroll
    for (int oo = 0; oo < 4; ++oo) {
      const int o = kg * 4 + oo;
      if (o < CO) {  // workgroup-uniform
        // outstanding, oldest first: this wave's DMAs of octet o, then (a wave that stores) the depthwise-output store of octet o - 1
        if (u_any && o > 0) wait_vm_barrier<1>(); else wait_vm_barrier<0>();
        if (o + 1 < CO) issue(o + 1);
        const h16x8* rs = rbase + (o & 1) * nchunk * 64;
        h16x8 cur[3] = {rs[0], rs[WP], rs[2 * WP]};
        if (BNIN) {
          const float4 s0 = reinterpret_cast<const float4*>(inbn_s[0])[2 * o], s1 = reinterpret_cast<const float4*>(inbn_s[0])[2 * o + 1];
          const float4 t0 = reinterpret_cast<const float4*>(inbn_s[1])[2 * o], t1 = reinterpret_cast<const float4*>(inbn_s[1])[2 * o + 1];
          const float sv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w}, tv[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
#pragma unroll
          for (int dy = 0; dy < 3; ++dy) {
            const bool ok = bn_colok && (bn_row + dy - 1) >= R && (bn_row + dy - 1) < R + H;  // padded-plane row of this lane's pixel, one up / same / one down
            h16x8 y;
#pragma unroll
            for (int k8 = 0; k8 < 8; ++k8) y[k8] = (h16)fmaxf(fmaf((float)cur[dy][k8], sv[k8], tv[k8]), 0.0f);  // f32 fma, ReLU, ONE rounding to f16: the stored y_a
            cur[dy] = ok ? y : zero_h();
          }
        }
        const h16x8 dd = relu_in ? dw_octet<3, true>(cur, dw + (int64_t)o * KK * 8) : dw_octet<3, false>(cur, dw + (int64_t)o * KK * 8);
        if (UOUT) {
          if (u_live) reinterpret_cast<h16x8*>(u_out)[((int64_t)b * CO + o) * plane + q] = dd;
        }
        d[oo] = as_u(dd);
      } else {
        d[oo] = (u32x4){0u, 0u, 0u, 0u};
      }
    }
    octets_to_fragments(d);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const h16x8 a = pw_s[(kg * MT + m) * 64 + lane];
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[m][t] = mfma_h(a, as_h(d[t]), acc[m][t]);
    }
  }
  if (!STATS && !wave_live) return;  // with the statistics epilogue every wave stays for its workgroup barrier (no barrier behind a divergent return); a wave without a window contributes zeros

  // ---- epilogue of sepconv_h_kernel, plane (0) and x-pooled (2) layouts
  float sc_r[MT][4], sh_r[MT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = m * 16 + lk * 4 + r;
      sc_r[m][r] = co < Cout ? scale[co] : 0.0f;
      sh_r[m][r] = co < Cout ? shift[co] : 0.0f;
    }
  const int Wx = (W + 1) >> 1, WPx = (Wx + 3) & ~3;
  float st[STATS ? 8 * MT : 1];  // STATS: [sum | sum of squares][m][r] over this lane's stored pixels (f32 values, before the f16 rounding)
  if (STATS) {
#pragma unroll
    for (int j = 0; j < 8 * MT; ++j) st[j] = 0.0f;
  }
#pragma unroll
  for (int tp = 0; tp < 4; tp += 2) {
    const int wl = 16 * (tp + (lk & 1)) + lj, flat = qbase + wl;
    const int row = (int)__umulhi((uint32_t)flat, magic_WP);
    const int x = flat - row * WP;
    const bool live = wave_live && wl >= lo && wl < 64 - lo && x < W && row < R + H;
    bool live0 = false, live1 = false;  // STATS: the pixels of column tiles tp and tp + 1 this lane holds BEFORE the octet exchange
    if (STATS) {
      const int w0 = 16 * tp + lj, w1 = w0 + 16, g0 = qbase + w0, g1 = qbase + w1;
      const int y0 = (int)__umulhi((uint32_t)g0, magic_WP), y1 = (int)__umulhi((uint32_t)g1, magic_WP);
      live0 = wave_live && w0 >= lo && w0 < 64 - lo && (g0 - y0 * WP) < W && y0 < R + H;
      live1 = wave_live && w1 >= lo && w1 < 64 - lo && (g1 - y1 * WP) < W && y1 < R + H;
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      float a[4], bq[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        a[r] = fmaf(acc[m][tp][r], sc_r[m][r], sh_r[m][r]);
        bq[r] = fmaf(acc[m][tp + 1][r], sc_r[m][r], sh_r[m][r]);
        if (relu_out) { a[r] = fmaxf(a[r], 0.0f); bq[r] = fmaxf(bq[r], 0.0f); }
        if (STATS) {
          const float l0 = live0 ? a[r] : 0.0f, l1 = live1 ? bq[r] : 0.0f;
          st[m * 4 + r] += l0 + l1;
          st[4 * MT + m * 4 + r] = fmaf(l1, l1, fmaf(l0, l0, st[4 * MT + m * 4 + r]));
        }
        if (XP) {
          const int f0 = qbase + 16 * tp + lj, f1 = f0 + 16;
          const int r0 = (int)__umulhi((uint32_t)f0, magic_WP), r1 = (int)__umulhi((uint32_t)f1, magic_WP);
          const float oa = __shfl_xor(a[r], 1, 64), ob = __shfl_xor(bq[r], 1, 64);
          if ((f0 - r0 * WP) + 1 < W) a[r] = fmaxf(a[r], oa);
          if ((f1 - r1 * WP) + 1 < W) bq[r] = fmaxf(bq[r], ob);
        }
      }
      float o8[8];
      tiles_to_octet(a, bq, o8);
      const int oq = 2 * m + (lk >> 1);
      if (!live || oq >= COo) continue;
      const h16x8 val = pack8(o8);
      if (!XP) reinterpret_cast<h16x8*>(out)[((int64_t)b * COo + oq) * plane + flat] = val;
      else if ((x & 1) == 0) reinterpret_cast<h16x8*>(out)[(((int64_t)b * COo + oq) * H + (row - R)) * WPx + (x >> 1)] = val;
    }
  }
  if (STATS) {
    // BatchNorm batch statistics of the tensor just written (the f32 path's scheme, model_fwd.hip sepconv_tile_kernel): inclusive DPP scan
    // over the 16 lanes of a row, rows of the 8 waves through LDS, one f64 atomic per value and workgroup into one of 32 accumulator copies
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
      for (int j = lane & 15; j < 8 * MT; j += 16) {  // j = [sum | sum of squares] * 4 MT + m * 4 + r -> channel m * 16 + g * 4 + r
        float tot = 0.0f;
#pragma unroll
        for (int w2 = 0; w2 < NWV; ++w2) tot += stat_s[w2][g][j];
        const int isq = j / (4 * MT), m = (j / 4) % MT, r = j & 3;
        const int octet = m * 2 + (g >> 1);
        if (octet < COo) atomicAdd(&shards[((((bx + b * 7) & 31) * COo) + octet) * 16 + isq * 8 + (g & 1) * 4 + r], (double)tot);
      }
    }
  }
}

struct SepArgsH {
  const h16 *in, *dw, *pwf;
  const float *scale, *shift;
  void* out;
  int B, Cin, H, W, WP, RP, Cout, relu_in, relu_out, out_layout, H2, WP2;
  h16* u_out;
  double* shards = nullptr;  // flat tiles with the depthwise-output store: BatchNorm statistics of the output in the epilogue
  InBnH ib;                  // ib.mean != nullptr: BatchNorm + ReLU of the input applied on load (with the statistics epilogue)
};

template <int KS, int MT>
int launch_sepconv_h(hipStream_t st, const SepArgsH& a) {
  const int lo = (a.out_layout == 2) ? ((KS / 2 + 1) & ~1) : KS / 2;  // x-pooled output: windows start on an even pixel
  const int VAL = 64 - 2 * lo;
  const int tasks = (a.H * a.WP + VAL - 1) / VAL;
  if ((int64_t)(a.H + 2 * a.RP) * a.WP >= (1ll << 29)) return ORCAI_E_UNSUPPORTED;
  if constexpr (KS == 3) {  // k = 3, plane or x-pooled output: rows shared through LDS (the f32 launcher's rule, orcai_sepconv_tile_mode)
    const int nchunk = (7 * VAL + 64 + 2 * a.WP + 63) / 64;
    const int KG = ((a.Cin + 7) / 8 + 3) / 4;
    const size_t lds = ((size_t)2 * nchunk * 64 + (size_t)KG * MT * 64) * 16;
    if (orcai_sepconv_tile_mode(-1) != 0 && a.RP == 1 && (a.out_layout == 0 || (a.out_layout == 2 && !a.u_out)) && nchunk <= 24 && lds <= 64 * 1024 &&
        (int64_t)((a.Cout + 7) / 8) * (a.H + 2) * a.WP < (1ll << 27)) {
      dim3 tgrid((tasks + 7) / 8, a.B);
#define ORCAI_HFTILE(XP, UOUT)                                                                                                                       \
  hipLaunchKernelGGL((sepconv_h_ftile_kernel<MT, XP, UOUT>), tgrid, dim3(512), lds, st, a.in, a.Cin, a.H, a.W, a.WP, a.relu_in, a.dw, a.pwf, a.scale, \
                     a.shift, a.Cout, a.relu_out, a.out, tasks, magic_for(a.WP), nchunk, a.u_out)
      if (a.out_layout == 2) ORCAI_HFTILE(true, false);
      else if (a.u_out && a.shards) {
        if (a.ib.mean)
          hipLaunchKernelGGL((sepconv_h_ftile_kernel<MT, false, true, true, true>), tgrid, dim3(512), lds, st, a.in, a.Cin, a.H, a.W, a.WP, 0, a.dw, a.pwf, a.scale, a.shift,
                             a.Cout, a.relu_out, a.out, tasks, magic_for(a.WP), nchunk, a.u_out, a.shards, a.ib);
        else
          hipLaunchKernelGGL((sepconv_h_ftile_kernel<MT, false, true, true>), tgrid, dim3(512), lds, st, a.in, a.Cin, a.H, a.W, a.WP, a.relu_in, a.dw, a.pwf, a.scale, a.shift,
                             a.Cout, a.relu_out, a.out, tasks, magic_for(a.WP), nchunk, a.u_out, a.shards);
      }
      else if (a.u_out) ORCAI_HFTILE(false, true);
      else ORCAI_HFTILE(false, false);
#undef ORCAI_HFTILE
      return (int)hipGetLastError();
    }
  }
  if (a.shards) return ORCAI_E_UNSUPPORTED;  // the statistics epilogue exists in the flat-tile kernel only
  dim3 grid((tasks + 3) / 4, a.B);
  hipLaunchKernelGGL((sepconv_h_kernel<KS, MT>), grid, dim3(256), 0, st, a.in, a.Cin, a.H, a.W, a.WP, a.relu_in, a.dw, a.pwf, a.scale, a.shift, a.Cout, a.relu_out,
                     a.out_layout, a.out, tasks, magic_for(a.WP), lo, a.RP, a.H2, a.WP2, a.u_out);
  return (int)hipGetLastError();
}

template <int KS>
int launch_sepconv_h_mt(hipStream_t st, const SepArgsH& a) {
  switch ((a.Cout + 15) / 16) {
    case 1: return launch_sepconv_h<KS, 1>(st, a);
    case 2: return launch_sepconv_h<KS, 2>(st, a);
    case 3: return launch_sepconv_h<KS, 3>(st, a);
    case 4: return launch_sepconv_h<KS, 4>(st, a);
    default: return ORCAI_E_UNSUPPORTED;
  }
}

}  // namespace

extern "C" {

int orcai_h_conv0_affine(const float* in, int64_t snippet_stride, int B, int H, int W, int ksize, const float* w, const float* scale, const float* shift, int relu,
                         void* out, void* stream) {
  if (!in || !w || !scale || !shift || !out || B <= 0 || H <= 0 || W <= 0) return ORCAI_E_BADARG;
  dim3 grid((W + 31) / 32, (H + 7) / 8, B);
  hipStream_t st = (hipStream_t)stream;
  const int WP = orcai_padded_width(W, ksize);
  h16* o = (h16*)out;
  switch (ksize) {
    case 3: hipLaunchKernelGGL((conv0_h_kernel<3, false>), grid, dim3(256), 0, st, in, snippet_stride, H, W, WP, w, scale, shift, o, relu, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0.0f, (h16*)nullptr); break;
    case 5: hipLaunchKernelGGL((conv0_h_kernel<5, false>), grid, dim3(256), 0, st, in, snippet_stride, H, W, WP, w, scale, shift, o, relu, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0.0f, (h16*)nullptr); break;
    case 7: hipLaunchKernelGGL((conv0_h_kernel<7, false>), grid, dim3(256), 0, st, in, snippet_stride, H, W, WP, w, scale, shift, o, relu, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0.0f, (h16*)nullptr); break;
    default: return ORCAI_E_UNSUPPORTED;
  }
  return (int)hipGetLastError();
}

int orcai_h_conv0_affine_bn(const float* in, int64_t snippet_stride, int B, int H, int W, int ksize, const float* w, const float* scale, const float* shift,
                            const float* bn_mean, const float* bn_var, const float* bn_gamma, const float* bn_beta, float bn_eps, void* v_out, void* y_out, void* stream) {
  if (!in || !w || !scale || !shift || !bn_mean || !bn_var || !bn_gamma || !bn_beta || !v_out || !y_out || B <= 0 || H <= 0 || W <= 0) return ORCAI_E_BADARG;
  dim3 grid((W + 31) / 32, (H + 7) / 8, B);
  hipStream_t st = (hipStream_t)stream;
  const int WP = orcai_padded_width(W, ksize);
  switch (ksize) {
    case 3: hipLaunchKernelGGL((conv0_h_kernel<3, true>), grid, dim3(256), 0, st, in, snippet_stride, H, W, WP, w, scale, shift, (h16*)v_out, 0, bn_mean, bn_var, bn_gamma, bn_beta, bn_eps, (h16*)y_out); break;
    case 5: hipLaunchKernelGGL((conv0_h_kernel<5, true>), grid, dim3(256), 0, st, in, snippet_stride, H, W, WP, w, scale, shift, (h16*)v_out, 0, bn_mean, bn_var, bn_gamma, bn_beta, bn_eps, (h16*)y_out); break;
    case 7: hipLaunchKernelGGL((conv0_h_kernel<7, true>), grid, dim3(256), 0, st, in, snippet_stride, H, W, WP, w, scale, shift, (h16*)v_out, 0, bn_mean, bn_var, bn_gamma, bn_beta, bn_eps, (h16*)y_out); break;
    default: return ORCAI_E_UNSUPPORTED;
  }
  return (int)hipGetLastError();
}

int orcai_h_sepconv(const void* in, int B, int Cin, int H, int W, int ksize_planes, int ktap, int relu_in, const void* dw, const void* pwf, const float* scale,
                    const float* shift, int Cout, int relu_out, int out_layout, int H2, int W2, void* out, void* u_out, void* stream) {
  if (!in || !dw || !pwf || !scale || !shift || !out || B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) return ORCAI_E_BADARG;
  if (Cout > 64 || Cin > 64 || (((uintptr_t)in | (uintptr_t)dw | (uintptr_t)pwf | (uintptr_t)out) & 15) || ktap > ksize_planes) return ORCAI_E_UNSUPPORTED;
  if (out_layout < 0 || out_layout > 3) return ORCAI_E_BADARG;
  if (out_layout == 3 && (H2 < 2 * H - 1 || W2 < 2 * W - 1)) return ORCAI_E_BADARG;
  SepArgsH a{(const h16*)in, (const h16*)dw, (const h16*)pwf, scale, shift, out, B, Cin, H, W, orcai_padded_width(W, ksize_planes), ksize_planes / 2, Cout, relu_in,
             relu_out, out_layout, H2, out_layout == 3 ? orcai_padded_width(W2, ksize_planes) : 0, (h16*)u_out};
  hipStream_t st = (hipStream_t)stream;
  switch (ktap) {
    case 1: return launch_sepconv_h_mt<1>(st, a);
    case 3: return launch_sepconv_h_mt<3>(st, a);
    case 5: return launch_sepconv_h_mt<5>(st, a);
    case 7: return launch_sepconv_h_mt<7>(st, a);
    default: return ORCAI_E_UNSUPPORTED;
  }
}

static int h_sepconv_stats_impl(const void* in, int B, int Cin, int H, int W, int relu_in, const void* dw, const void* pwf, const float* scale, const float* shift, int Cout,
                                void* out, void* u_out, double* shards, const InBnH& ib, void* stream) {
  if (!in || !dw || !pwf || !scale || !shift || !out || !u_out || !shards || B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) return ORCAI_E_BADARG;
  if (Cout > 64 || Cin > 64 || (((uintptr_t)in | (uintptr_t)dw | (uintptr_t)pwf | (uintptr_t)out | (uintptr_t)u_out) & 15) || B > 65535) return ORCAI_E_UNSUPPORTED;
  // the launcher's own conditions for the flat-tile kernel, checked BEFORE anything is touched
  const int WP = orcai_padded_width(W, 3), nchunk = (7 * 62 + 64 + 2 * WP + 63) / 64, KG = ((Cin + 7) / 8 + 3) / 4, MTv = (Cout + 15) / 16;
  const size_t lds = ((size_t)2 * nchunk * 64 + (size_t)KG * MTv * 64) * 16;
  if (orcai_sepconv_tile_mode(-1) == 0 || nchunk > 24 || lds + 4096 + 512 > 64 * 1024 /*+ the kernel's static statistics slots and folded input BatchNorm*/ || (int64_t)((Cout + 7) / 8) * (H + 2) * WP >= (1ll << 27) ||
      (int64_t)(H + 2) * WP >= (1ll << 29))
    return ORCAI_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int nacc = 16 * ((Cout + 7) / 8) * 32;
  {
    hipError_t e = orcai_zero::zero_async(shards, sizeof(double) * nacc, st);
    if (e != hipSuccess) return (int)e;
  }
  SepArgsH a{(const h16*)in, (const h16*)dw, (const h16*)pwf, scale, shift, out, B, Cin, H, W, WP, 1, Cout, relu_in, 0, 0, 0, 0, (h16*)u_out, shards};
  a.ib = ib;
  return launch_sepconv_h_mt<3>(st, a);
}

int orcai_h_sepconv_stats(const void* in, int B, int Cin, int H, int W, int relu_in, const void* dw, const void* pwf, const float* scale, const float* shift, int Cout,
                          void* out, void* u_out, double* shards, void* stream) {
  return h_sepconv_stats_impl(in, B, Cin, H, W, relu_in, dw, pwf, scale, shift, Cout, out, u_out, shards, InBnH{}, stream);
}

int orcai_h_sepconv_stats_bn(const void* v_in, int B, int Cin, int H, int W, const float* in_mean, const float* in_var, const float* in_gamma, const float* in_beta,
                             float in_eps, const void* dw, const void* pwf, const float* scale, const float* shift, int Cout, void* out, void* u_out, double* shards,
                             void* stream) {
  if (!in_mean || !in_var || !in_gamma || !in_beta) return ORCAI_E_BADARG;
  InBnH ib;
  ib.mean = in_mean; ib.var = in_var; ib.gamma = in_gamma; ib.beta = in_beta; ib.eps = in_eps;
  return h_sepconv_stats_impl(v_in, B, Cin, H, W, 0, dw, pwf, scale, shift, Cout, out, u_out, shards, ib, stream);
}

int orcai_h_pool_res_add(const void* s, const void* prev, int B, int C, int Cp, int H, int W, int ksize, const void* wrf, const float* br, void* out, int xpooled,
                         const float* bn_mean, const float* bn_var, const float* bn_gamma, const float* bn_beta, float bn_eps, void* stream) {
  if (!s || !prev || !wrf || !br || !out || B <= 0 || C <= 0 || Cp <= 0 || H <= 0 || W <= 0 || C > 64 || Cp > 64) return ORCAI_E_BADARG;
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  int tot_h = (Ho - 1) * 2 + 3 - H, tot_w = (Wo - 1) * 2 + 2 - W;
  if (tot_h < 0) tot_h = 0;
  if (tot_w < 0) tot_w = 0;
  if (xpooled & ~1) return ORCAI_E_BADARG;
  if (xpooled && tot_w / 2 != 0) return ORCAI_E_UNSUPPORTED;
  if (bn_mean && (xpooled || !bn_var || !bn_gamma || !bn_beta)) return ORCAI_E_BADARG;
  const int WP = orcai_padded_width(W, ksize), WPo = orcai_padded_width(Wo, ksize), R = ksize / 2;
  const int tasks = (Ho * WPo + 63) / 64;
  dim3 grid((tasks + 3) / 4, B);
  hipStream_t st = (hipStream_t)stream;
#define ORCAI_HPOOL(MT)                                                                                                                                        \
  hipLaunchKernelGGL(pool_res_add_h_kernel<MT>, grid, dim3(256), 0, st, (const h16*)s, (const h16*)prev, C, Cp, H, W, WP, R, Ho, Wo, WPo, tot_h / 2, tot_w / 2, \
                     (const h16*)wrf, br, (h16*)out, xpooled, tasks, magic_for(WPo), bn_mean, bn_var, bn_gamma, bn_beta, bn_eps)
  switch ((C + 15) / 16) {
    case 1: ORCAI_HPOOL(1); break;
    case 2: ORCAI_HPOOL(2); break;
    case 3: ORCAI_HPOOL(3); break;
    case 4: ORCAI_HPOOL(4); break;
    default: return ORCAI_E_UNSUPPORTED;
  }
#undef ORCAI_HPOOL
  return (int)hipGetLastError();
}

int orcai_h_gemm_bias_act(const float* A, const void* Wt, const float* bias, const float* scale, const float* shift, float* C, int64_t M, int N, int K, int act,
                          void* stream) {
  if (!A || !Wt || !C || M <= 0 || N <= 0 || K <= 0 || (scale && !shift)) return ORCAI_E_BADARG;
  if ((K & 3) || ((uintptr_t)A & 15) || ((uintptr_t)Wt & 15)) return ORCAI_E_UNSUPPORTED;  // 16-byte f32 loads along k
  const int Kp = (K + 31) & ~31;
  dim3 grid((N + 63) / 64, (unsigned)((M + 63) / 64));
  hipLaunchKernelGGL(gemm_h_kernel, grid, dim3(256), 0, (hipStream_t)stream, A, (const h16*)Wt, bias, scale, shift, C, M, N, K, Kp, act);
  return (int)hipGetLastError();
}

}  // extern "C"
