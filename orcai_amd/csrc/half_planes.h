// half_planes.h -- device helpers of the f16 path (BASELINE configs[4]: "fp16 MFMA path"): f16 channel-OCTET planes and the
// v_mfma_f32_16x16x32_f16 fragment plumbing shared by half_fwd.hip (inference / training forward) and half_bwd.hip (backward).
//
// Activation layout, f16:   [snippet][CO = ceil(C/8)][HP = H + 2R][WP = roundup4(W + R)][8]            R = k/2
//   the f32 path's padded channel-quad planes with 8 instead of 4 channels per pixel vector, so one pixel of one octet is still a
//   16-byte vector (a wave reading 64 consecutive pixels moves 1 KiB per instruction -- the "never < 16 B per lane" rule of
//   DESIGN.md 4.2) and half the HBM bytes per channel.  Pads are zero, written once by the host and never by a kernel; channels past
//   C inside the last octet are kept at zero.
//
// Contractions: D[cout][pixel] += W[cout][cin] * X[cin][pixel] on v_mfma_f32_16x16x32_f16 (f32 accumulate), 32 input channels
// (4 octets, a "K group") per instruction:
//   A (weights)      lane l holds W[row = l & 15][k = 8 (l >> 4) + e], e = 0..7   -- pre-packed per (K group, output tile, lane)
//   B (activations)  lane l holds X[k = 8 (l >> 4) + e][col = l & 15]             -- octet (l >> 4) of pixel (l & 15) of a 16-pixel tile
//   D                lane l holds D[row = 4 (l >> 4) + r][col = l & 15], r = 0..3  -- 4 consecutive output channels of one pixel
// The depthwise stage works with lane = pixel over 64 consecutive pixels (horizontal taps = DPP lane shifts), so the four octets of a K
// group -- four 16-byte registers per lane -- are turned into the four B fragments by a 4 x 4 transpose of 16-lane rows per dword
// (two v_permlane32_swap + two v_permlane16_swap), and two D tiles are turned into full output octets by one v_permlane16_swap per
// register: 16-byte stores again.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace orcai_half {

typedef _Float16 h16;
typedef h16 h16x2 __attribute__((ext_vector_type(2)));
typedef h16 h16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma_h(h16x8 a, h16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

// BatchNorm (+ ReLU) of a kernel's INPUT applied on load (BNIN): the planes hold the pre-normalisation tensor v, y = f16(relu(fma(v, s, t))) with
// s = gamma * rsqrt(var + eps), t = beta - mean * s is formed where a pixel is used -- the value bn_planes_apply_h_kernel would have stored.
struct InBnH {
  const float *mean = nullptr, *var = nullptr, *gamma = nullptr, *beta = nullptr;
  float eps = 0.0f;
};

inline uint32_t magic_for(uint32_t d) { return (uint32_t)((0x100000000ull + d - 1) / d); }  // __umulhi(n, magic) == n / d while n*d < 2^32

// XCD-aware block order (see model_fwd.hip): every XCD gets one contiguous band of the (snippet, window) space.
__device__ __forceinline__ void xcd_remap(int& bx, int& by) {
  const unsigned nbx = gridDim.x, total = nbx * gridDim.y;
  const unsigned L = blockIdx.y * nbx + blockIdx.x;
  const unsigned k = L & 7u, q = total >> 3, r = total & 7u;
  const unsigned Lp = k * q + (k < r ? k : r) + (L >> 3);
  by = (int)(Lp / nbx);
  bx = (int)(Lp - (unsigned)by * nbx);
}

__device__ __forceinline__ void swap32u(uint32_t& a, uint32_t& b) {  // a's lanes 32..63 <-> b's lanes 0..31
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = r[0];
  b = r[1];
}
__device__ __forceinline__ void swap16u(uint32_t& a, uint32_t& b) {  // a's odd 16-lane rows <-> b's even 16-lane rows
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  a = r[0];
  b = r[1];
}
__device__ __forceinline__ void swap16f(float& a, float& b) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
}

__device__ __forceinline__ u32x4 as_u(h16x8 v) { return __builtin_bit_cast(u32x4, v); }
__device__ __forceinline__ h16x8 as_h(u32x4 v) { return __builtin_bit_cast(h16x8, v); }

// d[o] = octet o of a K group, lane = pixel (64 consecutive pixels)  ->  d[t] = B fragment of 16-pixel tile t
// (lane (g, p): octet g of pixel 16 t + p).  Per dword: the 4 x 4 transpose of 16-lane rows of the f32 path.
__device__ __forceinline__ void octets_to_fragments(u32x4 (&d)[4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    uint32_t x0 = d[0][j], x1 = d[1][j], x2 = d[2][j], x3 = d[3][j];
    swap32u(x0, x2);
    swap32u(x1, x3);
    swap16u(x0, x1);
    swap16u(x2, x3);
    d[0][j] = x0; d[1][j] = x1; d[2][j] = x2; d[3][j] = x3;
  }
}

// value of lane (l + SH), SH in [-3, 3], for all four dwords of an octet register (lanes shifted in from outside the wave: don't care)
template <int SH>
__device__ __forceinline__ uint32_t lane_shift_u(uint32_t v) {
  if constexpr (SH == 0) return v;
  else if constexpr (SH < 0) return lane_shift_u<SH + 1>(__builtin_amdgcn_update_dpp(0u, v, 0x138 /*wave_shr:1*/, 0xf, 0xf, true));
  else return lane_shift_u<SH - 1>(__builtin_amdgcn_update_dpp(0u, v, 0x130 /*wave_shl:1*/, 0xf, 0xf, true));
}
template <int SH>
__device__ __forceinline__ h16x8 lane_shift_h(h16x8 v) {
  u32x4 u = as_u(v);
#pragma unroll
  for (int j = 0; j < 4; ++j) u[j] = lane_shift_u<SH>(u[j]);
  return as_h(u);
}

__device__ __forceinline__ h16x8 zero_h() { return (h16x8){0, 0, 0, 0, 0, 0, 0, 0}; }
__device__ __forceinline__ h16x8 relu_h(h16x8 a) { return __builtin_elementwise_max(a, zero_h()); }
__device__ __forceinline__ h16x8 max_h(h16x8 a, h16x8 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ h16x8 min_h(h16x8 a, h16x8 b) { return __builtin_elementwise_min(a, b); }

__device__ __forceinline__ h16x8 pack8(const float (&v)[8]) {
  return (h16x8){(h16)v[0], (h16)v[1], (h16)v[2], (h16)v[3], (h16)v[4], (h16)v[5], (h16)v[6], (h16)v[7]};
}
__device__ __forceinline__ void unpack8(h16x8 h, float (&v)[8]) {
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (float)h[e];
}

// Two D tiles (m, t) and (m, t + 1) -- a[r] / b[r] = output channels 16 m + 4 lk + r of pixel (16 t + lj) / (16 (t + 1) + lj) -- to one
// full octet per lane: afterwards o[0..7] = channels 8 (2 m + (lk >> 1)) + e of pixel 16 (t + (lk & 1)) + lj.
__device__ __forceinline__ void tiles_to_octet(const float (&a)[4], const float (&b)[4], float (&o)[8]) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float x = a[r], y = b[r];
    swap16f(x, y);
    o[r] = x;
    o[4 + r] = y;
  }
}

}  // namespace orcai_half
