// frontend.hip -- STFT / dB / exact quantile clip / normalise for gfx950 (MI355X).
//
// Replaces the arithmetic of /root/reference/src/orcAI/spectrogram.py:34-39 (librosa.stft),
// :51-53 (amplitude_to_db(ref=np.max)) and :58-87 (preprocess_spectrogram).
//
// Data flow (one recording):
//   pass A  stft_db_kernel      PCM f32 -> L[t][k] = 10*log10(max(|X|^2,1e-10)), k < k_crop   (HBM: R pcm, W L)
//                               + max |X|^2 over all 257 bins (atomicMax) + level-1 histogram of L (LDS)
//   scan1   select_scan1_kernel level-1 histogram -> key interval of each wanted rank
//   pass C  hist2_kernel        re-read L, histogram the keys inside the two intervals (<= 2 rounds)
//   scan2   select_scan2_kernel interval -> exact key
//   final   clip_normalize      in place: max(L - ref_db, -80) -> clip -> (v - lo)/(hi - lo)     (HBM: R L, W out)
//
// The order statistic is selected on the un-referenced values L: v = max(L - ref_db, -80) is monotone
// non-decreasing in L, so the k-th smallest v is that function of the k-th smallest L (exact).
//
// STFT kernel layout: a wavefront (64 lanes) transforms 4 consecutive frames at once, 16 lanes per
// frame.  A 512-point real frame is packed as 256 complex points z[n] = x[2n] + i x[2n+1] and
// transformed as 16 x 16 (four-step): FFT-16 in registers over n2 (n = n1 + 16 n2, n1 = lane),
// twiddle W256^(n1 k2), 16x16 transpose through a wave-private padded LDS tile, FFT-16 in registers
// over n1, then the real-FFT split with the mirrored bin fetched by ds_bpermute.  The 4 frames' 171
// kept bins are staged in the same LDS tile and leave as one contiguous 2736-byte run of dwordx4 stores.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <type_traits>
#include <cstring>
#include <mutex>

#include "orcai_hip.h"
#include "zero_fill.h"

namespace {

constexpr int NFFT = 512;
constexpr int NB1 = 2561;    // level-1 buckets: 1280 negative + 1 around zero + 1280 positive
constexpr int NB1_PAD = 2568;
constexpr int NB2 = 65536;   // level-2 bins per round
constexpr int NB2C = NB2 / 256;
constexpr float AMIN_POW = 1e-10f;
constexpr float DB_PER_LOG2 = 3.0102999566398120f;  // 10*log10(2)

struct SelState {
  uint32_t klo;       // first key of the current interval
  uint32_t done;      // 1 once the interval is a single key
  uint64_t width;     // number of keys in the interval
  uint64_t rank;      // 0-based rank of the wanted element inside the interval
};

struct Workspace {
  uint32_t hist1[NB1_PAD];
  uint32_t hist2[2][NB2];
  uint32_t hist2c[2][NB2C];  // coarse: one counter per 256 fine bins
  SelState sel[2];
  uint32_t pmax_bits;  // max |X|^2 (non-negative float, so uint order == float order)
  uint32_t use_ref;
  float ref_db;        // 10*log10(max(pmax, 1e-10))
  float floor_db;      // -top_db
  float p_lo, p_hi;    // clip bounds in final dB
  float sel_raw[2];    // selected raw order statistics
};

__device__ float g_window[NFFT];
__device__ float2 g_tw256[256];  // [k2*16 + n1] = exp(-2 pi i n1 k2 / 256)
__device__ float2 g_tw512[260];  // [k] = (cos, sin)(2 pi k / 512), k = 0..256

__device__ __forceinline__ uint32_t f2key(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(uint32_t k) {
  uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(u);
}
__device__ __forceinline__ float power_to_db(float p) {
  return __builtin_amdgcn_logf(fmaxf(p, AMIN_POW)) * DB_PER_LOG2;
}
// Monotone map key -> level-1 bucket: 128 buckets per octave for 0.125 <= |x| < 128, one bucket around 0.
__device__ __forceinline__ int bucket1(uint32_t key) {
  // k16 <  0x3D00 -> 0;  0x3D00..0x41FF -> k16 - 0x3D00;  0x4200..0xBDFF -> 1280;  0xBE00..0xC2FF -> 1281 + (k16 - 0xBE00);  above -> 2560
  const int k16 = (int)(key >> 16);
  const int lo = min(max(k16 - 0x3D00, 0), 1280);
  const int hi = 1281 + min(k16 - 0xBE00, 0x4FF);
  return k16 >= 0xBE00 ? hi : lo;
}
// The same bucket straight from the float (bucket1(f2key(f)) == bucket1f(f) for every float; tests/test_host_logic.py restates both and checks all 2^16
// high halves): magnitude bits m16 = bits[30:16], t = clamp(m16 - 0x3DFF, 0, 0x500) counts 1/128-octave steps above 0.125 (0 below),
// bucket = 1280 +- t by the sign.  7 integer instructions instead of 13 in the STFT kernel's per-bin tail.
__device__ __forceinline__ int bucket1f(float f) {
  const int u = (int)__float_as_uint(f);
  const int sgn = u >> 31;  // 0 or -1
  const int m16 = (int)(((uint32_t)u >> 16) & 0x7FFFu);  // v_bfe_u32
  const int t = min(max(m16 - 0x3DFF, 0), 0x500);         // v_med3_i32
  return 1280 + ((t ^ sgn) - sgn);
}
__device__ __forceinline__ void bucket1_range(int b, uint32_t& klo, uint64_t& width) {
  if (b == 0) { klo = 0u; width = 0x3D010000ull; }
  else if (b < 1280) { klo = (0x3D00u + (uint32_t)b) << 16; width = 65536ull; }
  else if (b == 1280) { klo = 0x42000000u; width = 0xBE000000ull - 0x42000000ull; }
  else if (b < 2560) { klo = (0xBE00u + (uint32_t)(b - 1281)) << 16; width = 65536ull; }
  else { klo = 0xC2FF0000u; width = 0x100000000ull - 0xC2FF0000ull; }
}
__device__ __forceinline__ int shift_for(uint64_t width) {
  int s = 0;
  while ((width >> s) > (uint64_t)NB2) ++s;
  return s;
}

// ---------------------------------------------------------------- FFT-16 in registers, packed complex arithmetic
// A complex number is one aligned register pair (re, im); every butterfly add / subtract is ONE v_pk_add_f32, a multiplication by
// -i or +i is folded into the add that consumes it through the operand selectors of VOP3P (op_sel / op_sel_hi pick which half of a
// source feeds which half of the result, neg_lo / neg_hi negate it), and a complex multiplication is v_pk_mul_f32 + v_pk_fma_f32.
// Written as inline asm: left to itself the compiler keeps re[] / im[] apart and spends a fifth of the kernel's vector instructions
// on register moves that assemble packed operands.
typedef float c2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ c2 add_mi(c2 a, c2 b) {  // a - i b = (a.re + b.im, a.im - b.re)
  c2 r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ c2 add_pi(c2 a, c2 b) {  // a + i b = (a.re - b.im, a.im + b.re)
  c2 r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ c2 cmulv(c2 a, c2 w) {  // a * w, w = (wr, wi) in a VGPR pair
  c2 t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));                                  // (a.re wr, a.im wr)
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));  // (-a.im wi + ., a.re wi + .)
  return r;
}
__device__ __forceinline__ c2 cmuls(c2 a, c2 w) {  // the same with a wave-uniform constant in an SGPR pair
  c2 t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "s"(w));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "s"(w), "v"(t));
  return r;
}
__device__ __forceinline__ void bfly4(c2& a, c2& b, c2& c, c2& d) {
  const c2 t0 = a + c, t1 = a - c, t2 = b + d, t3 = b - d;
  a = t0 + t2;
  c = t0 - t2;
  b = add_mi(t1, t3);
  d = add_pi(t1, t3);
}
// In-place forward FFT-16; output bin k ends up at position P16(k).
#define P16(k) (4 * ((k) & 3) + ((k) >> 2))
__device__ __forceinline__ void fft16(c2 (&x)[16]) {
  constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R2 = 0.70710678118654752f;
#pragma unroll
  for (int a = 0; a < 4; ++a) bfly4(x[a], x[a + 4], x[a + 8], x[a + 12]);
  // position a + 4q holds y[a][q]; multiply by W16^(a q)
  x[1 + 4] = cmuls(x[1 + 4], (c2){C1, -S1});     // a=1,q=1 : W^1
  x[1 + 8] = cmuls(x[1 + 8], (c2){R2, -R2});     // a=1,q=2 : W^2
  x[1 + 12] = cmuls(x[1 + 12], (c2){S1, -C1});   // a=1,q=3 : W^3
  x[2 + 4] = cmuls(x[2 + 4], (c2){R2, -R2});     // a=2,q=1 : W^2
  x[2 + 8] = add_mi((c2){0.f, 0.f}, x[2 + 8]);   // a=2,q=2 : W^4 = -i
  x[2 + 12] = cmuls(x[2 + 12], (c2){-R2, -R2});  // a=2,q=3 : W^6
  x[3 + 4] = cmuls(x[3 + 4], (c2){S1, -C1});     // a=3,q=1 : W^3
  x[3 + 8] = cmuls(x[3 + 8], (c2){-R2, -R2});    // a=3,q=2 : W^6
  x[3 + 12] = cmuls(x[3 + 12], (c2){-C1, S1});   // a=3,q=3 : W^9
#pragma unroll
  for (int q = 0; q < 4; ++q) bfly4(x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]);
}

constexpr int ROW_BYTES = 144;               // 16 complex (128 B) + 16 B pad: conflict-free b128 row reads
constexpr int FRAME_BYTES = 16 * ROW_BYTES;  // 2304
constexpr int WAVE_BYTES = 4 * FRAME_BYTES;  // 9216, also holds the 4 x k_crop output staging (<= 4112 B)

int g_stft_blocks = 1536;  // persistent workgroups of the STFT launch (orcai_stft_blocks): six per compute unit, three resident (profiles/r03_ab_stft_blocks.log: 768 -> 1536 = -3 %)

struct __attribute__((aligned(16))) StftLds {
  unsigned char tile[4][WAVE_BYTES];  // per wave: PCM staging (5 hops) -> 16x16 transpose tile -> output staging
  float2 tw256[256];
  float2 tw512[260];
  float2 win2[256];                   // (w[2n], w[2n+1])
  uint32_t hist[NB1_PAD];
};

__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// FAST_ONLY: every group in [g_begin, g_end) is known (by the launcher) to satisfy the fast-path conditions for all four of its
// waves, so the generic body is not even instantiated (134 VGPRs and no spills instead of 168 + spills).  KC > 0: k_crop is the
// compile-time constant KC (171 for orcai-V1), which removes the per-bin crop branches from the real-FFT split.
template <bool EVEN_HOP, bool FAST_ONLY, int KC>
__global__ __launch_bounds__(256, 3) void stft_db_kernel(const float* __restrict__ pcm, int64_t n_samples, int hop, int64_t n_frames,
                                                       int k_crop_arg, float* __restrict__ out, Workspace* __restrict__ ws, int64_t g_begin,
                                                       int64_t g_end) {
  const int k_crop = KC > 0 ? KC : k_crop_arg;
  __shared__ StftLds lds;
  const int tid = threadIdx.x;
  const int wave = tid >> 6;
  const int lane = tid & 63;
  const int l16 = lane & 15;  // n1, later k2
  const int fsub = lane >> 4; // frame within the wave

  for (int i = tid; i < 256; i += 256) lds.tw256[i] = g_tw256[i];
  for (int i = tid; i < 257; i += 256) lds.tw512[i] = g_tw512[i];
  for (int i = tid; i < NB1_PAD; i += 256) lds.hist[i] = 0u;

  for (int i = tid; i < 256; i += 256) lds.win2[i] = *reinterpret_cast<const float2*>(&g_window[2 * i]);
  __syncthreads();

  unsigned char* my_tile = lds.tile[wave];
  unsigned char* my_frame = my_tile + fsub * FRAME_BYTES;
  float pmax = 0.0f;

  // One group of 4 frames per wave.  FAST (compile-time): hop 256, all 4 frames inside the recording, the 1280 samples they
  // cover inside the PCM array and 16-byte aligned -> no per-element predicates anywhere in the body.
  auto process = [&](auto fast_tag, int64_t t0w, const float4& p0, const float4& p1, const float4& p2, const float4& p3, const float4& p4 /*FAST_ONLY: this group's 5 x 16 B per lane, requested one group ahead*/) {
    constexpr bool FAST = decltype(fast_tag)::value;
    const int64_t t = t0w + fsub;
    const bool valid = FAST ? true : (t < n_frames);
    const int64_t s0 = t * (int64_t)hop - (NFFT / 2);

    c2 z[16];
    // The wave's 4 frames cover padded samples [hop*t0w, hop*(t0w+3) + 512): for the default hop 256 that is one
    // contiguous run of 1280 floats, fetched as 5 dwordx4 per lane (1 KiB per instruction) into the wave's tile.
    const int64_t w0 = t0w * (int64_t)hop - (NFFT / 2);  // first sample the wave needs
    if constexpr (FAST) {
      float* st = reinterpret_cast<float*>(my_tile);
      const float4* gp = reinterpret_cast<const float4*>(pcm + w0);
      if constexpr (FAST_ONLY) {
        reinterpret_cast<float4*>(st)[lane] = p0;
        reinterpret_cast<float4*>(st)[lane + 64] = p1;
        reinterpret_cast<float4*>(st)[lane + 128] = p2;
        reinterpret_cast<float4*>(st)[lane + 192] = p3;
        reinterpret_cast<float4*>(st)[lane + 256] = p4;
      } else {
#pragma unroll
        for (int u = 0; u < 5; ++u) reinterpret_cast<float4*>(st)[lane + 64 * u] = gp[lane + 64 * u];
      }
      wave_lds_fence();
      const float* fr = st + fsub * 256;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const c2 v = *reinterpret_cast<const c2*>(fr + 2 * (l16 + 16 * j));
        const c2 w = *reinterpret_cast<const c2*>(&lds.win2[l16 + 16 * j]);
        z[j] = v * w;  // (x[2n] w[2n], x[2n+1] w[2n+1]): the 512-point real frame packed as 256 complex points
      }
      wave_lds_fence();  // staging reads done before the tile is reused for the transpose
    } else {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int64_t sidx = s0 + 2 * (l16 + 16 * j);
        const float x0 = (valid && sidx >= 0 && sidx < n_samples) ? pcm[sidx] : 0.0f;
        const float x1 = (valid && sidx + 1 >= 0 && sidx + 1 < n_samples) ? pcm[sidx + 1] : 0.0f;
        const float2 w = lds.win2[l16 + 16 * j];
        z[j] = (c2){x0 * w.x, x1 * w.y};
      }
    }

    // step 1: FFT-16 over n2 -> Y[n1][k2] at position P16(k2); step 2: twiddle W256^(n1 k2)
    fft16(z);
#pragma unroll
    for (int k2 = 1; k2 < 16; ++k2) z[P16(k2)] = cmulv(z[P16(k2)], *reinterpret_cast<const c2*>(&lds.tw256[k2 * 16 + l16]));
    // step 3: transpose through the wave-private tile: row k2, column n1
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) *reinterpret_cast<c2*>(my_frame + k2 * ROW_BYTES + l16 * 8) = z[P16(k2)];
    wave_lds_fence();
#pragma unroll
    for (int h = 0; h < 8; ++h) {
      const float4 v = *reinterpret_cast<const float4*>(my_frame + l16 * ROW_BYTES + h * 16);
      z[2 * h] = (c2){v.x, v.y};
      z[2 * h + 1] = (c2){v.z, v.w};
    }
    // step 4: FFT-16 over n1 -> Z[16 k1 + k2] at position P16(k1), lane = k2
    fft16(z);
    wave_lds_fence();  // all lanes have read the tile before it is reused as output staging

    // step 5: real-FFT split.  A = Z[k], B = Z[256-k] (lane (16-k2)&15, register 15-k1 for k2 != 0, (16-k1)&15 for k2 == 0): every lane
    // parks its 16 bins in the wave's tile ([frame][k] complex, 2 KiB + 128 B per frame) and reads the mirrored ones back with a per-lane address --
    // 16 ds_write_b64 + 16 ds_read_b64 (both conflict-free: the 16 lanes of a frame touch 16 consecutive bins) where round 2 had 32
    // ds_bpermute + 32 selects for the k2 == 0 lane.
    //   2 X[k] = (A + conj B) + (c - i s)(-i)(A - conj B)        with (c, s) = tw512[k]
    // in packed form: E = A + conj B, O = (A.im + B.im, B.re - A.re), 2 X = E + (c O.re + s O.im, c O.im - s O.re); power = |2X|^2 / 4:
    // the 1/4 is applied once to the running maximum and, in the dB value, to the argument of the logarithm's clamp (0.25 s >= 1e-10
    // <=> s >= 4e-10) and as log2(s) - 2 (exact scaling by a power of two; the sum rounds like log2(0.25 s) up to one ulp).
    c2* zt = reinterpret_cast<c2*>(my_tile + fsub * 2176);  // 257 bins (Z[256] = Z[0]) + pad: frames 0 / 1 (and 2 / 3) of a half-wave land in different bank halves
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) zt[16 * k1 + l16] = z[P16(k1)];
    if (l16 == 0) zt[256] = z[P16(0)];  // so that bin 256 - k needs no wrap: the mirrored address is one per-lane base + a compile-time offset
    wave_lds_fence();
    const c2* zm = zt + (16 - l16);  // Z[256 - (16 k1 + k2)] = zm[16 (15 - k1)]
    float Lk[(KC > 0 ? (KC + 15) / 16 : 16)];
    const c2 z0 = z[P16(0)];
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) {
      const c2 A = z[P16(k1)];
      const int k = 16 * k1 + l16;
      const c2 Bv = zm[16 * (15 - k1)];
      const c2 cs = *reinterpret_cast<const c2*>(&lds.tw512[k]);
      c2 E, O, P, T;
      asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(E) : "v"(A), "v"(Bv));                                              // (A.re + B.re, A.im - B.im)
      asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0] neg_hi:[1,0]" : "=v"(O) : "v"(A), "v"(Bv));                 // (A.im + B.im, B.re - A.re)
      asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(T) : "v"(O), "v"(cs));                              // (c O.re, c O.im)
      asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(P) : "v"(O), "v"(cs), "v"(T));  // (s O.im + ., -s O.re + .)
      const c2 X2 = E + P;
      const float p4 = fmaf(X2.x, X2.x, X2.y * X2.y);  // 4 |X|^2
      if (valid) pmax = fmaxf(pmax, p4);
      if (KC > 0 ? (16 * k1 < KC) : true) Lk[KC > 0 ? k1 : k1] = (__builtin_amdgcn_logf(fmaxf(p4, 4.0f * AMIN_POW)) - 2.0f) * DB_PER_LOG2;
    }
    float Lnyq = 0.0f;
    if (l16 == 0) {  // Nyquist bin 256: X = Re Z0 - Im Z0
      const float x = z0.x - z0.y;
      const float p4 = 4.0f * x * x;
      if (valid) pmax = fmaxf(pmax, p4);
      Lnyq = (__builtin_amdgcn_logf(fmaxf(p4, 4.0f * AMIN_POW)) - 2.0f) * DB_PER_LOG2;
    }
    wave_lds_fence();  // every mirrored read is done before the tile becomes the output staging
    float* stage = reinterpret_cast<float*>(my_tile) + fsub * k_crop;
#pragma unroll
    for (int k1 = 0; k1 < (KC > 0 ? (KC + 15) / 16 : 16); ++k1) {
      const int k = 16 * k1 + l16;
      if (k < k_crop) {
        stage[k] = Lk[k1];
        if (valid) atomicAdd(&lds.hist[bucket1f(Lk[k1])], 1u);
      }
    }
    if (l16 == 0 && 256 < k_crop) {
      stage[256] = Lnyq;
      if (valid) atomicAdd(&lds.hist[bucket1f(Lnyq)], 1u);
    }
    wave_lds_fence();

    // step 6: the wave's valid frames are one contiguous run of out[]
    int64_t nv = n_frames - t0w;
    nv = nv < 0 ? 0 : (nv > 4 ? 4 : nv);
    const int count = (int)nv * k_crop;
    float* dst = out + t0w * (int64_t)k_crop;  // 16-byte aligned: t0w % 4 == 0
    const float* src = reinterpret_cast<const float*>(my_tile);
    const int n4 = count >> 2;
    for (int i = lane; i < n4; i += 64) reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(src)[i];
    for (int i = (n4 << 2) + lane; i < count; i += 64) dst[i] = src[i];
    // The staging tile may be overwritten once its reads have returned (the stores carry their data in registers).  An LDS-only wait:
    // wave_lds_fence() here made every group end by waiting for its own output stores to be acknowledged (a release fence drains vmcnt).
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  };
  if constexpr (FAST_ONLY) {
    // The PCM of the NEXT group is requested before the current one is transformed (a wave otherwise starts every group by waiting a full
    // HBM round trip with only two other waves of its SIMD to cover it): two register sets, the loop unrolled by two so that no set with
    // loads in flight is ever moved; the request past the last group is clamped to the last one (unconditional, never used).
    const int64_t G = gridDim.x, g_last = g_end - 1;
    auto base = [&](int64_t g) {
      const int64_t gc = g < g_end ? g : g_last;
      return reinterpret_cast<const float4*>(pcm + (((gc << 4) + (wave << 2)) * (int64_t)hop - (NFFT / 2))) + lane;
    };
    int64_t g = g_begin + blockIdx.x;
    if (g < g_end) {
      const float4* gp = base(g);
      float4 a0 = gp[0], a1 = gp[64], a2 = gp[128], a3 = gp[192], a4 = gp[256];
      while (true) {
        gp = base(g + G);
        const float4 b0 = gp[0], b1 = gp[64], b2 = gp[128], b3 = gp[192], b4 = gp[256];
        process(std::true_type{}, (g << 4) + (wave << 2), a0, a1, a2, a3, a4);
        g += G;
        if (g >= g_end) break;
        gp = base(g + G);
        a0 = gp[0]; a1 = gp[64]; a2 = gp[128]; a3 = gp[192]; a4 = gp[256];
        process(std::true_type{}, (g << 4) + (wave << 2), b0, b1, b2, b3, b4);
        g += G;
        if (g >= g_end) break;
      }
    }
  } else {
    for (int64_t g = g_begin + blockIdx.x; g < g_end; g += gridDim.x) {
      const int64_t t0w = (g << 4) + (wave << 2);  // first frame of this wave
      const int64_t w0 = t0w * (int64_t)hop - (NFFT / 2);
      const bool fast = EVEN_HOP && hop == 256 && w0 >= 0 && ((w0 & 3) == 0) && (w0 + 1280 <= n_samples) && (t0w + 3 < n_frames);
      const float4 none = make_float4(0.f, 0.f, 0.f, 0.f);
      if (fast) process(std::true_type{}, t0w, none, none, none, none, none);  // wave-uniform
      else process(std::false_type{}, t0w, none, none, none, none, none);
    }
  }

  // wave max -> one atomic per wave
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) pmax = fmaxf(pmax, __shfl_xor(pmax, off, 64));
  if (lane == 0) atomicMax(&ws->pmax_bits, __float_as_uint(0.25f * pmax));  // the loop tracked 4 |X|^2 (exact scaling)
  __syncthreads();
  for (int i = tid; i < NB1; i += 256) {
    uint32_t c = lds.hist[i];
    if (c) atomicAdd(&ws->hist1[i], c);
  }
}

// ---------------------------------------------------------------- STFT + dB for any power-of-two transform size (the reference reads nfft
// from the parameter file, spectrogram.py:34-39; every shipped file says 512 and runs stft_db_kernel above).  Plain and slow by comparison: one
// workgroup per frame, the windowed frame (centred, zero padding, periodic Hann: librosa.stft's defaults) bit-reversed into LDS, log2(nfft) radix-2
// stages with a barrier each, twiddles by sincospif; L for the first k_crop bins, max |X|^2 over ALL 1 + nfft/2 bins.  The level-1 histogram is
// taken by hist1_kernel in a second pass over L (the 512-point kernel builds it on the fly).
constexpr int NFFT_ANY_MAX = 4096;

__global__ __launch_bounds__(256) void stft_any_kernel(const float* __restrict__ pcm, int64_t n_samples, int n_fft, int log2n, int hop, int64_t n_frames, int k_crop,
                                                        float* __restrict__ out_db, Workspace* __restrict__ ws) {
  __shared__ float2 buf[NFFT_ANY_MAX];
  __shared__ float wmax[4];
  const int tid = threadIdx.x;
  const int half = n_fft >> 1;
  float pmax = 0.0f;
  for (int64_t t = blockIdx.x; t < n_frames; t += gridDim.x) {
    const int64_t s0 = t * hop - half;
    for (int n = tid; n < n_fft; n += 256) {
      const int64_t i = s0 + n;
      const float x = (i >= 0 && i < n_samples) ? pcm[i] : 0.0f;
      const float w = 0.5f - 0.5f * cospif(2.0f * (float)n / (float)n_fft);
      buf[__brev((uint32_t)n) >> (32 - log2n)] = make_float2(x * w, 0.0f);
    }
    __syncthreads();
    for (int st = 1; st <= log2n; ++st) {
      const int hm = 1 << (st - 1);
      for (int j = tid; j < half; j += 256) {
        const int k = j & (hm - 1);
        const int i0 = ((j >> (st - 1)) << st) + k, i1 = i0 + hm;
        float sn, cs;
        sincospif(-(float)k / (float)hm, &sn, &cs);  // exp(-2 pi i k / (2 hm))
        const float2 a = buf[i0], b = buf[i1];
        const float2 tb = make_float2(fmaf(b.x, cs, -b.y * sn), fmaf(b.x, sn, b.y * cs));
        buf[i0] = make_float2(a.x + tb.x, a.y + tb.y);
        buf[i1] = make_float2(a.x - tb.x, a.y - tb.y);
      }
      __syncthreads();
    }
    for (int k = tid; k <= half; k += 256) {
      const float2 z = buf[k];
      const float p = fmaf(z.x, z.x, z.y * z.y);
      pmax = fmaxf(pmax, p);
      if (k < k_crop) out_db[t * (int64_t)k_crop + k] = power_to_db(p);
    }
    __syncthreads();
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) pmax = fmaxf(pmax, __shfl_xor(pmax, o, 64));
  if ((tid & 63) == 0) wmax[tid >> 6] = pmax;
  __syncthreads();
  if (tid == 0) atomicMax(&ws->pmax_bits, __float_as_uint(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]))));
}

// ---------------------------------------------------------------- STFT + dB for ANY transform size 2 .. 4096 (odd, 500, 1000 ...: the reference passes
// whatever the parameter file says to librosa.stft, spectrogram.py:34-39; no shipped file uses one).  A direct DFT, O(nfft^2) per frame: one workgroup per
// frame, the windowed frame and the nfft twiddles exp(-2 pi i j / nfft) as float64 in dynamic LDS (96 KB at 4096), thread = bin, the twiddle index stepped
// by k modulo nfft, float64 accumulation (numpy's rfft, which librosa calls, is float64 too).  Completeness, not speed: ~1 s per hour of audio at nfft 1000.
__global__ __launch_bounds__(256) void stft_dft_kernel(const float* __restrict__ pcm, int64_t n_samples, int n_fft, int hop, int64_t n_frames, int k_crop,
                                                        float* __restrict__ out_db, Workspace* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) double dft_lds[];  // [n_fft] frame, then [n_fft][2] twiddles
  __shared__ float wmax[4];
  double* xw = dft_lds;
  double2* tw = reinterpret_cast<double2*>(dft_lds + n_fft + (n_fft & 1));
  const int tid = threadIdx.x;
  const int half = n_fft >> 1;
  for (int j = tid; j < n_fft; j += 256) {
    double sn, cs;
    sincospi(2.0 * (double)j / (double)n_fft, &sn, &cs);
    tw[j] = make_double2(cs, -sn);
  }
  __syncthreads();
  float pmax = 0.0f;
  for (int64_t t = blockIdx.x; t < n_frames; t += gridDim.x) {
    const int64_t s0 = t * hop - half;
    for (int n = tid; n < n_fft; n += 256) {
      const int64_t i = s0 + n;
      const double x = (i >= 0 && i < n_samples) ? (double)pcm[i] : 0.0;
      xw[n] = x * (0.5 - 0.5 * tw[n].x);  // periodic Hann
    }
    __syncthreads();
    for (int k = tid; k <= half; k += 256) {
      double re = 0.0, im = 0.0;
      int idx = 0;
      for (int n = 0; n < n_fft; ++n) {
        const double2 w = tw[idx];
        const double x = xw[n];
        re = fma(x, w.x, re);
        im = fma(x, w.y, im);
        idx += k;
        idx -= idx >= n_fft ? n_fft : 0;
      }
      const float fr = (float)re, fi = (float)im;  // complex64, as librosa stores the transform
      const float p = fmaf(fr, fr, fi * fi);
      pmax = fmaxf(pmax, p);
      if (k < k_crop) out_db[t * (int64_t)k_crop + k] = power_to_db(p);
    }
    __syncthreads();
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) pmax = fmaxf(pmax, __shfl_xor(pmax, o, 64));
  if ((tid & 63) == 0) wmax[tid >> 6] = pmax;
  __syncthreads();
  if (tid == 0) atomicMax(&ws->pmax_bits, __float_as_uint(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]))));
}

// ---------------------------------------------------------------- generic level-1 histogram
__global__ __launch_bounds__(256) void hist1_kernel(const float* __restrict__ x, int64_t n, Workspace* __restrict__ ws) {
  __shared__ uint32_t h[NB1_PAD];
  for (int i = threadIdx.x; i < NB1_PAD; i += 256) h[i] = 0u;
  __syncthreads();
  const int64_t n4 = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    float4 v = reinterpret_cast<const float4*>(x)[i];
    atomicAdd(&h[bucket1(f2key(v.x))], 1u);
    atomicAdd(&h[bucket1(f2key(v.y))], 1u);
    atomicAdd(&h[bucket1(f2key(v.z))], 1u);
    atomicAdd(&h[bucket1(f2key(v.w))], 1u);
  }
  if (blockIdx.x == 0) {
    for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += 256) atomicAdd(&h[bucket1(f2key(x[i]))], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NB1; i += 256) {
    uint32_t c = h[i];
    if (c) atomicAdd(&ws->hist1[i], c);
  }
}

// ---------------------------------------------------------------- block-wide "which bin holds rank r"
// 1024 threads; hist has nb bins; returns (bin, exclusive prefix of that bin) to every thread.
// If zero_after, bins are reset to 0 after being read (ready for the next round).
__device__ void find_bin(uint32_t* hist, int nb, uint64_t rank, bool zero_after, int& bin_out, uint64_t& prefix_out) {
  __shared__ uint64_t wave_tot[16];
  __shared__ int s_bin;
  __shared__ uint64_t s_prefix;
  const int tid = threadIdx.x;
  const int per = (nb + 1023) / 1024;
  const int b0 = tid * per;
  uint64_t sum = 0;
  for (int i = 0; i < per; ++i) {
    int b = b0 + i;
    if (b < nb) sum += hist[b];
  }
  // inclusive scan of `sum` over the block
  uint64_t incl = sum;
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    uint64_t o = __shfl_up(incl, off, 64);
    if (lane >= off) incl += o;
  }
  if (lane == 63) wave_tot[wave] = incl;
  if (tid == 0) { s_bin = nb - 1; s_prefix = 0; }
  __syncthreads();
  uint64_t base = 0;
  for (int w = 0; w < wave; ++w) base += wave_tot[w];
  const uint64_t excl = base + incl - sum;
  if (rank >= excl && rank < excl + sum) {
    uint64_t run = excl;
    for (int i = 0; i < per; ++i) {
      int b = b0 + i;
      if (b >= nb) break;
      uint64_t c = hist[b];
      if (rank < run + c) { s_bin = b; s_prefix = run; break; }
      run += c;
    }
  }
  __syncthreads();
  bin_out = s_bin;
  prefix_out = s_prefix;
  if (zero_after) {
    for (int i = 0; i < per; ++i) {
      int b = b0 + i;
      if (b < nb) hist[b] = 0u;
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(1024) void select_scan1_kernel(Workspace* __restrict__ ws, int64_t rank_lo, int64_t rank_hi) {
  for (int q = 0; q < 2; ++q) {
    int bin; uint64_t prefix;
    const uint64_t rank = (uint64_t)(q == 0 ? rank_lo : rank_hi);
    find_bin(ws->hist1, NB1, rank, false, bin, prefix);
    if (threadIdx.x == 0) {
      uint32_t klo; uint64_t width;
      bucket1_range(bin, klo, width);
      ws->sel[q].klo = klo;
      ws->sel[q].width = width;
      ws->sel[q].rank = rank - prefix;
      ws->sel[q].done = (width == 1) ? 1u : 0u;
    }
  }
}

// One element against both pending intervals.  All 64 lanes of the wave call this together.
__device__ __forceinline__ void hist2_add(uint32_t key, bool live, const SelState& s0, const SelState& s1, int sh0, int sh1, int lane,
                                          Workspace* __restrict__ ws) {
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const SelState& s = q ? s1 : s0;
    if (s.done) continue;  // wave-uniform
    const uint32_t d = key - s.klo;
    const bool in = live && key >= s.klo && (uint64_t)d < s.width;
    const uint32_t bin = d >> (q ? sh1 : sh0);
    const unsigned long long mask = __ballot(in);
    if (mask == 0ull) continue;
    const int first = __ffsll((long long)mask) - 1;
    const uint32_t b0 = __shfl(bin, first, 64);
    const bool uniform = __all(!in || bin == b0);
    if (uniform) {  // ties / constant input: one atomic per wave instead of 64 on one address
      if (lane == first) {
        const uint32_t c = (uint32_t)__popcll(mask);
        atomicAdd(&ws->hist2[q][b0], c);
        atomicAdd(&ws->hist2c[q][b0 >> 8], c);
      }
    } else if (in) {
      atomicAdd(&ws->hist2[q][bin], 1u);
      atomicAdd(&ws->hist2c[q][bin >> 8], 1u);
    }
  }
}

__global__ __launch_bounds__(256) void hist2_kernel(const float* __restrict__ x, int64_t n, Workspace* __restrict__ ws) {
  const SelState s0 = ws->sel[0], s1 = ws->sel[1];
  if (s0.done && s1.done) return;
  const int sh0 = shift_for(s0.width), sh1 = shift_for(s1.width);
  const int lane = threadIdx.x & 63;
  const int64_t stride = (int64_t)gridDim.x * 256;
  const int64_t n4 = n >> 2;
  const int64_t n4_round = ((n4 + stride - 1) / stride) * stride;  // whole waves stay in the loop for the ballots
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4_round; i += stride) {
    const bool live = i < n4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) v = reinterpret_cast<const float4*>(x)[i];
    hist2_add(f2key(v.x), live, s0, s1, sh0, sh1, lane, ws);
    hist2_add(f2key(v.y), live, s0, s1, sh0, sh1, lane, ws);
    hist2_add(f2key(v.z), live, s0, s1, sh0, sh1, lane, ws);
    hist2_add(f2key(v.w), live, s0, s1, sh0, sh1, lane, ws);
  }
  if (blockIdx.x == 0 && threadIdx.x < 64) {  // tail (< 4 elements), one full wave
    const int64_t i = (n4 << 2) + threadIdx.x;
    const bool live = i < n;
    hist2_add(live ? f2key(x[i]) : 0u, live, s0, s1, sh0, sh1, lane, ws);
  }
}

__global__ __launch_bounds__(1024) void select_scan2_kernel(Workspace* __restrict__ ws) {
  for (int q = 0; q < 2; ++q) {
    SelState s = ws->sel[q];
    if (s.done) continue;  // uniform across the block
    const int sh = shift_for(s.width);
    int cbin, bin; uint64_t cprefix, prefix;
    find_bin(ws->hist2c[q], NB2C, s.rank, false, cbin, cprefix);              // which group of 256 fine bins
    find_bin(ws->hist2[q] + cbin * 256, 256, s.rank - cprefix, false, bin, prefix);  // which bin inside it
    bin += cbin * 256;
    prefix += cprefix;
    if (threadIdx.x == 0) {
      const uint64_t off = (uint64_t)bin << sh;
      uint64_t w = 1ull << sh;
      if (off + w > s.width) w = s.width - off;
      ws->sel[q].klo = s.klo + (uint32_t)off;
      ws->sel[q].width = w;
      ws->sel[q].rank = s.rank - prefix;
      ws->sel[q].done = (w == 1) ? 1u : 0u;
    }
    __syncthreads();
  }
}

__global__ void finalize_kernel(Workspace* __restrict__ ws, int use_ref, float top_db) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const float lo = key2f(ws->sel[0].klo), hi = key2f(ws->sel[1].klo);
  ws->sel_raw[0] = lo;
  ws->sel_raw[1] = hi;
  ws->use_ref = (uint32_t)use_ref;
  if (use_ref) {
    const float ref_db = power_to_db(__uint_as_float(ws->pmax_bits));
    ws->ref_db = ref_db;
    ws->floor_db = -top_db;
    ws->p_lo = fmaxf(lo - ref_db, -top_db);
    ws->p_hi = fmaxf(hi - ref_db, -top_db);
  } else {
    ws->ref_db = 0.0f;
    ws->floor_db = -INFINITY;
    ws->p_lo = lo;
    ws->p_hi = hi;
  }
}

__device__ __forceinline__ float norm1(float x, float ref_db, float floor_db, float lo, float hi, float range) {
  float v = fmaxf(x - ref_db, floor_db);
  v = fminf(fmaxf(v, lo), hi);
  return (v - lo) / range;
}

__global__ __launch_bounds__(256) void clip_normalize_kernel(float* __restrict__ x, int64_t n, const Workspace* __restrict__ ws) {
  const float ref_db = ws->ref_db, floor_db = ws->floor_db, lo = ws->p_lo, hi = ws->p_hi;
  const float range = hi - lo;
  const int64_t n4 = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    float4 v = reinterpret_cast<float4*>(x)[i];
    v.x = norm1(v.x, ref_db, floor_db, lo, hi, range);
    v.y = norm1(v.y, ref_db, floor_db, lo, hi, range);
    v.z = norm1(v.z, ref_db, floor_db, lo, hi, range);
    v.w = norm1(v.w, ref_db, floor_db, lo, hi, range);
    reinterpret_cast<float4*>(x)[i] = v;
  }
  if (blockIdx.x == 0)
    for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += 256) x[i] = norm1(x[i], ref_db, floor_db, lo, hi, range);
}

__global__ __launch_bounds__(256) void db_reference_kernel(float* __restrict__ x, int64_t n, const Workspace* __restrict__ ws) {
  const float ref_db = ws->ref_db, floor_db = ws->floor_db;  // set by finalize_kernel(use_ref = 1)
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) x[i] = fmaxf(x[i] - ref_db, floor_db);
}

// [F,T] -> [T,K]: 32x32 tiles through LDS
__global__ __launch_bounds__(256) void crop_transpose_kernel(const float* __restrict__ in, int64_t n_freq, int64_t n_frames, int f_lo, int K,
                                                              float* __restrict__ out) {
  __shared__ float tile[32][33];
  const int64_t t0 = (int64_t)blockIdx.x * 32;
  const int k0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int k = k0 + r;
    const int64_t t = t0 + tx;
    tile[r][tx] = (k < K && t < n_frames) ? in[(int64_t)(f_lo + k) * n_frames + t] : 0.0f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int64_t t = t0 + r;
    const int k = k0 + tx;
    if (k < K && t < n_frames) out[t * K + k] = tile[tx][r];
  }
}

std::once_flag g_tables_once;
int g_tables_err = 0;

void init_tables() {
  static float window[NFFT];
  static float2 tw256[256];
  static float2 tw512[260];
  const double PI = 3.14159265358979323846;
  for (int n = 0; n < NFFT; ++n) window[n] = (float)(0.5 - 0.5 * std::cos(2.0 * PI * n / NFFT));  // periodic Hann
  for (int k2 = 0; k2 < 16; ++k2)
    for (int n1 = 0; n1 < 16; ++n1) {
      const double a = -2.0 * PI * (double)(n1 * k2) / 256.0;
      tw256[k2 * 16 + n1] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
  for (int k = 0; k < 260; ++k) {
    const double a = 2.0 * PI * (double)k / 512.0;
    tw512[k] = make_float2((float)std::cos(a), (float)std::sin(a));
  }
  hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_window), window, sizeof(window));
  if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(g_tw256), tw256, sizeof(tw256));
  if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(g_tw512), tw512, sizeof(tw512));
  g_tables_err = (int)e;
}

inline int grid_for(int64_t items_per_block_unit, int64_t n_units) {
  (void)items_per_block_unit;
  int64_t g = n_units;
  if (g > 256 * 8) g = 256 * 8;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

extern "C" {

size_t orcai_frontend_workspace_bytes(void) {
  // Setup call (every caller needs it before it can allocate the workspace): the window / twiddle tables are uploaded here, with a
  // synchronous hipMemcpyToSymbol, so that no later call -- orcai_frontend_reset, orcai_stft_db, ... -- does anything but enqueue
  // work on the caller's stream (hipGraph-capturable from the first call on).  An upload error is reported by orcai_stft_db.
  std::call_once(g_tables_once, init_tables);
  return (sizeof(Workspace) + 255) & ~(size_t)255;
}

int orcai_frontend_reset(void* workspace, void* stream) {
  if (!workspace) return ORCAI_E_BADARG;
  return (int)orcai_zero::zero_async(workspace, sizeof(Workspace), (hipStream_t)stream);
}

int orcai_stft_db(const float* pcm, int64_t n_samples, int n_fft, int hop, int64_t n_frames, int k_crop, float* out_db, void* workspace,
                  void* stream) {
  if (!pcm || !out_db || !workspace || n_samples <= 0 || hop <= 0 || k_crop < 1 || n_fft < 2 || k_crop > 1 + n_fft / 2) return ORCAI_E_BADARG;
  if (n_frames != 1 + (n_samples - (n_fft & 1)) / hop) return ORCAI_E_BADARG;  // librosa, center = True: 1 + (n + 2 (n_fft / 2) - n_fft) / hop
  if (((uintptr_t)out_db & 15) || ((uintptr_t)pcm & 15)) return ORCAI_E_BADARG;
  if (n_fft != NFFT) {  // any other size up to 4096: the plain radix-2 kernel (powers of two from 32) or the direct transform, + a separate level-1 histogram pass
    if (n_fft > NFFT_ANY_MAX) return ORCAI_E_UNSUPPORTED;
    if (n_fft < 32 || (n_fft & (n_fft - 1))) {  // not a power of two (or a tiny one): the direct transform
      const size_t lds = sizeof(double) * ((size_t)n_fft + (n_fft & 1) + 2 * (size_t)n_fft);
      if (lds > 48 * 1024) {
        static bool opted_dev[64] = {};
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return (int)e;
        if (dev < 0 || dev >= 64) return ORCAI_E_UNSUPPORTED;
        if (!opted_dev[dev]) {
          e = hipFuncSetAttribute((const void*)stft_dft_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * NFFT_ANY_MAX * (int)sizeof(double));
          if (e != hipSuccess) return (int)e;
          opted_dev[dev] = true;
        }
      }
      const int64_t blocks = n_frames < 256 * 4 ? n_frames : 256 * 4;
      hipLaunchKernelGGL(stft_dft_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, pcm, n_samples, n_fft, hop, n_frames, k_crop, out_db, (Workspace*)workspace);
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) return (int)e;
      return orcai_hist_level1(out_db, n_frames * (int64_t)k_crop, workspace, stream);
    }
    int log2n = 0;
    while ((1 << log2n) < n_fft) ++log2n;
    const int64_t blocks = n_frames < 256 * 16 ? n_frames : 256 * 16;
    hipLaunchKernelGGL(stft_any_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, pcm, n_samples, n_fft, log2n, hop, n_frames, k_crop, out_db,
                       (Workspace*)workspace);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    return orcai_hist_level1(out_db, n_frames * (int64_t)k_crop, workspace, stream);
  }
  std::call_once(g_tables_once, init_tables);  // no-op: orcai_frontend_workspace_bytes() already ran it (kept for callers that size the workspace themselves)
  if (g_tables_err) return g_tables_err;
  const int64_t n_groups = (n_frames + 15) / 16;
  Workspace* ws = (Workspace*)workspace;
  hipStream_t st = (hipStream_t)stream;
  auto grid_for_groups = [](int64_t n) { return dim3((unsigned)(n < g_stft_blocks ? n : g_stft_blocks)); };
  if ((hop & 1) != 0) {
    hipLaunchKernelGGL((stft_db_kernel<false, false, 0>), grid_for_groups(n_groups), dim3(256), 0, st, pcm, n_samples, hop, n_frames, k_crop, out_db, ws,
                       (int64_t)0, n_groups);
    return (int)hipGetLastError();
  }
  // Interior groups [g_lo, g_hi): all 16 frames exist and the samples of every wave lie inside the recording, 16-byte aligned
  // (hop 256): t0w >= 1 for the first wave of the group, (16g + 12) * 256 + 1024 <= n_samples and 16g + 15 < n_frames for the last.
  int64_t g_lo = n_groups, g_hi = n_groups;
  if (hop == 256) {
    g_lo = 1;
    const int64_t by_samples = n_samples >= 4096 ? ((n_samples - 1024) / 256 - 12) / 16 + 1 : 0;  // groups g with (16g+12)*256+1024 <= n_samples
    const int64_t by_frames = n_frames >= 16 ? (n_frames - 16) / 16 + 1 : 0;                       // groups g with 16g + 15 < n_frames
    g_hi = by_samples < by_frames ? by_samples : by_frames;
    if (g_hi > n_groups) g_hi = n_groups;
    if (g_hi < g_lo) g_lo = g_hi = n_groups;  // no interior: everything goes through the mixed kernel
  }
  if (g_hi > g_lo) {
    if (k_crop == 171)
      hipLaunchKernelGGL((stft_db_kernel<true, true, 171>), grid_for_groups(g_hi - g_lo), dim3(256), 0, st, pcm, n_samples, hop, n_frames, k_crop, out_db, ws, g_lo, g_hi);
    else
      hipLaunchKernelGGL((stft_db_kernel<true, true, 0>), grid_for_groups(g_hi - g_lo), dim3(256), 0, st, pcm, n_samples, hop, n_frames, k_crop, out_db, ws, g_lo, g_hi);
    if (g_lo > 0)
      hipLaunchKernelGGL((stft_db_kernel<true, false, 0>), grid_for_groups(g_lo), dim3(256), 0, st, pcm, n_samples, hop, n_frames, k_crop, out_db, ws, (int64_t)0, g_lo);
    if (g_hi < n_groups)
      hipLaunchKernelGGL((stft_db_kernel<true, false, 0>), grid_for_groups(n_groups - g_hi), dim3(256), 0, st, pcm, n_samples, hop, n_frames, k_crop, out_db, ws, g_hi,
                         n_groups);
  } else {
    hipLaunchKernelGGL((stft_db_kernel<true, false, 0>), grid_for_groups(n_groups), dim3(256), 0, st, pcm, n_samples, hop, n_frames, k_crop, out_db, ws, (int64_t)0,
                       n_groups);
  }
  return (int)hipGetLastError();
}

int orcai_stft_blocks(int blocks) {  // experiments: workgroups of the persistent STFT launch (default 1536 = twice the three resident per compute unit); < 0 queries
  const int prev = g_stft_blocks;
  if (blocks > 0) g_stft_blocks = blocks;
  return prev;
}

int orcai_stft_occupancy(void) {  // workgroups of stft_db_kernel<true, true, 171> the runtime says fit one compute unit (LDS: 52 KiB each)
  int n = -1;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)stft_db_kernel<true, true, 171>, 256, 0) != hipSuccess) return -1;
  return n;
}

int orcai_hist_level1(const float* x, int64_t n, void* workspace, void* stream) {
  if (!x || !workspace || n <= 0 || ((uintptr_t)x & 15)) return ORCAI_E_BADARG;
  int grid = grid_for(0, (n / 4 + 255) / 256);
  hipLaunchKernelGGL(hist1_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, n, (Workspace*)workspace);
  return (int)hipGetLastError();
}

int orcai_quantile_select(const float* x, int64_t n, int64_t rank_lo, int64_t rank_hi, void* workspace, void* stream) {
  if (!x || !workspace || n <= 0 || rank_lo < 0 || rank_hi < 0 || rank_lo >= n || rank_hi >= n) return ORCAI_E_BADARG;
  Workspace* ws = (Workspace*)workspace;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(select_scan1_kernel, dim3(1), dim3(1024), 0, s, ws, rank_lo, rank_hi);
  if ((uintptr_t)x & 15) return ORCAI_E_BADARG;
  int grid = grid_for(0, (n / 4 + 255) / 256);
  for (int round = 0; round < 2; ++round) {  // a level-1 bucket is at most 2^31 keys wide: two 16-bit rounds
    hipError_t e = orcai_zero::zero_async(ws->hist2, sizeof(ws->hist2) + sizeof(ws->hist2c), s);  // hist2c follows hist2
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(hist2_kernel, dim3(grid), dim3(256), 0, s, x, n, ws);
    hipLaunchKernelGGL(select_scan2_kernel, dim3(1), dim3(1024), 0, s, ws);
  }
  return (int)hipGetLastError();
}

int orcai_frontend_finalize(int use_ref, float top_db, void* workspace, void* stream) {
  if (!workspace) return ORCAI_E_BADARG;
  hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (Workspace*)workspace, use_ref, top_db);
  return (int)hipGetLastError();
}

int orcai_clip_normalize(float* x, int64_t n, const void* workspace, void* stream) {
  if (!x || !workspace || n <= 0 || ((uintptr_t)x & 15)) return ORCAI_E_BADARG;
  int grid = grid_for(0, (n / 4 + 255) / 256);
  hipLaunchKernelGGL(clip_normalize_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, n, (const Workspace*)workspace);
  return (int)hipGetLastError();
}

int orcai_db_reference(float* x, int64_t n, const void* workspace, void* stream) {
  if (!x || !workspace || n <= 0) return ORCAI_E_BADARG;
  int grid = grid_for(0, (n + 255) / 256);
  hipLaunchKernelGGL(db_reference_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, n, (const Workspace*)workspace);
  return (int)hipGetLastError();
}

int orcai_crop_transpose(const float* in_ft, int64_t n_freq, int64_t n_frames, int f_lo, int f_hi, float* out_tf, void* stream) {
  if (!in_ft || !out_tf || n_freq <= 0 || n_frames <= 0 || f_lo < 0 || f_hi <= f_lo || f_hi > n_freq) return ORCAI_E_BADARG;
  const int K = f_hi - f_lo;
  dim3 grid((unsigned)((n_frames + 31) / 32), (unsigned)((K + 31) / 32));
  hipLaunchKernelGGL(crop_transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, in_ft, n_freq, n_frames, f_lo, K, out_tf);
  return (int)hipGetLastError();
}

int orcai_frontend_stats_host(const void* workspace, float stats_host[6], void* stream) {
  if (!workspace || !stats_host) return ORCAI_E_BADARG;
  Workspace* h = new Workspace;
  hipError_t e = hipMemcpyAsync(h, workspace, sizeof(Workspace), hipMemcpyDeviceToHost, (hipStream_t)stream);
  if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
  if (e == hipSuccess) {
    float pm;
    std::memcpy(&pm, &h->pmax_bits, 4);
    stats_host[0] = pm;
    stats_host[1] = h->ref_db;
    stats_host[2] = h->p_lo;
    stats_host[3] = h->p_hi;
    stats_host[4] = h->sel_raw[0];
    stats_host[5] = h->sel_raw[1];
  }
  delete h;
  return (int)e;
}

int orcai_make_spectrogram(const float* pcm, int64_t n_samples, int n_fft, int hop, int64_t n_frames, int k_crop, int64_t rank_lo,
                           int64_t rank_hi, float top_db, float* out, void* workspace, void* stream) {
  int e = orcai_frontend_reset(workspace, stream);
  if (e) return e;
  e = orcai_stft_db(pcm, n_samples, n_fft, hop, n_frames, k_crop, out, workspace, stream);
  if (e) return e;
  const int64_t n = n_frames * (int64_t)k_crop;
  e = orcai_quantile_select(out, n, rank_lo, rank_hi, workspace, stream);
  if (e) return e;
  e = orcai_frontend_finalize(1, top_db, workspace, stream);
  if (e) return e;
  return orcai_clip_normalize(out, n, workspace, stream);
}

}  // extern "C"
