// train_head.hip -- training-mode forward extras and backward of the recurrent head of ResNetLSTM for gfx950:
// strided f32-MFMA GEMM (weight / input gradients), row-tensor BatchNorm / dropout / ReLU kernels, masked BCE,
// LSTM backward through time, Adam.   Reference: architectures.py:210-239 (layers), :244-286 (loss, metric),
// train.py:155-161 (Adam(lr), MaskedBinaryCrossentropy, MaskedBinaryAccuracy).
//
// "Row tensors" are [M][C] row-major with M = snippets * time steps.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "orcai_hip.h"
#include "zero_fill.h"

extern int g_orcai_lstm_split;  // model_fwd.hip: 1 = LSTM recurrences of the f32 path on split-f16 MFMA (orcai_lstm_split)

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// =========================================================================================
// C[M][N] (=|+=) alpha * sum_k A(m,k) B(k,n) + beta_w * Wreg[m][n],  A(m,k) = A[m*sam + k*sak], B(k,n) = B[k*sbk + n*sbn]
// (covers A^T B for weight gradients and A B^T for input gradients; Wreg adds the L2 term 2*lambda*W).
// 64 x 64 block tile, BK = 16, 4 waves as 2 x 2, wave tile 32 x 32.
// =========================================================================================
constexpr int SP = 64 + 16;

__global__ __launch_bounds__(256) void gemm_strided_kernel(const float* __restrict__ A, int64_t sam, int64_t sak, const float* __restrict__ B, int64_t sbk,
                                                            int64_t sbn, float* __restrict__ C, int M, int N, int K, float alpha, int accumulate,
                                                            const float* __restrict__ Wreg, float beta_w) {
  __shared__ float As[16][SP];
  __shared__ float Bs[16][SP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lk = lane >> 4, lj = lane & 15;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < K; k0 += 16) {
    {
      const int row = tid >> 2, kq = (tid & 3) * 4;  // 64 rows x 16 k
      const int m = m0 + row;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int k = k0 + kq + q;
        As[kq + q][row] = (m < M && k < K) ? A[(int64_t)m * sam + (int64_t)k * sak] : 0.0f;
      }
    }
    {
      const int k = tid >> 4, nn = (tid & 15) * 4;  // 16 k x 64 n
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + nn + q;
        Bs[k][nn + q] = (k0 + k < K && n < N) ? B[(int64_t)(k0 + k) * sbk + (int64_t)n * sbn] : 0.0f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      float a[2], bq[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = As[kk * 4 + lk][wm * 32 + i * 16 + lj];
#pragma unroll
      for (int j = 0; j < 2; ++j) bq[j] = Bs[kk * 4 + lk][wn * 32 + j * 16 + lj];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma16(a[i], bq[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + wm * 32 + i * 16 + lk * 4 + r;
      if (m >= M) continue;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 32 + j * 16 + lj;
        if (n >= N) continue;
        float v = alpha * acc[i][j][r];
        if (Wreg) v = fmaf(beta_w, Wreg[(int64_t)m * N + n], v);
        float* c = C + (int64_t)m * N + n;
        *c = accumulate ? *c + v : v;
      }
    }
}

// =========================================================================================
// The same product for the two stride patterns training uses, with 16-byte global loads, a register-prefetched
// 64 x 64 x 32 tile pipeline and split-K (partial products added with global_atomic_add_f32):
//   XM = 0: X(r, k) = X[k*ld + r]  (r contiguous; weight gradients A^T B)      tile row k, one float4 along r per thread
//   XM = 1: X(r, k) = X[r*ld + k]  (k contiguous; input gradients  A B^T)      one float4 along k per thread, r fastest over lanes
// =========================================================================================
constexpr int GBK = 32;

template <int XM>
__device__ __forceinline__ void gemm_tile_load(const float* __restrict__ X, int64_t ld, int r0, int rmax, int k0, int kend, bool vec, int tid, float4 (&reg)[2]) {
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int f = tid + 256 * g;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (XM == 0) {
      const int k = k0 + (f >> 4), r = r0 + (f & 15) * 4;
      const float* base = X + (int64_t)k * ld + r;
      if (k < kend) {
        if (vec && r + 3 < rmax) {
          const float4 t = *reinterpret_cast<const float4*>(base);
          v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (r + e < rmax) ? base[e] : 0.0f;
        }
      }
    } else {
      const int r = r0 + (f & 63), k = k0 + (f >> 6) * 4;
      const float* base = X + (int64_t)r * ld + k;
      if (r < rmax) {
        if (vec && k + 3 < kend) {
          const float4 t = *reinterpret_cast<const float4*>(base);
          v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (k + e < kend) ? base[e] : 0.0f;
        }
      }
    }
    reg[g] = make_float4(v[0], v[1], v[2], v[3]);
  }
}

template <int XM>
__device__ __forceinline__ void gemm_tile_store(float (*S)[SP], int tid, const float4 (&reg)[2]) {
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int f = tid + 256 * g;
    if constexpr (XM == 0) {
      *reinterpret_cast<float4*>(&S[f >> 4][(f & 15) * 4]) = reg[g];
    } else {
      const int r = f & 63, k = (f >> 6) * 4;
      S[k][r] = reg[g].x; S[k + 1][r] = reg[g].y; S[k + 2][r] = reg[g].z; S[k + 3][r] = reg[g].w;
    }
  }
}

template <int AM, int BM>
__global__ __launch_bounds__(256) void gemm_tiled_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb, float* __restrict__ C,
                                                          int M, int N, int K, float alpha, int accumulate, const float* __restrict__ Wreg, float beta_w,
                                                          int k_per_split, int vec_a, int vec_b) {
  __shared__ __attribute__((aligned(16))) float As[GBK][SP];
  __shared__ __attribute__((aligned(16))) float Bs[GBK][SP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lk = lane >> 4, lj = lane & 15;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int kbeg = blockIdx.z * k_per_split;
  const int kend = (kbeg + k_per_split < K) ? kbeg + k_per_split : K;
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float4 ra[2], rb[2];
  gemm_tile_load<AM>(A, lda, m0, M, kbeg, kend, vec_a != 0, tid, ra);
  gemm_tile_load<BM>(B, ldb, n0, N, kbeg, kend, vec_b != 0, tid, rb);
  for (int k0 = kbeg; k0 < kend; k0 += GBK) {
    __syncthreads();  // previous tile fully consumed
    gemm_tile_store<AM>(As, tid, ra);
    gemm_tile_store<BM>(Bs, tid, rb);
    __syncthreads();
    if (k0 + GBK < kend) {  // next tile in flight during this tile's MFMAs
      gemm_tile_load<AM>(A, lda, m0, M, k0 + GBK, kend, vec_a != 0, tid, ra);
      gemm_tile_load<BM>(B, ldb, n0, N, k0 + GBK, kend, vec_b != 0, tid, rb);
    }
#pragma unroll
    for (int kk = 0; kk < GBK / 4; ++kk) {
      float a[2], bq[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = As[kk * 4 + lk][wm * 32 + i * 16 + lj];
#pragma unroll
      for (int j = 0; j < 2; ++j) bq[j] = Bs[kk * 4 + lk][wn * 32 + j * 16 + lj];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma16(a[i], bq[j], acc[i][j]);
    }
  }
  const bool split = gridDim.z > 1;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + wm * 32 + i * 16 + lk * 4 + r;
      if (m >= M) continue;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 32 + j * 16 + lj;
        if (n >= N) continue;
        float v = alpha * acc[i][j][r];
        if (Wreg && blockIdx.z == 0) v = fmaf(beta_w, Wreg[(int64_t)m * N + n], v);
        float* c = C + (int64_t)m * N + n;
        if (split) atomicAdd(c, v);  // C was zeroed by the launcher unless accumulating
        else *c = accumulate ? *c + v : v;
      }
    }
}

// column sums of a row tensor: out[c] (=|+=) sum_m x[m][c]      (bias gradients)
// One workgroup per 16 columns: 64 row groups x 16 consecutive columns (a wave reads four 64-byte row segments per step), eight loads in flight per
// thread.  (The first version -- one workgroup per column, 4-byte reads a row apart, one load in flight -- took 24 us for the 12 MB of an LSTM bias gradient.)
__global__ __launch_bounds__(1024) void colsum_kernel(const float* __restrict__ x, int M, int C, float* __restrict__ out, int accumulate) {
  __shared__ float part[64][17];
  const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4, c = blockIdx.x * 16 + cl;
  float s = 0.0f;
  if (c < C) {
    const float* xc = x + c;
    int m = rg;
    for (; m + 7 * 64 < M; m += 8 * 64) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = xc[(int64_t)(m + 64 * k) * C];
#pragma unroll
      for (int k = 0; k < 8; ++k) s += v[k];
    }
    for (; m < M; m += 64) s += xc[(int64_t)m * C];
  }
  part[rg][cl] = s;
  __syncthreads();
  if (threadIdx.x < 16 && c < C) {
    float t = 0.0f;
#pragma unroll
    for (int r = 0; r < 64; ++r) t += part[r][threadIdx.x];
    out[c] = accumulate ? out[c] + t : t;
  }
}

// =========================================================================================
// BatchNormalization (training) on a row tensor whose column index is (position * C + channel), i.e. the channel is
// column % C: Dense-128 output (C = 128 = columns) and the final separable conv in Keras Reshape layout (C = 36).
// stats[c] = {mean, biased var}; one workgroup per channel, float64 accumulation.
// =========================================================================================
// (1 024 threads over the flattened (row, position) index, the row recovered by a multiply-high with the launcher's reciprocal of per_row, eight loads in
// flight per thread: the first version divided a 64-bit element index per load, one load in flight, and took 59 us for the 4.6 MB of the final conv's output)
__device__ __forceinline__ int64_t rows_index(uint32_t i, int per_row, uint32_t magic, int cols, int C) {
  const uint32_t m = per_row == 1 ? i : __umulhi(i, magic);
  const uint32_t p = i - m * (uint32_t)per_row;
  return (int64_t)m * cols + (int64_t)p * C;
}

__global__ __launch_bounds__(1024) void bn_rows_stats_kernel(const float* __restrict__ x, int M, int cols, int C, uint32_t magic, float* __restrict__ mean,
                                                              float* __restrict__ var) {
  __shared__ double s1[1024], s2[1024];
  const int c = blockIdx.x;
  const int per_row = cols / C;
  double a = 0.0, b = 0.0;
  const uint32_t total = (uint32_t)M * (uint32_t)per_row;
  const float* xc = x + c;
  uint32_t i = threadIdx.x;
  for (; i + 7 * 1024 < total; i += 8 * 1024) {
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = xc[rows_index(i + 1024 * k, per_row, magic, cols, C)];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      a += (double)v[k];
      b += (double)v[k] * (double)v[k];
    }
  }
  for (; i < total; i += 1024) {
    const double v = (double)xc[rows_index(i, per_row, magic, cols, C)];
    a += v;
    b += v * v;
  }
  s1[threadIdx.x] = a;
  s2[threadIdx.x] = b;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (threadIdx.x < o) { s1[threadIdx.x] += s1[threadIdx.x + o]; s2[threadIdx.x] += s2[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double mu = s1[0] / (double)total;
    mean[c] = (float)mu;
    double vv = s2[0] / (double)total - mu * mu;
    var[c] = (float)(vv < 0.0 ? 0.0 : vv);
  }
}

// y = [relu]( (x - mean) * gamma * rsqrt(var + eps) + beta )
// (grid: column blocks x rows -- the channel is a 32-bit column % C)
__global__ __launch_bounds__(256) void bn_rows_apply_kernel(const float* __restrict__ x, int M, int cols, int C, const float* __restrict__ mean,
                                                             const float* __restrict__ var, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float eps, int relu, float* __restrict__ y) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= cols) return;
  const int c = col % C;
  const float inv = gamma[c] * rsqrtf(var[c] + eps), sh = beta[c] - mean[c] * inv;
  for (int m = blockIdx.y; m < M; m += gridDim.y) {
    const int64_t i = (int64_t)m * cols + col;
    float v = fmaf(x[i], inv, sh);
    if (relu) v = fmaxf(v, 0.0f);
    y[i] = v;
  }
}

// BN backward, reduction part: with dy_eff = relu ? dy * (y > 0) : dy  (y = BN output),
//   sums[c] = {sum dy_eff, sum dy_eff * xhat};  also dgamma = sums[1], dbeta = sums[0].
__global__ __launch_bounds__(1024) void bn_rows_bwd_stats_kernel(const float* __restrict__ dy, const float* __restrict__ x, int M, int cols, int C, uint32_t magic,
                                                                  const float* __restrict__ mean, const float* __restrict__ var,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta, float eps, int relu,
                                                                  float* __restrict__ dbeta, float* __restrict__ dgamma) {
  __shared__ double s1[1024], s2[1024];
  const int c = blockIdx.x;
  const int per_row = cols / C;
  const float inv = rsqrtf(var[c] + eps), mu = mean[c], g = gamma[c], bt = beta[c];
  double a = 0.0, b = 0.0;
  const uint32_t total = (uint32_t)M * (uint32_t)per_row;
  auto take = [&](float xv, float d) {
    const float xh = (xv - mu) * inv;
    if (relu && !(fmaf(xh, g, bt) > 0.0f)) d = 0.0f;
    a += (double)d;
    b += (double)d * (double)xh;
  };
  uint32_t i = threadIdx.x;
  for (; i + 3 * 1024 < total; i += 4 * 1024) {
    float xv[4], dv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t idx = rows_index(i + 1024 * k, per_row, magic, cols, C) + c;
      xv[k] = x[idx];
      dv[k] = dy[idx];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) take(xv[k], dv[k]);
  }
  for (; i < total; i += 1024) {
    const int64_t idx = rows_index(i, per_row, magic, cols, C) + c;
    take(x[idx], dy[idx]);
  }
  s1[threadIdx.x] = a;
  s2[threadIdx.x] = b;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (threadIdx.x < o) { s1[threadIdx.x] += s1[threadIdx.x + o]; s2[threadIdx.x] += s2[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { dbeta[c] = (float)s1[0]; dgamma[c] = (float)s2[0]; }
}

// dx = gamma * inv * (dy_eff - dbeta/N - xhat * dgamma/N)
__global__ __launch_bounds__(256) void bn_rows_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x, int M, int cols, int C, float count,
                                                                 const float* __restrict__ mean, const float* __restrict__ var,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta, float eps, int relu,
                                                                 const float* __restrict__ dbeta, const float* __restrict__ dgamma,
                                                                 float* __restrict__ dx) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= cols) return;
  const int c = col % C;
  const float inv = rsqrtf(var[c] + eps), mu = mean[c], g = gamma[c], bt = beta[c], db = dbeta[c] / count, dg = dgamma[c] / count;
  for (int m = blockIdx.y; m < M; m += gridDim.y) {
    const int64_t i = (int64_t)m * cols + col;
    const float xh = (x[i] - mu) * inv;
    float d = dy[i];
    if (relu && !(fmaf(xh, g, bt) > 0.0f)) d = 0.0f;
    dx[i] = g * inv * (d - db - xh * dg);
  }
}

// ---------------------------------------------------------------- the same reductions with COALESCED reads (the kernels the launchers use up to 2 048 columns)
// One workgroup per channel reads 4 bytes out of every 4 C-byte group of the tensor: 64 cache lines per wave load, 39 us for 4.6 MB.  Here a workgroup owns a
// slab of rows and a thread a column (consecutive lanes read consecutive floats); the columns of a channel meet in LDS, the slabs in a library-internal
// device array (ROWS_SLABS x 2 x ROWS_MAXC doubles) that a second launch folds in slab order -- no atomics, the same bits every run.  That array makes the
// row-BatchNorm launchers one-stream-at-a-time per device, like orcai_lstm_bwd (include/orcai_hip.h).
constexpr int ROWS_SLABS = 64, ROWS_MAXC = 512, ROWS_MAXCOLS = 2048;
__device__ double g_rows_partials[ROWS_SLABS * 2 * ROWS_MAXC];

template <bool BWD>
__global__ __launch_bounds__(512) void bn_rows_partial_kernel(const float* __restrict__ dy, const float* __restrict__ x, int M, int cols, int C, int rows_per_slab,
                                                               const float* __restrict__ mean, const float* __restrict__ var, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float eps, int relu, double* __restrict__ part) {
  __shared__ double ca[ROWS_MAXCOLS], cb[ROWS_MAXCOLS];
  const int m0 = blockIdx.x * rows_per_slab, m1 = (m0 + rows_per_slab < M) ? m0 + rows_per_slab : M;
  for (int col = threadIdx.x; col < cols; col += 512) {
    const int c = col % C;
    float inv = 0.f, mu = 0.f, g = 0.f, bt = 0.f;
    if (BWD) {
      inv = rsqrtf(var[c] + eps);
      mu = mean[c];
      g = gamma[c];
      bt = beta[c];
    }
    double a = 0.0, b = 0.0;
    auto take = [&](float xv, float d) {
      if (BWD) {
        const float xh = (xv - mu) * inv;
        if (relu && !(fmaf(xh, g, bt) > 0.0f)) d = 0.0f;
        a += (double)d;
        b += (double)d * (double)xh;
      } else {
        a += (double)xv;
        b += (double)xv * (double)xv;
      }
    };
    int m = m0;
    for (; m + 7 < m1; m += 8) {
      float xv[8], dv[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        xv[k] = x[(int64_t)(m + k) * cols + col];
        dv[k] = BWD ? dy[(int64_t)(m + k) * cols + col] : 0.0f;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) take(xv[k], dv[k]);
    }
    for (; m < m1; ++m) take(x[(int64_t)m * cols + col], BWD ? dy[(int64_t)m * cols + col] : 0.0f);
    ca[col] = a;
    cb[col] = b;
  }
  __syncthreads();
  const int per_row = cols / C;
  for (int c = threadIdx.x; c < C; c += 512) {
    double a = 0.0, b = 0.0;
    for (int p = 0; p < per_row; ++p) {
      a += ca[p * C + c];
      b += cb[p * C + c];
    }
    part[(blockIdx.x * 2 + 0) * ROWS_MAXC + c] = a;
    part[(blockIdx.x * 2 + 1) * ROWS_MAXC + c] = b;
  }
}

// the slabs' partials folded in slab order, 16 loads in flight per thread (one load at a time is 64 dependent L2 round trips: 30 us); thread = channel.
// bwd == 0: mean / biased variance;  bwd != 0: out0 = d beta, out1 = d gamma
__global__ __launch_bounds__(512) void bn_rows_finish_kernel(const double* __restrict__ part, int C, int slabs, double total, int bwd, float* __restrict__ out0,
                                                              float* __restrict__ out1) {
  const int c = threadIdx.x;
  if (c >= C) return;
  double a = 0.0, b = 0.0;
  for (int s0 = 0; s0 < slabs; s0 += 16) {
    double va[16], vb[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int sl = s0 + k < slabs ? s0 + k : slabs - 1;
      va[k] = part[(sl * 2 + 0) * ROWS_MAXC + c];
      vb[k] = part[(sl * 2 + 1) * ROWS_MAXC + c];
    }
#pragma unroll
    for (int k = 0; k < 16; ++k)
      if (s0 + k < slabs) {
        a += va[k];
        b += vb[k];
      }
  }
  if (bwd) {
    out0[c] = (float)a;
    out1[c] = (float)b;
    return;
  }
  const double mu = a / total;
  out0[c] = (float)mu;
  const double vv = b / total - mu * mu;
  out1[c] = (float)(vv < 0.0 ? 0.0 : vv);
}

// y = x * mask * scale (Dropout forward and backward; mask holds 0/1)
__global__ __launch_bounds__(256) void mask_scale_kernel(const float* __restrict__ x, const float* __restrict__ mask, float scale, int64_t n,
                                                          float* __restrict__ y) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) y[i] = x[i] * mask[i] * scale;
}

// dx = dy * (y > 0)
__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, int64_t n, float* __restrict__ dx) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dx[i] = y[i] > 0.0f ? dy[i] : 0.0f;
}

// counter-based Bernoulli(keep) mask: splitmix64 of (seed, element index) -> 0/1
__global__ __launch_bounds__(256) void dropout_mask_kernel(float* __restrict__ mask, int64_t n, uint64_t seed, float keep) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (uint64_t)(i + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  const float u = (float)(z >> 40) * (1.0f / 16777216.0f);
  mask[i] = u < keep ? 1.0f : 0.0f;
}

// =========================================================================================
// MaskedBinaryCrossentropy + MaskedBinaryAccuracy (architectures.py:244-286).
// pass 1: acc[0] += sum of BCE over unmasked elements, acc[1] += count, acc[2] += correct ((p > 0.5) == y)
// pass 2: dz = dL/d(logit) = mask * (dL/dp) * p (1 - p) / count   (dL/dp = 0 where p is clipped)
// =========================================================================================
__global__ __launch_bounds__(256) void bce_reduce_kernel(const float* __restrict__ p, const float* __restrict__ y, int64_t n, float mask_value,
                                                          double* __restrict__ acc, const float* __restrict__ loss_weight) {
  __shared__ double s0[256], s1[256], s2[256];
  double l = 0.0, c = 0.0, k = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float t = y[i];
    if (t != mask_value) {
      const float q = fminf(fmaxf(p[i], 1e-7f), 1.0f - 1e-7f);
      l += -((double)t * (double)logf(q) + (1.0 - (double)t) * (double)logf(1.0f - q));
      c += 1.0;
      k += ((p[i] > 0.5f ? 1.0f : 0.0f) == t) ? 1.0 : 0.0;
    }
  }
  s0[threadIdx.x] = l; s1[threadIdx.x] = c; s2[threadIdx.x] = k;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) { s0[threadIdx.x] += s0[threadIdx.x + o]; s1[threadIdx.x] += s1[threadIdx.x + o]; s2[threadIdx.x] += s2[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { atomicAdd(&acc[0], loss_weight ? s0[0] * (double)*loss_weight : s0[0]); atomicAdd(&acc[1], s1[0]); atomicAdd(&acc[2], s2[0]); }
}

__global__ __launch_bounds__(256) void bce_grad_kernel(const float* __restrict__ p, const float* __restrict__ y, int64_t n, float mask_value,
                                                        const double* __restrict__ acc, float* __restrict__ dz, const float* __restrict__ loss_weight, float grad_scale) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float t = y[i], q = p[i];
  float g = 0.0f;
  if (t != mask_value && q > 1e-7f && q < 1.0f - 1e-7f) {
    const float dLdp = -t / q + (1.0f - t) / (1.0f - q);
    g = dLdp * q * (1.0f - q) / (float)acc[1];
    if (loss_weight) g *= *loss_weight;  // Keras class_weight on a scalar loss: loss * mean(sample weight)
  }
  dz[i] = g * grad_scale;  // grad_scale: static loss scale of the f16 path (1 otherwise)
}

// The same mask with its seed taken from DEVICE memory: seed = seed_add + counter[0] * golden ratio.  A captured hipGraph bakes kernel
// arguments in, so the per-step part of the seed (and Adam's step number below) must live in a buffer the graph reads at replay time.
__global__ __launch_bounds__(256) void dropout_mask_dev_kernel(float* __restrict__ mask, int64_t n, const uint64_t* __restrict__ counter, uint64_t seed_add,
                                                                float keep) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint64_t seed = seed_add + counter[0] * 0xD1B54A32D192ED03ull;
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (uint64_t)(i + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  const float u = (float)(z >> 40) * (1.0f / 16777216.0f);
  mask[i] = u < keep ? 1.0f : 0.0f;
}

// Adam with the step number and the learning rate read from device memory (step = counter[0] + 1): alpha is formed once per block.
__global__ __launch_bounds__(256) void adam_dev_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                        const float* __restrict__ lr, float b1, float b2, float eps, const uint64_t* __restrict__ counter, float gscale,
                                                        const int32_t* __restrict__ ok = nullptr) {
  __shared__ float alpha_s;
  if (ok && !ok[0]) return;  // a voided step (f16 overflow): weights, moments untouched
  if (threadIdx.x == 0) {
    const double t = (double)(counter[0] + 1);
    alpha_s = (float)((double)lr[0] * sqrt(1.0 - pow((double)b2, t)) / (1.0 - pow((double)b1, t)));
  }
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i] * gscale;
  const float mi = m[i] + (gi - m[i]) * (1.0f - b1);
  const float vi = v[i] + (gi * gi - v[i]) * (1.0f - b2);
  m[i] = mi;
  v[i] = vi;
  w[i] = w[i] - alpha_s * mi / (sqrtf(vi) + eps);
}

__global__ void counter_advance_kernel(uint64_t* counter, const int32_t* ok = nullptr) {
  if (threadIdx.x == 0 && blockIdx.x == 0 && !(ok && !ok[0])) counter[0] += 1;
}

// ok[0] &= every value of a[0, n) is finite (ok is set to 1 by the launcher's first kernel); skipped[0] += !ok by the finishing kernel
__global__ void flag_set_kernel(int32_t* ok) {
  if (threadIdx.x == 0 && blockIdx.x == 0) ok[0] = 1;
}
__global__ __launch_bounds__(256) void all_finite_kernel(const float* __restrict__ a, int64_t n, int32_t* __restrict__ ok) {
  bool bad = false;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const uint32_t bits = __float_as_uint(a[i]);
    bad |= (bits & 0x7f800000u) == 0x7f800000u;  // Inf or NaN
  }
  if (__ballot(bad) != 0ull && (threadIdx.x & 63) == 0) atomicAnd(ok, 0);
}
// g[0] = NaN when any statistic is not finite: a rank-local verdict folded into the gradient bucket BEFORE the all-reduce, so that every rank's
// orcai_step_ok sees it in the summed gradient and all replicas void the step together
__global__ __launch_bounds__(256) void poison_if_nonfinite_kernel(const float* __restrict__ a, int64_t n, float* __restrict__ g) {
  bool bad = false;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const uint32_t bits = __float_as_uint(a[i]);
    bad |= (bits & 0x7f800000u) == 0x7f800000u;
  }
  if (__ballot(bad) != 0ull && (threadIdx.x & 63) == 0) g[0] = __uint_as_float(0x7fc00000u);
}
__global__ void count_skipped_kernel(const int32_t* ok, int64_t* skipped) {
  if (threadIdx.x == 0 && blockIdx.x == 0 && !ok[0]) skipped[0] += 1;
}

// sum of squares (L2 penalty value): out += lambda * sum w^2
__global__ __launch_bounds__(256) void l2_value_kernel(const float* __restrict__ w, int64_t n, float lambda, double* __restrict__ out) {
  __shared__ double s[256];
  double a = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) a += (double)w[i] * (double)w[i];
  s[threadIdx.x] = a;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicAdd(out, (double)lambda * s[0]);
}

// Adam (Keras 3 form): m += (g - m)(1 - b1); v += (g^2 - v)(1 - b2); w -= alpha * m / (sqrt(v) + eps),  alpha = lr*sqrt(1-b2^t)/(1-b1^t)
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                    int64_t n, float alpha, float b1, float b2, float eps, float gscale) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i] * gscale;
  const float mi = m[i] + (gi - m[i]) * (1.0f - b1);
  const float vi = v[i] + (gi * gi - v[i]) * (1.0f - b2);
  m[i] = mi;
  v[i] = vi;
  w[i] = w[i] - alpha * mi / (sqrtf(vi) + eps);
}

// =========================================================================================
// LSTM, training forward: the inference recurrence (model_fwd.hip lstm_kernel) plus stores of the gate activations
// (i, f, g, o in the kernel's permuted column order) and of the cell state, needed by the backward pass.
// =========================================================================================
// v_rcp_f32 (1 ulp) instead of an IEEE division: the quotient cost ten instructions per gate value (v_div_scale / v_div_fmas / v_div_fixup around the
// same v_rcp) in a step whose SIMDs issue 96 % of the time, next to a fast exponential that is less accurate than either
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return fmaf(-2.0f, __builtin_amdgcn_rcpf(__expf(2.0f * x) + 1.0f), 1.0f); }


// The gate stage of a training-recurrence step, shared by the three kernels below.  After the MFMAs lane (lk, lj) holds, for its rows 4 lk + r, the
// pre-activations of column lj of the wave's two 16-column tiles: tile 0 = gate i of unit lj (lj < 8) or f of unit lj - 8, tile 1 = g or o.  Every lane
// activates its OWN eight values (and stores them: the gate tensor of the backward pass), then the two half-rows trade what the other needs -- i, g of rows
// 2, 3 go up, f, o of rows 0, 1 go down, four DPP row rotations -- and each finishes TWO (row, unit) cells: 10 exponential / reciprocal pairs per lane and
// step.  (The first version exchanged the pre-activations and let both half-rows evaluate all four gates of all four rows: 20 pairs, half of them thrown
// away, in a step whose SIMDs were busy 96 % of the time with exactly these.)  Same functions of the same numbers: the results are bit-identical.
__device__ __forceinline__ float dpp_xor8(float v) {  // lane l <- lane l ^ 8 (rotation by 8 within a row of 16 lanes)
  return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x128 /*row_ror:8*/, 0xf, 0xf, true));
}
template <class StoreGates, class StoreState>
__device__ __forceinline__ void lstm_gate_stage(const float (&z0)[4], const float (&z1)[4], float (&cst)[4], int lj, StoreGates store_gates, StoreState store_state) {
  const bool low = lj < 8;
  const float m1 = low ? 2.0f : -1.0f;  // tile 1: tanh (g) on the low half-row, sigmoid (o) on the high one -- one exponential / reciprocal either way
  float a0[4], a1[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    a0[r] = sigmoidf_(z0[r]);
    const float rc = __builtin_amdgcn_rcpf(__expf(m1 * z1[r]) + 1.0f);
    a1[r] = low ? fmaf(-2.0f, rc, 1.0f) : rc;  // tanhf_ / sigmoidf_ to the bit
    store_gates(r, a0[r], a1[r]);
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const float got0 = dpp_xor8(low ? a0[2 + j] : a0[j]), got1 = dpp_xor8(low ? a1[2 + j] : a1[j]);
    const float gi = low ? a0[j] : got0, gf = low ? got0 : a0[2 + j];
    const float gg = low ? a1[j] : got1, go = low ? got1 : a1[2 + j];
    const float c = gf * cst[j] + gi * gg;
    const float h = go * tanhf_(c);
    cst[j] = c;
    store_state(j, low, h, c);  // row 4 lk + (low ? j : 2 + j); j is a compile-time constant after unrolling
  }
}

template <int U>
__global__ __launch_bounds__(U * 8) void lstm_train_fwd_kernel(const float* __restrict__ xz /*[B][T][2][4U] permuted*/, const float* __restrict__ Uw /*[2][U][4U] permuted*/,
                                                                int B, int T, float* __restrict__ out /*[B][T][2U]*/,
                                                                float* __restrict__ gates /*[B][T][2][4U] permuted: i f g o activations*/,
                                                                float* __restrict__ cstate /*[B][T][2][U]*/) {
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
      const float a = hbuf[cur][lj][kk * 4 + lk];
      acc[0] = mfma16(a, ufrag[0][kk], acc[0]);
      acc[1] = mfma16(a, ufrag[1][kk], acc[1]);
    }
    {
      const float z0[4] = {acc[0][0], acc[0][1], acc[0][2], acc[0][3]}, z1[4] = {acc[1][0], acc[1][1], acc[1][2], acc[1][3]};
      lstm_gate_stage(
          z0, z1, cst, lj,
          [&](int r, float g0, float g1) {  // this lane's own two gate columns: tile 0 column lj (i or f), tile 1 column lj (g or o)
            const int bb = b0 + lk * 4 + r;
            if (bb < B) {
              float* gp = gates + (((int64_t)bb * T + t) * 2 + dir) * (4 * U) + wave * 32 + lj;
              gp[0] = g0;
              gp[16] = g1;
            }
          },
          [&](int j, bool low, float h, float c) {
            const int row = lk * 4 + (low ? j : 2 + j), bb = b0 + row;
            hbuf[cur ^ 1][row][unit] = h;
            if (bb < B) {
              out[((int64_t)bb * T + t) * (2 * U) + dir * U + unit] = h;
              cstate[(((int64_t)bb * T + t) * 2 + dir) * U + unit] = c;
            }
          });
    }
    __syncthreads();
  }
}

// The f16 path's training recurrence (BASELINE configs[4]): the same kernel with the recurrent product h U on
// v_mfma_f32_16x16x32_f16 -- h is kept in LDS as f16 (A operand: lane l reads the 8 consecutive k of row l & 15 as one 16-byte
// read), the wave's 32 gate columns of U as f16 B fragments in 32 registers, f32 accumulation on top of the f32 input projection,
// gate arithmetic, cell state and every stored tensor in f32.  Eight MFMAs of 16 cycles per step instead of 64 of 32: the step is no
// longer bound by the one compute unit's f32 MFMA rate.
typedef _Float16 lh16;
typedef lh16 lh16x8 __attribute__((ext_vector_type(8)));

template <int U>
__global__ __launch_bounds__(U * 8) void lstm_train_fwd_h_kernel(const float* __restrict__ xz /*[B][T][2][4U] permuted*/, const float* __restrict__ Uw /*[2][U][4U] permuted*/,
                                                                  int B, int T, float* __restrict__ out /*[B][T][2U]*/, float* __restrict__ gates, float* __restrict__ cstate) {
  constexpr int KB = U / 32, HPh = U + 8;  // k blocks of 32; row pitch in halves (16-byte multiple)
  __shared__ __attribute__((aligned(16))) lh16 hbuf[2][16][HPh];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lk = lane >> 4, lj = lane & 15;
  const int dir = blockIdx.y;
  const int b0 = blockIdx.x * 16;
  const float* Ud = Uw + (int64_t)dir * U * 4 * U;
  lh16x8 ufrag[2][KB];  // B[k = 32 kb + 8 lk + e][col = lj] of gate-column tile nt
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
      for (int e = 0; e < 8; ++e) ufrag[nt][kb][e] = (lh16)Ud[(int64_t)(kb * 32 + lk * 8 + e) * (4 * U) + wave * 32 + nt * 16 + lj];
  for (int i = tid; i < 2 * 16 * HPh; i += U * 8) (&hbuf[0][0][0])[i] = (lh16)0.0f;
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
    if (step + 1 < T) load_xz(dir ? (T - 2 - step) : step + 1);
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const lh16x8 a = *reinterpret_cast<const lh16x8*>(&hbuf[cur][lj][kb * 32 + lk * 8]);  // A[row = batch lj][k = 32 kb + 8 lk + e]
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, ufrag[0][kb], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, ufrag[1][kb], acc[1], 0, 0, 0);
    }
    {
      const float z0[4] = {acc[0][0], acc[0][1], acc[0][2], acc[0][3]}, z1[4] = {acc[1][0], acc[1][1], acc[1][2], acc[1][3]};
      lstm_gate_stage(
          z0, z1, cst, lj,
          [&](int r, float g0, float g1) {
            const int bb = b0 + lk * 4 + r;
            if (bb < B) {
              float* gp = gates + (((int64_t)bb * T + t) * 2 + dir) * (4 * U) + wave * 32 + lj;
              gp[0] = g0;
              gp[16] = g1;
            }
          },
          [&](int j, bool low, float h, float c) {
            const int row = lk * 4 + (low ? j : 2 + j), bb = b0 + row;
            hbuf[cur ^ 1][row][unit] = (lh16)h;
            if (bb < B) {
              out[((int64_t)bb * T + t) * (2 * U) + dir * U + unit] = h;
              cstate[(((int64_t)bb * T + t) * 2 + dir) * U + unit] = c;
            }
          });
    }
    __syncthreads();
  }
}

// The f32 path's training recurrence with the recurrent product on f16 MFMA at f32 accuracy.  A step of lstm_train_fwd_kernel is bound by ONE
// compute unit's f32 MFMA rate (64 x v_mfma_f32_16x16x4_f32 of 32 cycles per wave and step, four waves per SIMD: 3.4 of the step's ~7 us), and
// training at batch 64 keeps 8 compute units busy.  Here every operand is split into two f16 numbers, x = hi + lo / 4096 with hi = f16(x) and
// lo = f16((x - hi) * 4096) (the scaling keeps lo a NORMAL f16 number for |x| down to 2^-14 / 4096; a product of two f16 numbers is exact in the
// MFMA's f32 accumulator), and the product h U is the sum of hi*hi (one accumulator, started from the input projection) and hi*lo + lo*hi (a
// second accumulator, scaled back by 2^-12 at the end); the dropped lo*lo term is 2^-24 of the product -- the rounding of an f32 multiply.
// 24 MFMAs of 16 cycles per wave and step instead of 64 of 32.  h in [-1, 1] and the recurrent weights are O(0.1): no range issue.  Gate
// arithmetic, cell state and every stored tensor in f32 as before.
// x = hi + lo / 4096 with two f16 numbers that are NORMAL (or zero) whatever the matrix unit does with f16 denormals: a value below the smallest
// normal f16 number travels in lo alone.  Exact to 2^-22 relative for |x| >= 2^-14, to 2^-11 relative (2^-25 of an operand scaled to 1) below.
__device__ __forceinline__ void split_f16(float x, lh16& hi, lh16& lo) {
  const lh16 h = fabsf(x) >= 6.103515625e-05f ? (lh16)x : (lh16)0.0f;
  hi = h;
  lo = (lh16)((x - (float)h) * 4096.0f);
}

template <int U>
__global__ __launch_bounds__(U * 8) void lstm_train_fwd_split_kernel(const float* __restrict__ xz /*[B][T][2][4U] permuted*/, const float* __restrict__ Uw /*[2][U][4U] permuted*/,
                                                                      int B, int T, float* __restrict__ out /*[B][T][2U]*/, float* __restrict__ gates, float* __restrict__ cstate) {
  constexpr int KB = U / 32, HPh = U + 8;  // k blocks of 32; row pitch in halves (16-byte multiple)
  constexpr float LO_INV = 1.0f / 4096.0f;
  __shared__ __attribute__((aligned(16))) lh16 hhi[2][16][HPh];
  __shared__ __attribute__((aligned(16))) lh16 hlo[2][16][HPh];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lk = lane >> 4, lj = lane & 15;
  const int dir = blockIdx.y;
  const int b0 = blockIdx.x * 16;
  const float* Ud = Uw + (int64_t)dir * U * 4 * U;
  lh16x8 uhi[2][KB], ulo[2][KB];  // B[k = 32 kb + 8 lk + e][col = lj] of gate-column tile nt
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        lh16 hi, lo;
        split_f16(Ud[(int64_t)(kb * 32 + lk * 8 + e) * (4 * U) + wave * 32 + nt * 16 + lj], hi, lo);
        uhi[nt][kb][e] = hi;
        ulo[nt][kb][e] = lo;
      }
  for (int i = tid; i < 2 * 16 * HPh; i += U * 8) { (&hhi[0][0][0])[i] = (lh16)0.0f; (&hlo[0][0][0])[i] = (lh16)0.0f; }
  float cst[4] = {0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  const int unit = wave * 8 + (lj & 7);
  // Addresses hoisted out of the step loop (PMC: 466 vector instructions per wave and step, two fifths of them 64-bit index arithmetic of the 8
  // loads and 12 stores, on SIMDs that issue 96 % of the time): one pointer per row and tensor, a wave-uniform offset per step.
  // (element offsets in 32 bits -- one register per row and layout; the launcher refuses batches whose tensors exceed 2^31 elements)
  uint32_t xo[4], oo[4], co[4];  // xz and gates share a layout
  bool rowok[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int bb = b0 + lk * 4 + r;
    rowok[r] = bb < B;
    const uint32_t bc = rowok[r] ? (uint32_t)bb : 0u;
    xo[r] = (bc * (uint32_t)T * 2u + (uint32_t)dir) * (uint32_t)(4 * U) + (uint32_t)(wave * 32 + lj);
    oo[r] = bc * (uint32_t)T * (uint32_t)(2 * U) + (uint32_t)(dir * U + wave * 8 + (lj & 7));
    co[r] = (bc * (uint32_t)T * 2u + (uint32_t)dir) * (uint32_t)U + (uint32_t)(wave * 8 + (lj & 7));
  }
  f32x4 xz_next[2];
  auto load_xz = [&](int tt) {
    const uint32_t to = (uint32_t)tt * (uint32_t)(8 * U);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) xz_next[nt][r] = rowok[r] ? xz[xo[r] + to + (uint32_t)(nt * 16)] : 0.0f;
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
      const lh16x8 ah = *reinterpret_cast<const lh16x8*>(&hhi[cur][lj][kb * 32 + lk * 8]);  // A[row = batch lj][k = 32 kb + 8 lk + e]
      const lh16x8 al = *reinterpret_cast<const lh16x8*>(&hlo[cur][lj][kb * 32 + lk * 8]);
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
      lstm_gate_stage(
          z0, z1, cst, lj,
          [&](int r, float g0, float g1) {
            if (rowok[r]) {
              float* gp = gates + (xo[r] + (uint32_t)t * (uint32_t)(8 * U));
              gp[0] = g0;
              gp[16] = g1;
            }
          },
          [&](int j, bool low, float h, float c) {
            const int row = lk * 4 + (low ? j : 2 + j);
            lh16 hh, hl;
            split_f16(h, hh, hl);
            hhi[cur ^ 1][row][unit] = hh;
            hlo[cur ^ 1][row][unit] = hl;
            if (low ? rowok[j] : rowok[2 + j]) {
              out[(low ? oo[j] : oo[2 + j]) + (uint32_t)t * (uint32_t)(2 * U)] = h;
              cstate[(low ? co[j] : co[2 + j]) + (uint32_t)t * (uint32_t)(2 * U)] = c;
            }
          });
    }
    __syncthreads();
  }
}

// =========================================================================================
// LSTM backward through time for one direction and 16 snippets.  Per step (walking the forward order backwards):
//   dh = dH[t] + dz[t_next] U^T
//   do = dh tanh(c); dc += dh o (1 - tanh^2 c); di = dc g; dg = dc i; df = dc c_prev; dc_prev = dc f
//   dz = (di i(1-i), df f(1-f), dg (1-g^2), do o(1-o))  -> dxz[t] (permuted columns), and the next recurrent term.
// U/16 waves.  Wave w owns units [16w, 16w+16): lane (lk, lj) does the gate arithmetic of unit 16w+lj for rows 4lk..4lk+3,
// and the wave computes the 16 x 16 tile dh[:, 16w..16w+16) = dz[16 x 4U] U^T of the next step with the whole 4U-long
// contraction in its own MFMA chain (k-step s takes columns {lk*U + s}, so A fragments are contiguous float4 LDS reads
// and the U^T fragments -- 4U/4 registers -- stay resident).  The D tile comes out exactly in the (row, unit) lane layout
// of the gate arithmetic, so the recurrent term never leaves registers; dz of all waves is exchanged through a
// double-buffered LDS tile with one barrier per step.  (LDS float atomics were measured at ~3 clk per lane and are avoided.)
// =========================================================================================
template <int U>
__global__ __launch_bounds__(U * 4) void lstm_bwd_kernel(const float* __restrict__ dH /*[B][T][2U]*/, const float* __restrict__ gates, const float* __restrict__ cstate,
                                                          const float* __restrict__ Uw /*[2][U][4U] permuted*/, int B, int T,
                                                          float* __restrict__ dxz /*[B][T][2][4U] permuted*/) {
  constexpr int ZP = 4 * U + 4;  // pitch = 4 (mod 32): the 8 lanes of one float4 LDS read phase cover all 32 banks
  __shared__ __attribute__((aligned(16))) float dzs[2][16][ZP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lk = lane >> 4, lj = lane & 15;
  const int dir = blockIdx.y;
  const int b0 = blockIdx.x * 16;
  const float* Ud = Uw + (int64_t)dir * U * 4 * U;
  const int unit = wave * 16 + lj;
  // B operand of k-step s: B[k = lk][j = lj] = U[unit][p = lk*U + s]
  float ut[U];
#pragma unroll
  for (int s = 0; s < U; ++s) ut[s] = Ud[(int64_t)unit * (4 * U) + lk * U + s];
  float dc[4] = {0.f, 0.f, 0.f, 0.f}, dhr[4] = {0.f, 0.f, 0.f, 0.f};
  const int pl = wave * 64 + (lj >> 3) * 32 + (lj & 7);  // permuted column of gate i of this unit; f, g, o follow at +8, +16, +24
  // register-prefetched operands of the current step: gates (i,f,g,o), c, c_prev, dH for the lane's 4 batch rows
  float pg[4][4], pc[4], pcp[4], pdh[4];
  auto load_step = [&](int step) {
    const int t = dir ? step : (T - 1 - step);
    const int tprev = dir ? t + 1 : t - 1;
    const bool has_prev = dir ? (t + 1 < T) : (t > 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int bb = b0 + lk * 4 + r;
      const bool ok = bb < B && step < T;
      const int64_t gbase = ok ? (((int64_t)bb * T + t) * 2 + dir) * (4 * U) + pl : 0;
      pg[r][0] = ok ? gates[gbase] : 0.f; pg[r][1] = ok ? gates[gbase + 8] : 0.f;
      pg[r][2] = ok ? gates[gbase + 16] : 0.f; pg[r][3] = ok ? gates[gbase + 24] : 0.f;
      pc[r] = ok ? cstate[(((int64_t)bb * T + t) * 2 + dir) * U + unit] : 0.f;
      pcp[r] = (ok && has_prev) ? cstate[(((int64_t)bb * T + tprev) * 2 + dir) * U + unit] : 0.f;
      pdh[r] = ok ? dH[((int64_t)bb * T + t) * (2 * U) + dir * U + unit] : 0.f;
    }
  };
  load_step(0);
  for (int step = 0; step < T; ++step) {
    const int t = dir ? step : (T - 1 - step);       // reverse of the forward order
    const int cur = step & 1;
    float cg[4][4], cc[4], ccp[4], cdh[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      cc[r] = pc[r]; ccp[r] = pcp[r]; cdh[r] = pdh[r];
#pragma unroll
      for (int q = 0; q < 4; ++q) cg[r][q] = pg[r][q];
    }
    load_step(step + 1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float gi = cg[r][0], gf = cg[r][1], gg = cg[r][2], go = cg[r][3];
      const float c = cc[r], cp = ccp[r];
      const float dh = cdh[r] + dhr[r];
      const float tc = tanhf_(c);
      const float dO = dh * tc;
      const float dct = dc[r] + dh * go * (1.0f - tc * tc);
      dc[r] = dct * gf;
      float* zr = &dzs[cur][lk * 4 + r][pl];  // rows past B carry zero gates, hence zero dz
      zr[0] = dct * gg * gi * (1.0f - gi);
      zr[8] = dct * cp * gf * (1.0f - gf);
      zr[16] = dct * gi * (1.0f - gg * gg);
      zr[24] = dO * go * (1.0f - go);
    }
    __syncthreads();  // dz[cur] of every wave is visible; buffer cur^1 is free again after the next barrier
    // dxz[t] rows of this wave's 64 columns: 16 rows x 16 float4, four per lane, 256 B contiguous per row
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = i * 4 + lk, bb = b0 + row;
      const float4 v = *reinterpret_cast<const float4*>(&dzs[cur][row][wave * 64 + lj * 4]);
      if (bb < B) *reinterpret_cast<float4*>(dxz + (((int64_t)bb * T + t) * 2 + dir) * (4 * U) + wave * 64 + lj * 4) = v;
    }
    // recurrent term of the next step: D[batch][unit] = sum_p dz[batch][p] U[unit][p], four interleaved accumulator chains
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* arow = &dzs[cur][lj][lk * U];  // A[i = batch lj][k = lk] of k-step s = dz[lj][lk*U + s]
#pragma unroll
    for (int s4 = 0; s4 < U / 4; ++s4) {
      const float4 a = *reinterpret_cast<const float4*>(arow + 4 * s4);
      acc[0] = mfma16(a.x, ut[4 * s4 + 0], acc[0]);
      acc[1] = mfma16(a.y, ut[4 * s4 + 1], acc[1]);
      acc[2] = mfma16(a.z, ut[4 * s4 + 2], acc[2]);
      acc[3] = mfma16(a.w, ut[4 * s4 + 3], acc[3]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) dhr[r] = (acc[0][r] + acc[1][r]) + (acc[2][r] + acc[3][r]);  // D row 4lk+r, col lj: this lane's (row, unit)
  }
}

// The f16 path's twin: the recurrent term dz U^T (contraction over the 4U gate columns) on v_mfma_f32_16x16x32_f16.  dz goes to LDS
// twice -- in f32 for the coalesced dxz rows, in f16 as the MFMA's A operand (lane l: row l & 15, the 8 consecutive columns
// 32 kb + 8 (l >> 4) of k block kb: one 16-byte read) -- and the wave's 16 units of U^T live in 4U / 32 f16 fragment registers
// quadruples (64 VGPRs instead of 128).  4U / 32 MFMAs of 16 cycles per step instead of U of 32; gate arithmetic, cell-state
// gradient and dxz in f32.
template <int U>
__global__ __launch_bounds__(U * 4) void lstm_bwd_h_kernel(const float* __restrict__ dH /*[B][T][2U]*/, const float* __restrict__ gates, const float* __restrict__ cstate,
                                                            const float* __restrict__ Uw /*[2][U][4U] permuted*/, int B, int T,
                                                            float* __restrict__ dxz /*[B][T][2][4U] permuted*/) {
  constexpr int ZP = 4 * U + 4, ZH = 4 * U + 8, KB = 4 * U / 32;
  extern __shared__ __attribute__((aligned(16))) float smem_lb[];
  float (*dzs)[16][ZP] = reinterpret_cast<float (*)[16][ZP]>(smem_lb);              // [2][16][ZP] f32: dz rows of the step (for dxz)
  lh16 (*dzh)[16][ZH] = reinterpret_cast<lh16 (*)[16][ZH]>(smem_lb + 2 * 16 * ZP);  // [2][16][ZH] f16: the MFMA's A operand
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lk = lane >> 4, lj = lane & 15;
  const int dir = blockIdx.y;
  const int b0 = blockIdx.x * 16;
  const float* Ud = Uw + (int64_t)dir * U * 4 * U;
  const int unit = wave * 16 + lj;
  lh16x8 ut[KB];  // B[k = 32 kb + 8 lk + e][col = lj] = U[unit][p = k]
#pragma unroll
  for (int kb = 0; kb < KB; ++kb)
#pragma unroll
    for (int e = 0; e < 8; ++e) ut[kb][e] = (lh16)Ud[(int64_t)unit * (4 * U) + kb * 32 + lk * 8 + e];
  float dc[4] = {0.f, 0.f, 0.f, 0.f}, dhr[4] = {0.f, 0.f, 0.f, 0.f};
  const int pl = wave * 64 + (lj >> 3) * 32 + (lj & 7);
  float pg[4][4], pc[4], pcp[4], pdh[4];
  auto load_step = [&](int step) {
    const int t = dir ? step : (T - 1 - step);
    const int tprev = dir ? t + 1 : t - 1;
    const bool has_prev = dir ? (t + 1 < T) : (t > 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int bb = b0 + lk * 4 + r;
      const bool ok = bb < B && step < T;
      const int64_t gbase = ok ? (((int64_t)bb * T + t) * 2 + dir) * (4 * U) + pl : 0;
      pg[r][0] = ok ? gates[gbase] : 0.f; pg[r][1] = ok ? gates[gbase + 8] : 0.f;
      pg[r][2] = ok ? gates[gbase + 16] : 0.f; pg[r][3] = ok ? gates[gbase + 24] : 0.f;
      pc[r] = ok ? cstate[(((int64_t)bb * T + t) * 2 + dir) * U + unit] : 0.f;
      pcp[r] = (ok && has_prev) ? cstate[(((int64_t)bb * T + tprev) * 2 + dir) * U + unit] : 0.f;
      pdh[r] = ok ? dH[((int64_t)bb * T + t) * (2 * U) + dir * U + unit] : 0.f;
    }
  };
  load_step(0);
  for (int step = 0; step < T; ++step) {
    const int t = dir ? step : (T - 1 - step);
    const int cur = step & 1;
    float cg[4][4], cc[4], ccp[4], cdh[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      cc[r] = pc[r]; ccp[r] = pcp[r]; cdh[r] = pdh[r];
#pragma unroll
      for (int q = 0; q < 4; ++q) cg[r][q] = pg[r][q];
    }
    load_step(step + 1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float gi = cg[r][0], gf = cg[r][1], gg = cg[r][2], go = cg[r][3];
      const float c = cc[r], cp = ccp[r];
      const float dh = cdh[r] + dhr[r];
      const float tc = tanhf_(c);
      const float dO = dh * tc;
      const float dct = dc[r] + dh * go * (1.0f - tc * tc);
      dc[r] = dct * gf;
      const float z0 = dct * gg * gi * (1.0f - gi), z1 = dct * cp * gf * (1.0f - gf), z2 = dct * gi * (1.0f - gg * gg), z3 = dO * go * (1.0f - go);
      float* zr = &dzs[cur][lk * 4 + r][pl];
      zr[0] = z0; zr[8] = z1; zr[16] = z2; zr[24] = z3;
      lh16* zh = &dzh[cur][lk * 4 + r][pl];
      zh[0] = (lh16)z0; zh[8] = (lh16)z1; zh[16] = (lh16)z2; zh[24] = (lh16)z3;
    }
    __syncthreads();  // dz[cur] of every wave is visible (f32 rows and f16 operand); the buffers cur^1 are free again after the next barrier
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = i * 4 + lk, bb = b0 + row;
      const float4 v = *reinterpret_cast<const float4*>(&dzs[cur][row][wave * 64 + lj * 4]);
      if (bb < B) *reinterpret_cast<float4*>(dxz + (((int64_t)bb * T + t) * 2 + dir) * (4 * U) + wave * 64 + lj * 4) = v;
    }
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const lh16x8 a = *reinterpret_cast<const lh16x8*>(&dzh[cur][lj][kb * 32 + lk * 8]);
      acc[kb & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, ut[kb], acc[kb & 3], 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) dhr[r] = (acc[0][r] + acc[1][r]) + (acc[2][r] + acc[3][r]);
  }
}

// The backward recurrence of the f32 path on split-f16 MFMA (see lstm_train_fwd_split_kernel).  The gradient dz has no natural scale, so the
// launch first takes max |dH| over the incoming gradient (lstm_grad_scale_kernel) and every dz is multiplied by the power of two S that brings
// that maximum into [0.5, 1) before it is split: hi = f16(dz S) (zero below the smallest normal f16), lo = f16((dz S - hi) 4096); the recurrent
// term is (acc_hi + acc_lo / 4096) / S.  A dz smaller than 2^-26 of the largest incoming gradient is lost -- the resolution an f32 accumulation of
// the same dot product has.  dxz (the gradient the weight-gradient GEMMs read) is written from the f32 values, untouched by the split.
__device__ float g_lstm_grad_scale[2];  // S, 1 / S of the launch in flight on this device (one backward recurrence at a time per process)

__global__ __launch_bounds__(256) void lstm_grad_max_kernel(const float* __restrict__ dH, int64_t n, uint32_t* __restrict__ maxbits) {
  float m = 0.0f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float v = fabsf(dH[i]);
    m = v > m ? v : m;  // NaN never wins: a non-finite gradient leaves the scale at the finite maximum (the caller's step check sees it anyway)
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0 && m > 0.0f && m < INFINITY) atomicMax(maxbits, __float_as_uint(m));
}
__global__ void lstm_grad_scale_kernel(uint32_t* maxbits) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const uint32_t b = *maxbits;
  // S = 2^(126 - exponent(max)): max * S in [0.5, 1); max = 0 (no gradient at all) -> S = 1
  const int e = (int)((b >> 23) & 0xff);
  const float S = (b == 0u || e == 0) ? 1.0f : __uint_as_float((uint32_t)(253 - e < 1 ? 1 : (253 - e > 254 ? 254 : 253 - e)) << 23);
  g_lstm_grad_scale[0] = S;
  g_lstm_grad_scale[1] = 1.0f / S;
  *maxbits = 0u;  // ready for the next launch
}
__device__ uint32_t g_lstm_grad_maxbits;

template <int U>
__global__ __launch_bounds__(U * 4) void lstm_bwd_split_kernel(const float* __restrict__ dH /*[B][T][2U]*/, const float* __restrict__ gates, const float* __restrict__ cstate,
                                                                const float* __restrict__ Uw /*[2][U][4U] permuted*/, int B, int T,
                                                                float* __restrict__ dxz /*[B][T][2][4U] permuted*/) {
  constexpr int ZP = 4 * U + 4, ZH = 4 * U + 8, KB = 4 * U / 32;
  constexpr float LO_INV = 1.0f / 4096.0f;
  extern __shared__ __attribute__((aligned(16))) float smem_ls[];
  float (*dzs)[16][ZP] = reinterpret_cast<float (*)[16][ZP]>(smem_ls);                       // [2][16][ZP] f32: dz rows of the step (for dxz)
  lh16 (*dzh)[16][ZH] = reinterpret_cast<lh16 (*)[16][ZH]>(smem_ls + 2 * 16 * ZP);           // [2][16][ZH] f16: hi part of dz S, the MFMA's A operand
  lh16 (*dzl)[16][ZH] = reinterpret_cast<lh16 (*)[16][ZH]>(smem_ls + 2 * 16 * ZP + 16 * ZH);  // [2][16][ZH] f16: lo part (2 x 16 x ZH halves = 16 x ZH floats)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lk = lane >> 4, lj = lane & 15;
  const int dir = blockIdx.y;
  const int b0 = blockIdx.x * 16;
  const float* Ud = Uw + (int64_t)dir * U * 4 * U;
  const int unit = wave * 16 + lj;
  const float S = g_lstm_grad_scale[0], Sinv = g_lstm_grad_scale[1];
  lh16x8 uth[KB], utl[KB];  // B[k = 32 kb + 8 lk + e][col = lj] = U[unit][p = k], split
#pragma unroll
  for (int kb = 0; kb < KB; ++kb)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      lh16 hi, lo;
      split_f16(Ud[(int64_t)unit * (4 * U) + kb * 32 + lk * 8 + e], hi, lo);
      uth[kb][e] = hi;
      utl[kb][e] = lo;
    }
  float dc[4] = {0.f, 0.f, 0.f, 0.f}, dhr[4] = {0.f, 0.f, 0.f, 0.f};
  const int pl = wave * 64 + (lj >> 3) * 32 + (lj & 7);
  float pg[4][4], pc[4], pcp[4], pdh[4];
  // per-row element offsets once (32 bits: the launcher refuses tensors of 2^31 elements), a wave-uniform term per step -- the 28 loads of a step
  // cost one add each instead of a 64-bit index computation (the forward kernel's change)
  uint32_t go_[4], co_[4], ho_[4];
  bool rowok[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int bb = b0 + lk * 4 + r;
    rowok[r] = bb < B;
    const uint32_t bc = rowok[r] ? (uint32_t)bb : 0u;
    go_[r] = (bc * (uint32_t)T * 2u + (uint32_t)dir) * (uint32_t)(4 * U) + (uint32_t)pl;
    co_[r] = (bc * (uint32_t)T * 2u + (uint32_t)dir) * (uint32_t)U + (uint32_t)unit;
    ho_[r] = bc * (uint32_t)T * (uint32_t)(2 * U) + (uint32_t)(dir * U + unit);
  }
  auto load_step = [&](int step) {
    const int t = dir ? step : (T - 1 - step);
    const int tprev = dir ? t + 1 : t - 1;
    const bool has_prev = dir ? (t + 1 < T) : (t > 0);
    const bool in_t = step < T;
    const uint32_t tc = (uint32_t)(in_t ? t : 0), tp = (uint32_t)((in_t && has_prev) ? tprev : 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool ok = rowok[r] && in_t;
      const float* gp = gates + (go_[r] + tc * (uint32_t)(8 * U));
      pg[r][0] = ok ? gp[0] : 0.f; pg[r][1] = ok ? gp[8] : 0.f;
      pg[r][2] = ok ? gp[16] : 0.f; pg[r][3] = ok ? gp[24] : 0.f;
      pc[r] = ok ? cstate[co_[r] + tc * (uint32_t)(2 * U)] : 0.f;
      pcp[r] = (ok && has_prev) ? cstate[co_[r] + tp * (uint32_t)(2 * U)] : 0.f;
      pdh[r] = ok ? dH[ho_[r] + tc * (uint32_t)(2 * U)] : 0.f;
    }
  };
  load_step(0);
  for (int step = 0; step < T; ++step) {
    const int t = dir ? step : (T - 1 - step);
    const int cur = step & 1;
    float cg[4][4], cc[4], ccp[4], cdh[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      cc[r] = pc[r]; ccp[r] = pcp[r]; cdh[r] = pdh[r];
#pragma unroll
      for (int q = 0; q < 4; ++q) cg[r][q] = pg[r][q];
    }
    load_step(step + 1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float gi = cg[r][0], gf = cg[r][1], gg = cg[r][2], go = cg[r][3];
      const float c = cc[r], cp = ccp[r];
      const float dh = cdh[r] + dhr[r];
      const float tc = tanhf_(c);
      const float dO = dh * tc;
      const float dct = dc[r] + dh * go * (1.0f - tc * tc);
      dc[r] = dct * gf;
      const float z[4] = {dct * gg * gi * (1.0f - gi), dct * cp * gf * (1.0f - gf), dct * gi * (1.0f - gg * gg), dO * go * (1.0f - go)};
      float* zr = &dzs[cur][lk * 4 + r][pl];
      lh16* zh = &dzh[cur][lk * 4 + r][pl];
      lh16* zl = &dzl[cur][lk * 4 + r][pl];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        zr[8 * q] = z[q];
        lh16 hi, lo;
        // saturated at +-2^15 (the largest magnitude whose lo part still fits f16): under exploding gradients dz can exceed 2^16 x max|dH| --
        // the recurrent term is then clipped instead of turning hi into inf and the whole gradient into NaN (the f32-MFMA kernel stays finite too)
        split_f16(__builtin_amdgcn_fmed3f(z[q] * S, -32768.0f, 32768.0f), hi, lo);
        zh[8 * q] = hi;
        zl[8 * q] = lo;
      }
    }
    __syncthreads();  // dz[cur] of every wave is visible (f32 rows and the split operand); the buffers cur^1 are free again after the next barrier
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = i * 4 + lk, bb = b0 + row;
      const float4 v = *reinterpret_cast<const float4*>(&dzs[cur][row][wave * 64 + lj * 4]);
      if (bb < B) *reinterpret_cast<float4*>(dxz + (((int64_t)bb * T + t) * 2 + dir) * (4 * U) + wave * 64 + lj * 4) = v;
    }
    f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}}, acl[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const lh16x8 ah = *reinterpret_cast<const lh16x8*>(&dzh[cur][lj][kb * 32 + lk * 8]);
      const lh16x8 al = *reinterpret_cast<const lh16x8*>(&dzl[cur][lj][kb * 32 + lk * 8]);
      acc[kb & 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, uth[kb], acc[kb & 1], 0, 0, 0);
      acl[kb & 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, utl[kb], acl[kb & 1], 0, 0, 0);
      acl[kb & 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, uth[kb], acl[kb & 1], 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) dhr[r] = fmaf(acl[0][r] + acl[1][r], LO_INV, acc[0][r] + acc[1][r]) * Sinv;
  }
}

// h_prev[b][t][dir][u] = h[b][t -+ 1][dir*U + u] (0 at the sequence start of each direction): left operand of dU = h_prev^T dxz
__global__ __launch_bounds__(256) void lstm_hprev_kernel(const float* __restrict__ h /*[B][T][2U]*/, int B, int T, int U, float* __restrict__ hp /*[B][T][2][U]*/) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t n = (int64_t)B * T * 2 * U;
  if (i >= n) return;
  const int u = (int)(i % U);
  const int dir = (int)((i / U) % 2);
  const int t = (int)((i / (2 * U)) % T);
  const int64_t b = i / ((int64_t)2 * U * T);
  const int tp = dir ? t + 1 : t - 1;
  hp[i] = (tp >= 0 && tp < T) ? h[(b * T + tp) * (2 * U) + dir * U + u] : 0.0f;
}

// =========================================================================================
// ResNet1DConv head, backward (architectures.py:100-115): ReduceFrequencyMean and Conv1D(num_labels, k, "same").
// =========================================================================================
// dfeat[m][x*C + c] = dfm[m][c] / W
__global__ __launch_bounds__(256) void freq_mean_bwd_kernel(const float* __restrict__ dfm, int64_t M, int W, int C, float* __restrict__ dfeat) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= M * W * C) return;
  const int64_t m = i / ((int64_t)W * C);
  const int c = (int)(i % C);
  dfeat[i] = dfm[m * C + c] / (float)W;
}

// dW[k][c][l] += sum_{b,t} x[b][t + k - left][c] * dz[b][t][l]      (one thread per weight; 2944 terms at batch 64)
__global__ __launch_bounds__(256) void conv1d_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dz, int B, int T, int C, int K, int L,
                                                            float* __restrict__ dW) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= K * C * L) return;
  const int l = i % L, c = (i / L) % C, k = i / (L * C);
  const int left = (K - 1) / 2;
  const int t_lo = left - k > 0 ? left - k : 0;                   // t + k - left >= 0
  const int t_hi = T - 1 + left - k < T - 1 ? T - 1 + left - k : T - 1;  // t + k - left <= T - 1
  float s0 = 0.0f, s1 = 0.0f;
  for (int b = 0; b < B; ++b) {
    const float* xb = x + ((int64_t)b * T + (k - left)) * C + c;
    const float* zb = dz + (int64_t)b * T * L + l;
    int t = t_lo;
    for (; t + 1 <= t_hi; t += 2) {
      s0 = fmaf(xb[(int64_t)t * C], zb[(int64_t)t * L], s0);
      s1 = fmaf(xb[(int64_t)(t + 1) * C], zb[(int64_t)(t + 1) * L], s1);
    }
    if (t <= t_hi) s0 = fmaf(xb[(int64_t)t * C], zb[(int64_t)t * L], s0);
  }
  dW[i] += s0 + s1;
}

// dx[b][t][c] = sum_k sum_l W[k][c][l] * dz[b][t - k + left][l]
__global__ __launch_bounds__(256) void conv1d_dgrad_kernel(const float* __restrict__ w, const float* __restrict__ dz, int B, int T, int C, int K, int L,
                                                            float* __restrict__ dx) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)B * T * C) return;
  const int c = (int)(i % C), t = (int)((i / C) % T);
  const int64_t b = i / ((int64_t)C * T);
  const int left = (K - 1) / 2;
  float s = 0.0f;
  for (int k = 0; k < K; ++k) {
    const int tz = t - k + left;
    if (tz < 0 || tz >= T) continue;
    const float* wr = w + ((int64_t)k * C + c) * L;
    const float* zr = dz + (b * T + tz) * L;
    for (int l = 0; l < L; ++l) s = fmaf(wr[l], zr[l], s);
  }
  dx[i] = s;
}

inline unsigned blocks_for(int64_t n) { return (unsigned)((n + 255) / 256); }
// floor(i / d) == __umulhi(i, rows_magic(d)) while i * d < 2^32 (magic = ceil(2^32 / d): the excess i * (magic d - 2^32) / (d 2^32) stays below 1 / d); the launchers check it
double* rows_partials();
inline uint32_t rows_magic(int d) { return d <= 1 ? 0u : (uint32_t)(((1ull << 32) + (uint64_t)d - 1) / (uint64_t)d); }

// ---------------------------------------------------------------- LSTM weights: Keras layout <-> the recurrence kernels' gate-column order
// Kernel column p = 32 w + 16 nt + j holds Keras column (2 nt + (j >> 3)) u + 8 w + (j & 7)   (architectures.lstm_column_permutation):
// wave w of lstm_kernel owns all four gates of units [8w, 8w + 8).
__device__ __forceinline__ int lstm_perm(int p, int u) {
  const int w = p >> 5, nt = (p >> 4) & 1, j = p & 15;
  return (2 * nt + (j >> 3)) * u + 8 * w + (j & 7);
}

// desc[i] = {src offset (floats, into w), dst offset (elements), rows, u, ld_dst, col_off, mode}:
//   mode 0: f32  dst[r * ld_dst + col_off + p] = src[r * 4u + perm(p)]        (kernel / bias / recurrent matrices for the f32 kernels)
//   mode 1: f16  dst[(col_off + p) * ld_dst + r] = src[r * 4u + perm(p)]      (TRANSPOSED copy for orcai_h_gemm_bias_act; pad stays zero)
// One workgroup per (descriptor, row block): replaces the per-step torch index / cat / stack kernels.
__global__ __launch_bounds__(256) void pack_lstm_kernel(const float* __restrict__ w, const int* __restrict__ desc, float* __restrict__ out32,
                                                         _Float16* __restrict__ out16) {
  const int* d = desc + blockIdx.x * 7;
  const int rows = d[2], u = d[3], ld = d[4], col_off = d[5], mode = d[6];
  const float* src = w + d[0];
  const uint32_t n = (uint32_t)rows * 4u * (uint32_t)u, u4 = 4u * (uint32_t)u;  // (< 2^31: the launcher's descriptors are weight matrices; 32-bit index arithmetic)
  for (uint32_t i = blockIdx.y * 256 + threadIdx.x; i < n; i += gridDim.y * 256) {
    const int r = (int)(i / u4), p = (int)(i - (uint32_t)r * u4);
    const float v = src[(int64_t)r * 4 * u + lstm_perm(p, u)];
    if (mode == 0) out32[d[1] + (int64_t)r * ld + col_off + p] = v;
    else out16[d[1] + (int64_t)(col_off + p) * ld + r] = (_Float16)v;
  }
}

// G[r * 4u + perm(p)] = src[r * ld_src + col_off + p] + l2g * W[r * 4u + perm(p)]: a kernel-order gradient back into the Keras-layout
// gradient buffer, with the L2 regulariser's term (W may be NULL when l2g == 0).
__global__ __launch_bounds__(256) void unpack_lstm_grad_kernel(const float* __restrict__ src, int ld_src, int col_off, int rows, int u, float* __restrict__ G,
                                                                const float* __restrict__ W, float l2g) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)rows * 4 * u) return;
  const int r = (int)(i / (4 * u)), p = (int)(i - (int64_t)r * 4 * u);
  const int64_t k = (int64_t)r * 4 * u + lstm_perm(p, u);
  float v = src[(int64_t)r * ld_src + col_off + p];
  if (W) v = fmaf(l2g, W[k], v);
  G[k] = v;
}

// the same for up to 16 (source, destination) pairs in ONE launch (both BiLSTM layers of a step: 12 pairs), descriptors by value in the kernel arguments
struct UnpackBatch {
  orcai_unpack_desc d[16];
};
__global__ __launch_bounds__(256) void unpack_lstm_grads_kernel(UnpackBatch b, int u) {
  const orcai_unpack_desc d = b.d[blockIdx.y];
  const uint32_t n = (uint32_t)d.rows * 4u * (uint32_t)u, u4 = 4u * (uint32_t)u;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int r = (int)(i / u4), p = (int)(i - (uint32_t)r * u4);
    const int64_t k = (int64_t)r * 4 * u + lstm_perm(p, u);
    float v = d.src[(int64_t)r * d.ld_src + d.col_off + p];
    if (d.W) v = fmaf(d.l2g, d.W[k], v);
    d.G[k] = v;
  }
}

// out += lambda * sum w^2 over up to 8 slices of one buffer in ONE launch (blockIdx.y = slice)
struct L2Batch {
  int64_t off[8], n[8];
};
__global__ __launch_bounds__(256) void l2_values_kernel(const float* __restrict__ base, L2Batch b, float lambda, double* __restrict__ out) {
  __shared__ double s[256];
  const float* w = base + b.off[blockIdx.y];
  const int64_t n = b.n[blockIdx.y];
  double a = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) a += (double)w[i] * (double)w[i];
  s[threadIdx.x] = a;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicAdd(out, (double)lambda * s[0]);
}

// moving = moving * momentum + batch * (1 - momentum) over one flat buffer of all BatchNorm statistics
__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ moving, const float* __restrict__ batch, int n, float momentum, const int32_t* __restrict__ ok = nullptr) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (ok && !ok[0]) return;
  if (i < n) moving[i] = moving[i] * momentum + batch[i] * (1.0f - momentum);
}

// address of the row-BatchNorm slab partials on the current device (looked up once per device; no allocation)
double* rows_partials() {
  constexpr int MAXDEV = 64;
  static double* addr[MAXDEV] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return nullptr;
  if (!addr[dev] && hipGetSymbolAddress((void**)&addr[dev], HIP_SYMBOL(g_rows_partials)) != hipSuccess) return nullptr;
  return addr[dev];
}

}  // namespace

extern "C" {

int orcai_gemm_strided(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk, int64_t sbn, float* C, int M, int N, int K, float alpha,
                       int accumulate, const float* Wreg, float beta_w, void* stream) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return ORCAI_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  const int am = sam == 1 ? 0 : (sak == 1 ? 1 : -1);  // 0: m contiguous, 1: k contiguous
  const int bm = sbn == 1 ? 0 : (sbk == 1 ? 1 : -1);  // 0: n contiguous, 1: k contiguous
  if (am >= 0 && am == bm) {  // A^T B (weight gradients) or A B^T (input gradients)
    const int64_t lda = am == 0 ? sak : sam, ldb = bm == 0 ? sbk : sbn;
    const int gx = (N + 63) / 64, gy = (M + 63) / 64;
    int splits = 1;
    if (gx * gy < 768 && K >= 8 * GBK) {  // too few tiles to fill 256 CUs: split K
      splits = (1024 + gx * gy - 1) / (gx * gy);
      const int max_splits = K / (4 * GBK);
      if (splits > max_splits) splits = max_splits;
      if (splits < 1) splits = 1;
    }
    int kps = (K + splits - 1) / splits;
    kps = (kps + GBK - 1) / GBK * GBK;
    splits = (K + kps - 1) / kps;
    if (splits > 1 && !accumulate) {
      hipError_t e = orcai_zero::zero_async(C, sizeof(float) * (size_t)M * N, st);
      if (e != hipSuccess) return (int)e;
    }
    const int va = ((uintptr_t)A % 16 == 0 && lda % 4 == 0) ? 1 : 0, vb = ((uintptr_t)B % 16 == 0 && ldb % 4 == 0) ? 1 : 0;
    dim3 grid(gx, gy, splits);
    if (am == 0)
      hipLaunchKernelGGL((gemm_tiled_kernel<0, 0>), grid, dim3(256), 0, st, A, lda, B, ldb, C, M, N, K, alpha, accumulate, Wreg, beta_w, kps, va, vb);
    else
      hipLaunchKernelGGL((gemm_tiled_kernel<1, 1>), grid, dim3(256), 0, st, A, lda, B, ldb, C, M, N, K, alpha, accumulate, Wreg, beta_w, kps, va, vb);
    return (int)hipGetLastError();
  }
  dim3 grid((N + 63) / 64, (M + 63) / 64);
  hipLaunchKernelGGL(gemm_strided_kernel, grid, dim3(256), 0, st, A, sam, sak, B, sbk, sbn, C, M, N, K, alpha, accumulate, Wreg, beta_w);
  return (int)hipGetLastError();
}

int orcai_freq_mean_bwd(const float* dfm, int64_t M, int W, int C, float* dfeat, void* stream) {
  if (!dfm || !dfeat || M <= 0 || W <= 0 || C <= 0) return ORCAI_E_BADARG;
  hipLaunchKernelGGL(freq_mean_bwd_kernel, dim3(blocks_for(M * W * C)), dim3(256), 0, (hipStream_t)stream, dfm, M, W, C, dfeat);
  return (int)hipGetLastError();
}

int orcai_conv1d_bwd(const float* x, const float* w, const float* dz, int B, int T, int C, int K, int L, float* dW, float* dx, void* stream) {
  if (!x || !w || !dz || !dW || !dx || B <= 0 || T <= 0 || C <= 0 || K <= 0 || L <= 0) return ORCAI_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(conv1d_wgrad_kernel, dim3(blocks_for((int64_t)K * C * L)), dim3(256), 0, st, x, dz, B, T, C, K, L, dW);
  hipLaunchKernelGGL(conv1d_dgrad_kernel, dim3(blocks_for((int64_t)B * T * C)), dim3(256), 0, st, w, dz, B, T, C, K, L, dx);
  return (int)hipGetLastError();
}

int orcai_colsum(const float* x, int M, int C, float* out, int accumulate, void* stream) {
  if (!x || !out || M <= 0 || C <= 0) return ORCAI_E_BADARG;
  hipLaunchKernelGGL(colsum_kernel, dim3((C + 15) / 16), dim3(1024), 0, (hipStream_t)stream, x, M, C, out, accumulate);
  return (int)hipGetLastError();
}

int orcai_bn_rows_stats(const float* x, int M, int cols, int C, float* mean, float* var, void* stream) {
  if (!x || !mean || !var || M <= 0 || C <= 0 || cols % C) return ORCAI_E_BADARG;
  if (C <= ROWS_MAXC && cols <= ROWS_MAXCOLS && cols > C) {  // (one position per row -- Dense-128's BatchNorm -- is two launches of 8 us: the channel kernel's one is faster)  // coalesced row slabs + an ordered fold
    double* part = rows_partials();
    if (!part) return (int)hipErrorInvalidDevice;
    const int rps = (M + ROWS_SLABS - 1) / ROWS_SLABS, slabs = (M + rps - 1) / rps;
    hipLaunchKernelGGL(bn_rows_partial_kernel<false>, dim3(slabs), dim3(512), 0, (hipStream_t)stream, (const float*)nullptr, x, M, cols, C, rps, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0.0f, 0, part);
    hipLaunchKernelGGL(bn_rows_finish_kernel, dim3(1), dim3(512), 0, (hipStream_t)stream, part, C, slabs, (double)M * (double)(cols / C), 0, mean, var);
    return (int)hipGetLastError();
  }
  if ((int64_t)M * (cols / C) * (cols / C) >= (1ll << 32) || (int64_t)M * (cols / C) >= (1ll << 31)) return ORCAI_E_UNSUPPORTED;
  hipLaunchKernelGGL(bn_rows_stats_kernel, dim3(C), dim3(1024), 0, (hipStream_t)stream, x, M, cols, C, rows_magic(cols / C), mean, var);
  return (int)hipGetLastError();
}

int orcai_bn_rows_apply(const float* x, int M, int cols, int C, const float* mean, const float* var, const float* gamma, const float* beta, float eps,
                        int relu, float* y, void* stream) {
  if (!x || !y || !mean || !var || !gamma || !beta || M <= 0 || C <= 0 || cols % C) return ORCAI_E_BADARG;
  const int64_t n = (int64_t)M * cols;
  hipLaunchKernelGGL(bn_rows_apply_kernel, dim3((cols + 255) / 256, M < 65535 ? M : 65535), dim3(256), 0, (hipStream_t)stream, x, M, cols, C, mean, var, gamma, beta, eps, relu, y);
  return (int)hipGetLastError();
}

int orcai_bn_rows_bwd(const float* dy, const float* x, int M, int cols, int C, const float* mean, const float* var, const float* gamma, const float* beta,
                      float eps, int relu, float* dbeta, float* dgamma, float* dx, void* stream) {
  if (!dy || !x || !dx || !dbeta || !dgamma || M <= 0 || C <= 0 || cols % C) return ORCAI_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  if (C <= ROWS_MAXC && cols <= ROWS_MAXCOLS && cols > C) {  // (one position per row -- Dense-128's BatchNorm -- is two launches of 8 us: the channel kernel's one is faster)
    double* part = rows_partials();
    if (!part) return (int)hipErrorInvalidDevice;
    const int rps = (M + ROWS_SLABS - 1) / ROWS_SLABS, slabs = (M + rps - 1) / rps;
    hipLaunchKernelGGL(bn_rows_partial_kernel<true>, dim3(slabs), dim3(512), 0, st, dy, x, M, cols, C, rps, mean, var, gamma, beta, eps, relu, part);
    hipLaunchKernelGGL(bn_rows_finish_kernel, dim3(1), dim3(512), 0, st, part, C, slabs, 0.0, 1, dbeta, dgamma);
    hipLaunchKernelGGL(bn_rows_bwd_apply_kernel, dim3((cols + 255) / 256, M < 65535 ? M : 65535), dim3(256), 0, st, dy, x, M, cols, C, (float)((int64_t)M * (cols / C)), mean, var,
                       gamma, beta, eps, relu, dbeta, dgamma, dx);
    return (int)hipGetLastError();
  }
  if ((int64_t)M * (cols / C) * (cols / C) >= (1ll << 32) || (int64_t)M * (cols / C) >= (1ll << 31)) return ORCAI_E_UNSUPPORTED;
  hipLaunchKernelGGL(bn_rows_bwd_stats_kernel, dim3(C), dim3(1024), 0, st, dy, x, M, cols, C, rows_magic(cols / C), mean, var, gamma, beta, eps, relu, dbeta, dgamma);
  const float count = (float)((int64_t)M * (cols / C));
  hipLaunchKernelGGL(bn_rows_bwd_apply_kernel, dim3((cols + 255) / 256, M < 65535 ? M : 65535), dim3(256), 0, st, dy, x, M, cols, C, count, mean, var, gamma, beta, eps, relu, dbeta, dgamma, dx);
  return (int)hipGetLastError();
}

int orcai_mask_scale(const float* x, const float* mask, float scale, int64_t n, float* y, void* stream) {
  if (!x || !mask || !y || n <= 0) return ORCAI_E_BADARG;
  hipLaunchKernelGGL(mask_scale_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, x, mask, scale, n, y);
  return (int)hipGetLastError();
}

int orcai_relu_bwd(const float* dy, const float* y, int64_t n, float* dx, void* stream) {
  if (!dy || !y || !dx || n <= 0) return ORCAI_E_BADARG;
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, dy, y, n, dx);
  return (int)hipGetLastError();
}

int orcai_dropout_mask(float* mask, int64_t n, uint64_t seed, float keep, void* stream) {
  if (!mask || n <= 0) return ORCAI_E_BADARG;
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, mask, n, seed, keep);
  return (int)hipGetLastError();
}

int orcai_dropout_mask_dev(float* mask, int64_t n, const uint64_t* counter, uint64_t seed_add, float keep, void* stream) {
  if (!mask || !counter || n <= 0) return ORCAI_E_BADARG;
  hipLaunchKernelGGL(dropout_mask_dev_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, mask, n, counter, seed_add, keep);
  return (int)hipGetLastError();
}

int orcai_adam_step_dev(float* w, const float* g, float* m, float* v, int64_t n, const float* lr, float b1, float b2, float eps, const uint64_t* counter, float gscale,
                        void* stream) {
  if (!w || !g || !m || !v || !lr || !counter || n <= 0) return ORCAI_E_BADARG;
  hipLaunchKernelGGL(adam_dev_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, w, g, m, v, n, lr, b1, b2, eps, counter, gscale, (const int32_t*)nullptr);
  return (int)hipGetLastError();
}

int orcai_counter_advance(uint64_t* counter, void* stream) {
  if (!counter) return ORCAI_E_BADARG;
  hipLaunchKernelGGL(counter_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, counter, (const int32_t*)nullptr);
  return (int)hipGetLastError();
}

int orcai_masked_bce_w(const float* p, const float* y, int64_t n, float mask_value, double* acc3, float* dz, const float* loss_weight, float grad_scale,
                       void* stream) {
  if (!p || !y || !acc3 || n <= 0 || !(grad_scale > 0.0f)) return ORCAI_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = orcai_zero::zero_async(acc3, 3 * sizeof(double), st);
  if (e != hipSuccess) return (int)e;
  unsigned g = blocks_for(n);
  if (g > 256) g = 256;
  hipLaunchKernelGGL(bce_reduce_kernel, dim3(g), dim3(256), 0, st, p, y, n, mask_value, acc3, loss_weight);
  if (dz) hipLaunchKernelGGL(bce_grad_kernel, dim3(blocks_for(n)), dim3(256), 0, st, p, y, n, mask_value, acc3, dz, loss_weight, grad_scale);
  return (int)hipGetLastError();
}

int orcai_masked_bce(const float* p, const float* y, int64_t n, float mask_value, double* acc3, float* dz, void* stream) {
  return orcai_masked_bce_w(p, y, n, mask_value, acc3, dz, nullptr, 1.0f, stream);
}

int orcai_l2_value(const float* w, int64_t n, float lambda, double* out, void* stream) {
  if (!w || !out || n <= 0) return ORCAI_E_BADARG;
  unsigned g = blocks_for(n);
  if (g > 128) g = 128;
  hipLaunchKernelGGL(l2_value_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, w, n, lambda, out);
  return (int)hipGetLastError();
}

int orcai_adam_step(float* w, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps, int step, float gscale, void* stream) {
  if (!w || !g || !m || !v || n <= 0 || step < 1) return ORCAI_E_BADARG;
  const double alpha = (double)lr * sqrt(1.0 - pow((double)b2, step)) / (1.0 - pow((double)b1, step));
  hipLaunchKernelGGL(adam_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, w, g, m, v, n, (float)alpha, b1, b2, eps, gscale);
  return (int)hipGetLastError();
}

int orcai_lstm_train_fwd(const float* xz, const float* Uw, int B, int T, int units, float* out, float* gates, float* cstate, void* stream) {
  if (!xz || !Uw || !out || !gates || !cstate || B <= 0 || T <= 0) return ORCAI_E_BADARG;
  if ((int64_t)B * T * 8 * units >= (1ll << 31)) return ORCAI_E_UNSUPPORTED;  // 32-bit element offsets inside the kernels
  dim3 grid((B + 15) / 16, 2);
  hipStream_t st = (hipStream_t)stream;
  if (g_orcai_lstm_split) {
    switch (units) {
      case 128: hipLaunchKernelGGL(lstm_train_fwd_split_kernel<128>, grid, dim3(1024), 0, st, xz, Uw, B, T, out, gates, cstate); break;
      case 64: hipLaunchKernelGGL(lstm_train_fwd_split_kernel<64>, grid, dim3(512), 0, st, xz, Uw, B, T, out, gates, cstate); break;
      default: return ORCAI_E_UNSUPPORTED;
    }
    return (int)hipGetLastError();
  }
  switch (units) {
    case 128: hipLaunchKernelGGL(lstm_train_fwd_kernel<128>, grid, dim3(1024), 0, st, xz, Uw, B, T, out, gates, cstate); break;
    case 64: hipLaunchKernelGGL(lstm_train_fwd_kernel<64>, grid, dim3(512), 0, st, xz, Uw, B, T, out, gates, cstate); break;
    default: return ORCAI_E_UNSUPPORTED;
  }
  return (int)hipGetLastError();
}

int orcai_h_lstm_bwd(const float* dH, const float* gates, const float* cstate, const float* Uw, int B, int T, int units, float* dxz, void* stream) {
  if (!dH || !gates || !cstate || !Uw || !dxz || B <= 0 || T <= 0) return ORCAI_E_BADARG;
  dim3 grid((B + 15) / 16, 2);
  hipStream_t st = (hipStream_t)stream;
  if (units == 128) {
    const size_t lds = (size_t)2 * 16 * (4 * 128 + 4) * 4 + (size_t)2 * 16 * (4 * 128 + 8) * 2;  // 99 328 B > 64 KiB: opt in once per device
    static bool opted_dev[64] = {};
    int devid = 0;
    hipError_t e = hipGetDevice(&devid);
    if (e != hipSuccess) return (int)e;
    if (devid < 0 || devid >= 64) return ORCAI_E_UNSUPPORTED;
    if (!opted_dev[devid]) {
      e = hipFuncSetAttribute((const void*)lstm_bwd_h_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return (int)e;
      opted_dev[devid] = true;
    }
    hipLaunchKernelGGL(lstm_bwd_h_kernel<128>, grid, dim3(512), lds, st, dH, gates, cstate, Uw, B, T, dxz);
  } else if (units == 64) {
    const size_t lds = (size_t)2 * 16 * (4 * 64 + 4) * 4 + (size_t)2 * 16 * (4 * 64 + 8) * 2;  // 50 176 B
    hipLaunchKernelGGL(lstm_bwd_h_kernel<64>, grid, dim3(256), lds, st, dH, gates, cstate, Uw, B, T, dxz);
  } else {
    return ORCAI_E_UNSUPPORTED;
  }
  return (int)hipGetLastError();
}

int orcai_h_lstm_train_fwd(const float* xz, const float* Uw, int B, int T, int units, float* out, float* gates, float* cstate, void* stream) {
  if (!xz || !Uw || !out || !gates || !cstate || B <= 0 || T <= 0) return ORCAI_E_BADARG;
  dim3 grid((B + 15) / 16, 2);
  hipStream_t st = (hipStream_t)stream;
  switch (units) {
    case 128: hipLaunchKernelGGL(lstm_train_fwd_h_kernel<128>, grid, dim3(1024), 0, st, xz, Uw, B, T, out, gates, cstate); break;
    case 64: hipLaunchKernelGGL(lstm_train_fwd_h_kernel<64>, grid, dim3(512), 0, st, xz, Uw, B, T, out, gates, cstate); break;
    default: return ORCAI_E_UNSUPPORTED;
  }
  return (int)hipGetLastError();
}

int orcai_lstm_bwd(const float* dH, const float* gates, const float* cstate, const float* Uw, int B, int T, int units, float* dxz, void* stream) {
  if (!dH || !gates || !cstate || !Uw || !dxz || B <= 0 || T <= 0) return ORCAI_E_BADARG;
  if ((int64_t)B * T * 8 * units >= (1ll << 31)) return ORCAI_E_UNSUPPORTED;  // 32-bit element offsets inside the split kernel
  dim3 grid((B + 15) / 16, 2);
  hipStream_t st = (hipStream_t)stream;
  if (g_orcai_lstm_split && (units == 128 || units == 64)) {
    // per DEVICE: the address of the max accumulator (a __device__ symbol has one instance per device) and the > 64 KiB LDS opt-ins (function
    // attributes are per device too).  Looked up on a device's first call, which is never inside a stream capture (warm-up steps come first).
    // The scale itself travels through the device global g_lstm_grad_scale: ONE backward recurrence in flight per device (include/orcai_hip.h).
    constexpr int MAXDEV = 64;
    static uint32_t* maxbits_dev[MAXDEV] = {};
    static bool opted_dev[MAXDEV][2] = {};
    int devid = 0;
    hipError_t e = hipGetDevice(&devid);
    if (e != hipSuccess) return (int)e;
    if (devid < 0 || devid >= MAXDEV) return ORCAI_E_UNSUPPORTED;
    if (!maxbits_dev[devid]) {
      e = hipGetSymbolAddress((void**)&maxbits_dev[devid], HIP_SYMBOL(g_lstm_grad_maxbits));
      if (e != hipSuccess) return (int)e;
    }
    uint32_t* maxbits = maxbits_dev[devid];
    const int64_t n = (int64_t)B * T * 2 * units;
    hipLaunchKernelGGL(lstm_grad_max_kernel, dim3((unsigned)((n + 256 * 16 - 1) / (256 * 16))), dim3(256), 0, st, dH, n, maxbits);
    hipLaunchKernelGGL(lstm_grad_scale_kernel, dim3(1), dim3(64), 0, st, maxbits);
    const size_t lds = (size_t)2 * 16 * (4 * units + 4) * 4 + (size_t)2 * 2 * 16 * (4 * units + 8) * 2;  // f32 rows + hi + lo operands
    if (units == 128) {
      if (!opted_dev[devid][0]) {  // 132 352 B > 64 KiB: opt in once per device
        e = hipFuncSetAttribute((const void*)lstm_bwd_split_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        opted_dev[devid][0] = true;
      }
      hipLaunchKernelGGL(lstm_bwd_split_kernel<128>, grid, dim3(512), lds, st, dH, gates, cstate, Uw, B, T, dxz);
    } else {
      if (!opted_dev[devid][1]) {  // 66 816 B
        e = hipFuncSetAttribute((const void*)lstm_bwd_split_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        opted_dev[devid][1] = true;
      }
      hipLaunchKernelGGL(lstm_bwd_split_kernel<64>, grid, dim3(256), lds, st, dH, gates, cstate, Uw, B, T, dxz);
    }
    return (int)hipGetLastError();
  }
  switch (units) {
    case 128: hipLaunchKernelGGL(lstm_bwd_kernel<128>, grid, dim3(512), 0, st, dH, gates, cstate, Uw, B, T, dxz); break;
    case 64: hipLaunchKernelGGL(lstm_bwd_kernel<64>, grid, dim3(256), 0, st, dH, gates, cstate, Uw, B, T, dxz); break;
    default: return ORCAI_E_UNSUPPORTED;
  }
  return (int)hipGetLastError();
}

int orcai_pack_lstm(const float* w, const int* desc, int n_desc, float* out32, void* out16, void* stream) {
  if (!w || !desc || n_desc <= 0 || (!out32 && !out16)) return ORCAI_E_BADARG;
  hipLaunchKernelGGL(pack_lstm_kernel, dim3(n_desc, 64), dim3(256), 0, (hipStream_t)stream, w, desc, out32, (_Float16*)out16);
  return (int)hipGetLastError();
}

int orcai_unpack_lstm_grad(const float* src, int ld_src, int col_off, int rows, int units, float* G, const float* W, float l2g, void* stream) {
  if (!src || !G || rows <= 0 || units <= 0 || (units & 7) || col_off < 0 || ld_src < col_off + 4 * units) return ORCAI_E_BADARG;
  hipLaunchKernelGGL(unpack_lstm_grad_kernel, dim3(blocks_for((int64_t)rows * 4 * units)), dim3(256), 0, (hipStream_t)stream, src, ld_src, col_off, rows, units, G, W, l2g);
  return (int)hipGetLastError();
}

int orcai_unpack_lstm_grads(const orcai_unpack_desc* descs_host, int n, int units, void* stream) {
  if (!descs_host || n <= 0 || n > 16 || units <= 0 || (units & 7)) return ORCAI_E_BADARG;
  UnpackBatch b;
  int64_t most = 0;
  for (int i = 0; i < n; ++i) {
    const orcai_unpack_desc& d = descs_host[i];
    if (!d.src || !d.G || d.rows <= 0 || d.col_off < 0 || d.ld_src < d.col_off + 4 * units) return ORCAI_E_BADARG;
    b.d[i] = d;
    const int64_t e = (int64_t)d.rows * 4 * units;
    most = e > most ? e : most;
  }
  unsigned gx = blocks_for(most);
  if (gx > 256) gx = 256;
  hipLaunchKernelGGL(unpack_lstm_grads_kernel, dim3(gx, n), dim3(256), 0, (hipStream_t)stream, b, units);
  return (int)hipGetLastError();
}

int orcai_l2_values(const float* base, const int64_t* off_host, const int64_t* n_host, int count, float lambda, double* out, void* stream) {
  if (!base || !off_host || !n_host || !out || count <= 0 || count > 8) return ORCAI_E_BADARG;
  L2Batch b;
  for (int i = 0; i < count; ++i) {
    if (off_host[i] < 0 || n_host[i] <= 0) return ORCAI_E_BADARG;
    b.off[i] = off_host[i];
    b.n[i] = n_host[i];
  }
  hipLaunchKernelGGL(l2_values_kernel, dim3(64, count), dim3(256), 0, (hipStream_t)stream, base, b, lambda, out);
  return (int)hipGetLastError();
}

int orcai_ema_update(float* moving, const float* batch, int n, float momentum, void* stream) {
  if (!moving || !batch || n <= 0) return ORCAI_E_BADARG;
  hipLaunchKernelGGL(ema_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, moving, batch, n, momentum, (const int32_t*)nullptr);
  return (int)hipGetLastError();
}

int orcai_step_ok(const float* g, int64_t ng, const float* stats, int64_t ns, int32_t* ok, int64_t* skipped, void* stream) {
  if (!g || !ok || !skipped || ng <= 0 || ns < 0 || (ns > 0 && !stats)) return ORCAI_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(flag_set_kernel, dim3(1), dim3(64), 0, st, ok);
  unsigned blocks = blocks_for(ng);
  hipLaunchKernelGGL(all_finite_kernel, dim3(blocks > 1024 ? 1024 : blocks), dim3(256), 0, st, g, ng, ok);
  if (ns > 0) hipLaunchKernelGGL(all_finite_kernel, dim3(blocks_for(ns) > 1024 ? 1024 : blocks_for(ns)), dim3(256), 0, st, stats, ns, ok);
  hipLaunchKernelGGL(count_skipped_kernel, dim3(1), dim3(64), 0, st, ok, skipped);
  return (int)hipGetLastError();
}

int orcai_poison_if_nonfinite(const float* stats, int64_t ns, float* g, void* stream) {
  if (!stats || !g || ns <= 0) return ORCAI_E_BADARG;
  const unsigned blocks = blocks_for(ns);
  hipLaunchKernelGGL(poison_if_nonfinite_kernel, dim3(blocks > 1024 ? 1024 : blocks), dim3(256), 0, (hipStream_t)stream, stats, ns, g);
  return (int)hipGetLastError();
}

int orcai_adam_step_guarded(float* w, const float* g, float* m, float* v, int64_t n, const float* lr, float b1, float b2, float eps, const uint64_t* counter, float gscale,
                            const int32_t* ok, void* stream) {
  if (!w || !g || !m || !v || !lr || !counter || !ok || n <= 0) return ORCAI_E_BADARG;
  hipLaunchKernelGGL(adam_dev_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, w, g, m, v, n, lr, b1, b2, eps, counter, gscale, ok);
  return (int)hipGetLastError();
}

int orcai_ema_update_guarded(float* moving, const float* batch, int n, float momentum, const int32_t* ok, void* stream) {
  if (!moving || !batch || !ok || n <= 0) return ORCAI_E_BADARG;
  hipLaunchKernelGGL(ema_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, moving, batch, n, momentum, ok);
  return (int)hipGetLastError();
}

int orcai_counter_advance_guarded(uint64_t* counter, const int32_t* ok, void* stream) {
  if (!counter || !ok) return ORCAI_E_BADARG;
  hipLaunchKernelGGL(counter_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, counter, ok);
  return (int)hipGetLastError();
}

int orcai_lstm_hprev(const float* h, int B, int T, int units, float* hprev, void* stream) {
  if (!h || !hprev || B <= 0 || T <= 0 || units <= 0) return ORCAI_E_BADARG;
  const int64_t n = (int64_t)B * T * 2 * units;
  hipLaunchKernelGGL(lstm_hprev_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, h, B, T, units, hprev);
  return (int)hipGetLastError();
}

}  // extern "C"
