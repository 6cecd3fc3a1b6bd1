/*
 * orcai_hip.h -- C ABI of liborcai_hip.so, the MI355X (gfx950) implementation of
 * orcAI's spectrogram -> label hot path.
 *
 * The reference (ethz-tb/orcAI v1.0.3) is pure Python and has no FFI layer; its
 * hot-path arithmetic is delegated to librosa/numpy/Keras calls.  Each entry
 * point below names the reference call site (file:line under
 * /root/reference/src/orcAI/) whose arithmetic it replaces.  INTEGRATION.md
 * shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - the caller owns and allocates every buffer (no ownership transfer);
 *   - the return value is a hipError_t as int (0 = hipSuccess), or a negative
 *     ORCAI_E_* code for argument errors detected on the host before launch;
 *   - nothing here allocates, frees or synchronises, so every call may be
 *     captured into a hipGraph.  The only exceptions are the *_host readback
 *     helpers, which synchronise the stream and say so, and the setup call
 *     orcai_frontend_workspace_bytes(), which uploads the front end's constant
 *     tables once (synchronously) before any capturable call can be made.
 */
#ifndef ORCAI_HIP_H
#define ORCAI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORCAI_E_BADARG (-1)      /* null pointer, non-positive size, misaligned buffer */
#define ORCAI_E_UNSUPPORTED (-2) /* parameter combination the HIP path does not implement */

/* library identification: "orcai_hip <version> gfx950" */
const char* orcai_version(void);

/* ------------------------------------------------------------------------------------------
 * Front end (spectrogram.py:15-87)
 * ------------------------------------------------------------------------------------------ */

/* Bytes of device workspace the front-end calls need (histograms + selection state).
 * The workspace must be 256-byte aligned.  SETUP CALL: also uploads the Hann window and the FFT twiddle tables to device
 * constants (once per process, synchronous), so that every later front-end call only enqueues work on `stream`. */
size_t orcai_frontend_workspace_bytes(void);

/* Zero the workspace (histograms, running max, selection state).  Must precede each
 * recording's orcai_stft_db / orcai_hist_level1. */
int orcai_frontend_reset(void* workspace, void* stream);

/* Rational polyphase resampler (the resampling half of librosa.load(sr=...), spectrogram.py:23-27; libsoxr itself
 * is absent, so this stage cannot be bit-compared: parity unpinned).
 *   out[n] = sum_j x[floor(n*M/L) - ntaps/2 + 1 + j] * table[(n*M) mod L][j]; table f32[L][ntaps], ntaps % 4 == 0. */
int orcai_resample_polyphase(const float* x, int64_t n_in, float* out, int64_t n_out, int L, int M, const float* table, int ntaps, void* stream);

/* librosa.stft(n_fft=512, hop_length=hop, window="hann", center=True, pad_mode="constant")
 * followed by the first half of amplitude_to_db (spectrogram.py:34-39, :51-53):
 *   out_db[t*k_crop + k] = 10*log10(max(|X[k,t]|^2, 1e-10))          k in [0, k_crop)
 * (the reference level, max-80 floor and normalisation are applied by orcai_clip_normalize).
 * Also accumulates into the workspace: max |X|^2 over ALL 257 bins and all frames, and the
 * level-1 selection histogram of the values written.
 *   pcm        f32[n_samples], mono, already at the target sampling rate
 *   n_frames   must equal 1 + (n_samples - (n_fft & 1)) / hop  (librosa's centred frame count; 1 + n_samples / hop for even n_fft)
 *   k_crop     1..1 + n_fft/2 leading rFFT bins to keep (171 for orcai-V1: first bin with f >= 16 kHz)
 *   out_db     f32[n_frames * k_crop], layout [frame][bin]  (the transposed layout
 *              preprocess_spectrogram returns, spectrogram.py:86)
 * n_fft = 512 (every shipped parameter file) runs the tuned kernel; any other power of two from 32 to 4096 a plain
 * one-workgroup-per-frame radix-2 kernel, every other size from 2 to 4096 (odd ones too) a direct float64 transform, O(n_fft^2)
 * per frame -- both followed by a separate level-1 histogram pass (same outputs); n_fft > 4096 is ORCAI_E_UNSUPPORTED.
 * With n_fft != 512 "ALL 257 bins" reads "all 1 + n_fft/2 bins". */
int orcai_stft_db(const float* pcm, int64_t n_samples, int n_fft, int hop, int64_t n_frames, int k_crop,
                  float* out_db, void* workspace, void* stream);

/* Level-1 selection histogram of an arbitrary f32 array (used when the dB array comes from the
 * caller, i.e. the drop-in preprocess_spectrogram(spectrogram, ...) entry, spectrogram.py:58). */
int orcai_stft_blocks(int blocks); /* experiments: persistent workgroups of the STFT launch (default 1536); <= 0 queries; returns the previous value */
int orcai_stft_occupancy(void);     /* workgroups of the STFT kernel per compute unit as the runtime computes it */
int orcai_hist_level1(const float* x, int64_t n, void* workspace, void* stream);

/* Exact order statistics: the rank_lo-th and rank_hi-th smallest (0-based) of x[0..n), i.e. what
 * np.percentile(x, q, method="nearest") returns for the indices computed on the host
 * (spectrogram.py:70-75).  Requires the level-1 histogram of exactly these n values to be in
 * the workspace.  Results stay in the workspace (see orcai_frontend_stats_host). */
int orcai_quantile_select(const float* x, int64_t n, int64_t rank_lo, int64_t rank_hi, void* workspace, void* stream);

/* Turn the selected raw order statistics into clip bounds.
 *   use_ref = 1: values are un-referenced dB from orcai_stft_db; ref_db = 10*log10(max(pmax,1e-10)),
 *                bound = max(selected - ref_db, -top_db)                    (spectrogram.py:51-53)
 *   use_ref = 0: values already are final dB (drop-in preprocess entry): bound = selected. */
int orcai_frontend_finalize(int use_ref, float top_db, void* workspace, void* stream);

/* In place: v = max(x - ref_db, -top_db) (when use_ref), clip to [p_lo, p_hi], (v - p_lo)/(p_hi - p_lo)
 * (spectrogram.py:78-83).  Bounds are read from the workspace on the device. */
int orcai_clip_normalize(float* x, int64_t n, const void* workspace, void* stream);

/* In place: x = max(x - ref_db, -top_db): the dB array calculate_spectrogram returns
 * (spectrogram.py:51-53). */
int orcai_db_reference(float* x, int64_t n, const void* workspace, void* stream);

/* [F, T] -> [T, f_hi - f_lo] crop + transpose (spectrogram.py:68, :86) for the drop-in
 * preprocess_spectrogram entry whose input is the reference's [freq, time] layout. */
int orcai_crop_transpose(const float* in_ft, int64_t n_freq, int64_t n_frames, int f_lo, int f_hi, float* out_tf, void* stream);

/* Synchronises `stream` and copies {pmax, ref_db, p_lo, p_hi, sel_lo_raw, sel_hi_raw} to the host. */
int orcai_frontend_stats_host(const void* workspace, float stats_host[6], void* stream);

/* One call = reset + stft_db + select + finalize + clip_normalize: the whole of make_spectrogram
 * (spectrogram.py:90-147) after file decode.  rank_lo/rank_hi as for orcai_quantile_select with
 * n = n_frames * k_crop. */
int orcai_make_spectrogram(const float* pcm, int64_t n_samples, int n_fft, int hop, int64_t n_frames, int k_crop,
                           int64_t rank_lo, int64_t rank_hi, float top_db, float* out, void* workspace, void* stream);


/* ------------------------------------------------------------------------------------------
 * Model forward, inference (architectures.py:162-241 as executed by model.predict, predict.py:265-268)
 * Activations are planar fp32 [snippet][channel][H = time][W = freq].  BatchNormalization is folded on
 * the host into per-channel (scale, shift): scale = gamma*rsqrt(var + 1e-3),
 * shift = beta - mean*scale + conv_bias*scale.
 * ------------------------------------------------------------------------------------------ */

/* Padded plane layout used by every convolutional activation tensor: an H x W plane is stored as HP x WP floats,
 * HP = H + 2*(k/2) (k/2 zero rows above and below), WP = orcai_padded_width(W, k) = roundup4(W + k/2) (zero columns
 * on the right); image pixel (y, x) is at (y + k/2)*WP + x.  The CALLER zero-fills the buffers once; the kernels
 * never write the pads, so "same" zero padding costs no bounds checks and every tile halo is one contiguous run. */
int orcai_padded_width(int W, int ksize);

/* Knob of the separable-conv launcher for k = 3 launches with plane or x-pooled output (the inference trunk, the training forward
 * with its depthwise-output store, the input-gradient passes).  1 (default): the LDS-shared-row kernels -- sepconv_tile_kernel (a
 * workgroup of 8 waves owns 8 image rows x 64 columns; the tile's 10 input rows per channel quad are fetched once by LDS-DMA) for
 * planes at least two strips wide with Cout in 17..32 and <= 8 input quads, sepconv_ftile_kernel (8 consecutive windows of the flat
 * plane share one contiguous range of rows) for the rest; 2: sepconv_ftile_kernel for all of them; 0: the one-window-per-wave
 * sepconv_kernel everywhere.  All three perform the same arithmetic in the same order (bit-identical results).  Returns the previous
 * value; values outside [0, 2] only query.  Process-wide, not thread-safe. */
int orcai_sepconv_tile_mode(int mode);

/* Same kind of knob for orcai_conv0_sepconv: windows per wave (>= 1; the next window's inputs are prefetched while the current
 * one is computed).  Returns the previous value; values outside [1, 64] only query. */
int orcai_entry_windows(int windows_per_wave);

/* orcai_conv0_sepconv on planes at least two 62-column strips wide with Cout in 17..32: waves per workgroup of the strip-tile
 * variant (conv0_sep_tile_kernel: every entry-activation row is computed once per tile and shared through LDS), 10 (default) or 16;
 * 0 = conv0_sep_kernel everywhere.  Bit-identical either way.  Returns the previous value; other values only query. */
int orcai_entry_tile(int waves);

/* Conv2D(16, k, padding="same") + BN + ReLU on the 1-channel spectrogram (architectures.py:164-168).
 *   in              f32, UNPADDED: snippet b starts at in + b*snippet_stride and is [H][W] row-major.  For the sliding
 *                   50 % overlap view of a [T][W] spectrogram use snippet_stride = (H/2)*W: no snippet copy is
 *                   materialised (predict.py:253-261 makes one)
 *   w               f32[k*k][16] (Keras kernel (k,k,1,16) flattened); scale/shift f32[16]
 *   out             f32[B][16][HP][WP] padded planes */
int orcai_conv0_bn_relu(const float* in, int64_t snippet_stride, int B, int H, int W, int ksize, const float* w, const float* scale,
                        const float* shift, float* out, void* stream);

/* Entry convolution fused into the first separable convolution, k = 3, inference (architectures.py:164-179):
 *   Conv2D(16, 3, same) + BN + ReLU  ->  ReLU -> SeparableConv2D(Cout, 3, same) -> BN -> [ReLU]
 * = orcai_conv0_bn_relu followed by orcai_sepconv_bn(relu_in = 1, out_layout = 0), bit for bit, without the 16-channel entry
 * activation ever reaching HBM.  in / snippet_stride / w0 / scale0 / shift0 as for orcai_conv0_bn_relu; dw f32[4][9][4] (channel-quad layout, see orcai_sepconv_bn),
 * pw f32[16][Cout], scale / shift f32[Cout] as for orcai_sepconv_bn; out: padded channel-quad planes of Cout channels.
 *   prev_sub (may be NULL)  f32[B][4][ceil(H/2)][ceil(W/2)][4]: the entry activation at pixels (2i, 2j) only -- all that the
 *                           strided 1x1 residual convolution of block 1 reads (orcai_pool_res_add with flag 2). */
int orcai_conv0_sepconv(const float* in, int64_t snippet_stride, int B, int H, int W, const float* w0, const float* scale0, const float* shift0,
                        const float* dw, const float* pw, const float* scale, const float* shift, int Cout, int relu_out, float* out, float* prev_sub,
                        void* stream);

/* [ReLU] -> SeparableConv2D(Cout, k, same) -> BN -> [ReLU]   (architectures.py:174-189, :198-206)
 *   in   f32[B][ceil(Cin/4)][HP][WP][4] padded channel-quad planes
 *   dw   f32[ceil(Cin/4)][k*k][4]   channel-quad layout: element (q, tap, j) = Keras depthwise kernel (k,k,Cin,1) at
 *        [tap / k][tap % k][4q + j][0], zero for channels >= Cin (what the kernels read with scalar loads; 16-byte aligned)
 *   pw   f32[Cin][Cout]  (Keras pointwise kernel (1,1,Cin,Cout))
 *   out_layout 0: padded channel-quad planes;  1: f32[B][H][W*Cout] with feature = x*Cout + c, i.e. Keras
 *   Reshape((-1, W*C)) of the NHWC tensor (architectures.py:208);  2: x-pooled f32[B][CQout][H][roundup4(ceil(W/2))][4]:
 *   element (y, j) = max over the column pair (2j, 2j+1) -- the first half of the MaxPooling2D((3,2), 2, "same") that
 *   follows (architectures.py:190), consumed by orcai_pool_res_add(xpooled = 1).  Cout <= 64, k in {3,5,7}. */
int orcai_sepconv_bn(const float* in, int B, int Cin, int H, int W, int ksize, int relu_in, const float* dw, const float* pw, const float* scale,
                     const float* shift, int Cout, int relu_out, int out_layout, float* out, void* stream);

/* MaxPooling2D((3,2), strides 2, "same")(s) + Conv2D(C, 1, strides 2, "same")(prev)   (architectures.py:190-196)
 *   xpooled is a flag word.  Bit 0: s is the x-pooled tensor written by orcai_sepconv_bn(out_layout = 2) instead of padded
 *   channel-quad planes of C channels.  Bit 1: prev is the compact subsample f32[B][ceil(Cp/4)][ceil(H/2)][ceil(W/2)][4] of
 *   pixels (2i, 2j) written by orcai_conv0_sepconv instead of padded channel-quad planes of Cp channels.  wr f32[Cp][C], br f32[C]
 *   -> out f32[B][C][ceil(H/2) + 2*(k/2)][orcai_padded_width(ceil(W/2), k)] padded planes */
int orcai_pool_res_add(const float* s, const float* prev, int B, int C, int Cp, int H, int W, int ksize, const float* wr, const float* br,
                       float* out, int xpooled, void* stream);
/* A residual block's second separable convolution WITH the block's tail in its epilogue (architectures.py:172-196, predict.py:265-268):
 *   out = MaxPooling2D((3,2), strides 2, "same")(scale * SepConv(relu_in ? relu(in) : in) + shift [relu_out]) + Conv2D(C, 1, strides 2)(prev) + br
 * = orcai_sepconv_bn(out_layout = 2) followed by orcai_pool_res_add(xpooled bit 0), bit for bit, without the x-pooled tensor in HBM.
 *   in f32[B][ceil(Cin/4)][H+2][orcai_padded_width(W,3)][4] planes, dw f32[ceil(Cin/4)][9][4], pw f32[Cin][C], scale / shift f32[C] (folded BatchNorm);
 *   prev: the block input, padded planes of Cp channels at H x W, or (prev_compact) its (2i, 2j) subsample f32[B][ceil(Cp/4)][H/2][ceil(W/2)][4];
 *   wr f32[Cp][C], br f32[C] -> out f32[B][ceil(C/4)][H/2 + 2][orcai_padded_width(ceil(W/2),3)][4] padded planes (pads untouched).
 * ORCAI_E_UNSUPPORTED (the caller runs the two launches) unless ksize == 3, 17 <= C <= 32, 17 <= Cin <= 32, Cp <= 16, H even and the plane is wide
 * enough for 60-column strips (orcai-V1 block 1).  orcai_pool_fused(nt): tiles of 8 conv rows a workgroup marches over (default 8); 0 switches the
 * fused tail off; < 0 queries; returns the previous value. */
int orcai_sepconv_pool_res(const float* in, const float* prev, int B, int Cin, int C, int Cp, int H, int W, int ksize, int relu_in, const float* dw, const float* pw,
                           const float* scale, const float* shift, int relu_out, const float* wr, const float* br, float* out, int prev_compact, void* stream);
int orcai_pool_fused(int nt);
int orcai_pool_vertical(int on); /* experiments: 1 (default) = the inference pooling kernel on stacked tiles (4 output rows x 16 columns per wave: the row two windows share is loaded once) where the pooled plane is >= 40 columns wide; 0 = flat 64-pixel windows everywhere; < 0 queries; returns the previous value.  Bit-identical results. */

/* C[M][N] = act(A[M][K] * Bm[K][N] + bias[N]) [* scale[N] + shift[N]]; act 0 = identity, 1 = ReLU; bias/scale/shift may be NULL.
 * Used for the LSTM input projections x*W + b (architectures.py:210-229) and Dense(128, relu) + BN (:231-237). */
int orcai_gemm_bias_act(const float* A, const float* Bm, const float* bias, const float* scale, const float* shift, float* C, int64_t M, int N,
                        int K, int act, void* stream);

/* Both directions of one Bidirectional(LSTM(units, return_sequences=True)) given xz = x*W + b.
 *   xz  f32[B][T][2][4*units], Uw f32[2][units][4*units], both with the gate columns in the kernel's order:
 *       column 32*w + 16*nt + j  <-  Keras column (2*nt + (j>>3))*units + 8*w + (j&7)     (gate order i,f,c,o)
 *   out f32[B][T][2*units] = concat(forward, backward) at each time step.  units in {64, 128}. */
int orcai_lstm_recurrent(const float* xz, const float* Uw, int B, int T, int units, float* out, void* stream);

/* out[M][N] = sigmoid(x[M][K] * w[K][N] + b[N]),  N <= 8   (architectures.py:239) */
int orcai_dense_sigmoid(const float* x, const float* w, const float* bias, int64_t M, int K, int N, float* out, void* stream);

/* predict.py:276-293: overlay the n snippet predictions [n][P][L] at offsets i*step, count overlaps, divide.
 *   agg f64[S][L], cnt f64[S];  float64 accumulation in snippet order (bit-exact with the numpy loop). */
int orcai_overlap_average(const float* pred, int n, int P, int L, int step, int64_t S, double* agg, double* cnt, void* stream);

/* ------------------------------------------------------------------------------------------
 * Training (train.py:155-219: Adam(lr) + MaskedBinaryCrossentropy + MaskedBinaryAccuracy; architectures.py:210-286)
 * "Row tensors" are f32[M][cols] row-major, M = snippets * time steps.
 * ------------------------------------------------------------------------------------------ */

/* C[M][N] (=|+=) alpha * sum_k A(m,k) B(k,n) + beta_w * Wreg[m][n] with A(m,k) = A[m*sam + k*sak], B(k,n) = B[k*sbk + n*sbn]:
 * weight gradients (A^T B), input gradients (A B^T) and the L2 term 2*lambda*W of kernel_regularizer=l2 (architectures.py:215,225,235). */
int orcai_gemm_strided(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk, int64_t sbn, float* C, int M, int N, int K, float alpha,
                       int accumulate, const float* Wreg, float beta_w, void* stream);

/* out[c] (=|+=) sum_m x[m][c]  (bias gradients) */
int orcai_colsum(const float* x, int M, int C, float* out, int accumulate, void* stream);

/* BatchNormalization in training mode on a row tensor whose channel is (column % C): batch mean / biased variance,
 * y = [relu]((x - mean) * gamma * rsqrt(var + eps) + beta), and its backward (dbeta, dgamma, dx) with the optional ReLU folded in.
 * Up to 2 048 columns / 512 channels the two reductions (orcai_bn_rows_stats, orcai_bn_rows_bwd) read row slabs with coalesced loads and fold the slabs'
 * partial sums in a fixed order through one device array inside the library: bit-reproducible, but at most ONE of these two launchers may be in flight per
 * device (calls on one stream are ordered and fine) -- the restriction orcai_lstm_bwd has. */
int orcai_bn_rows_stats(const float* x, int M, int cols, int C, float* mean, float* var, void* stream);
int orcai_bn_rows_apply(const float* x, int M, int cols, int C, const float* mean, const float* var, const float* gamma, const float* beta, float eps,
                        int relu, float* y, void* stream);
int orcai_bn_rows_bwd(const float* dy, const float* x, int M, int cols, int C, const float* mean, const float* var, const float* gamma, const float* beta,
                      float eps, int relu, float* dbeta, float* dgamma, float* dx, void* stream);

/* Dropout: mask[i] in {0,1} from a counter-based generator (seed, i); y = x * mask * scale (forward and backward). */
int orcai_dropout_mask(float* mask, int64_t n, uint64_t seed, float keep, void* stream);
int orcai_mask_scale(const float* x, const float* mask, float scale, int64_t n, float* y, void* stream);
/* dx = dy * (y > 0) */
int orcai_relu_bwd(const float* dy, const float* y, int64_t n, float* dx, void* stream);

/* MaskedBinaryCrossentropy / MaskedBinaryAccuracy (architectures.py:262-286): acc3 = {sum of BCE, unmasked count, correct count}
 * (f64, device); dz (may be NULL) = d(mean BCE)/d(logit of the final sigmoid). */
int orcai_masked_bce(const float* p, const float* y, int64_t n, float mask_value, double* acc3, float* dz, void* stream);
/* The same with Keras' class_weight semantics (train.py:125-136, 214: model.fit(class_weight=...)): Keras turns class_weight into a
 * sample weight per (snippet, step) -- class_weight[argmax over labels of y_true] -- and multiplies the SCALAR loss of
 * MaskedBinaryCrossentropy by the batch mean of those weights.  loss_weight (may be NULL = 1) points to that one device float:
 * acc3[0] and dz are multiplied by it.  grad_scale > 0 multiplies dz only (static loss scale of the f16 path, undone in
 * orcai_adam_step's gscale). */
int orcai_masked_bce_w(const float* p, const float* y, int64_t n, float mask_value, double* acc3, float* dz, const float* loss_weight, float grad_scale,
                       void* stream);
/* out += lambda * sum w^2   (value of the L2 penalty) */
int orcai_l2_value(const float* w, int64_t n, float lambda, double* out, void* stream);

/* Keras-3 Adam on flat buffers: g is scaled by gscale first (1/world_size after an all-reduce(sum)); step is 1-based. */
int orcai_adam_step(float* w, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps, int step, float gscale, void* stream);

/* Training forward of one Bidirectional(LSTM): as orcai_lstm_recurrent, plus the gate activations (i,f,g,o, permuted columns)
 * f32[B][T][2][4*units] and cell states f32[B][T][2][units] the backward pass needs. */
int orcai_lstm_split(int on); /* 1 (default): the f32 path's LSTM recurrences (inference, training forward and backward) run their recurrent product on f16 MFMA with every operand split as hi + lo / 4096 (f32 accuracy, 24 MFMAs of 16 cycles per step instead of 64 of 32); 0: v_mfma_f32_16x16x4_f32; < 0 queries; returns the previous value */
int orcai_lstm_train_fwd(const float* xz, const float* Uw, int B, int T, int units, float* out, float* gates, float* cstate, void* stream);
/* The f16 path's twin (same arguments, f32 tensors): the recurrent product on v_mfma_f32_16x16x32_f16 with h and U rounded to f16,
 * f32 accumulation on the f32 input projection, gates / cell state / outputs in f32; orcai_h_lstm_bwd likewise for the recurrent
 * term dz U^T of orcai_lstm_bwd (dz rounded to f16 for the product only; dxz is written in f32). */
int orcai_h_lstm_train_fwd(const float* xz, const float* Uw, int B, int T, int units, float* out, float* gates, float* cstate, void* stream);
int orcai_h_lstm_bwd(const float* dH, const float* gates, const float* cstate, const float* Uw, int B, int T, int units, float* dxz, void* stream);
/* Backward through time: dH f32[B][T][2*units] (gradient of the layer output) -> dxz f32[B][T][2][4*units] (permuted columns).
 * With orcai_lstm_split(1) the launch rescales dz by a power of two taken from max|dH| (two tiny launches in front of the recurrence); that scale
 * travels through one device global, so at most ONE orcai_lstm_bwd may be in flight per device (calls on one stream are ordered and fine;
 * concurrent calls on two streams of the same device are not supported).  Device-symbol addresses and the > 64 KiB LDS opt-ins are looked up
 * per device on the device's first call, which must not be inside a stream capture.  dz is saturated at 2^15 x the largest incoming gradient
 * before the f16 split: exploding recurrences are clipped, never turned into inf / NaN. */
int orcai_lstm_bwd(const float* dH, const float* gates, const float* cstate, const float* Uw, int B, int T, int units, float* dxz, void* stream);
/* hprev[b][t][dir][u] = h[b][t-1 (dir 0) | t+1 (dir 1)][dir*units + u], 0 at the sequence start: left operand of dU = hprev^T dxz. */
int orcai_lstm_hprev(const float* h, int B, int T, int units, float* hprev, void* stream);

/* conv0 with a selectable ReLU (training forward stores the pre-BatchNorm output: scale = 1, shift = bias, relu = 0). */
int orcai_conv0_affine(const float* in, int64_t snippet_stride, int B, int H, int W, int ksize, const float* w, const float* scale, const float* shift,
                       int relu, float* out, void* stream);

/* ResNet1DConv head, backward (architectures.py:100-115; Keras differentiates these inside model.fit, train.py:201-219):
 * orcai_freq_mean_bwd: dfeat[m][x*C + c] = dfm[m][c] / W (gradient of ReduceFrequencyMean on the Keras Reshape layout).
 * orcai_conv1d_bwd: x [B][T][C], w [K][C][L], dz [B][T][L] = gradient at the pre-sigmoid output; dW [K][C][L] is ACCUMULATED
 * (zero it first), dx [B][T][C] is written.  The bias gradient is the column sum of dz (orcai_colsum). */
int orcai_freq_mean_bwd(const float* dfm, int64_t M, int W, int C, float* dfeat, void* stream);
int orcai_conv1d_bwd(const float* x, const float* w, const float* dz, int B, int T, int C, int K, int L, float* dW, float* dx, void* stream);

/* Training-data path (replaces DataLoader.__getitem__ io.py:128-147 + reshape_labels io.py:101-126 and the
 * materialised tf.data snapshot io.py:187-218).  store: [total_rows][cols] f32 resident in HBM (recordings concatenated along time);
 * row_starts: device int64[B], first store row of each snippet (caller guarantees start + rows <= total_rows).
 * orcai_gather_snippets: out[b] = rows [start_b, start_b + rows) of the store, as [B][rows][cols].
 * orcai_downsample_labels: out[b][s][l] = round_half_even(mean over the `factor` rows of group s); rows % factor != 0 -> BADARG. */
int orcai_gather_snippets(const float* store, const int64_t* row_starts, int B, int rows, int cols, float* out, void* stream);
int orcai_downsample_labels(const float* labels, const int64_t* row_starts, int B, int rows, int L, int factor, float* out, void* stream);

/* ResNet1DConv head (architectures.py:10-15 ReduceFrequencyMean, :107-115 Conv1D(num_labels, kernel_size = 36, "same", sigmoid)).
 * orcai_freq_mean: feat [M][W*C] in the Keras Reshape layout (feature = x*C + c) -> out [M][C] = mean over x.
 * orcai_conv1d_sigmoid: x [B][T][C], w [K][C][L] (Keras Conv1D kernel layout), bias [L] -> out [B][T][L];
 * "same" padding as TensorFlow: (K-1)/2 zero steps before, K/2 after. */
int orcai_freq_mean(const float* feat, int64_t M, int W, int C, float* out, void* stream);
int orcai_conv1d_sigmoid(const float* x, const float* w, const float* bias, int B, int T, int C, int K, int L, float* out, void* stream);

/* orcai_sepconv_bn with the tap size (ktap in {1,3,5,7}) decoupled from the padding of the planes (ksize_planes >= ktap), used by the
 * backward pass: ktap = 1 is a pure pointwise conv (input gradient through the pointwise weights), out_layout 3 scatter-ADDS the
 * result to pixel (2y, 2x) of planes of an H2 x W2 image (input gradient of the stride-2 1x1 residual conv). */
int orcai_sepconv_planes(const float* in, int B, int Cin, int H, int W, int ksize_planes, int ktap, int relu_in, const float* dw, const float* pw,
                         const float* scale, const float* shift, int Cout, int relu_out, int out_layout, int H2, int W2, float* out, void* stream);
/* The same, additionally storing the depthwise output u (planes of Cin channels, u_out may be NULL): the training forward keeps u
 * because the pointwise weight gradient is sum_pixels u (x) dv (architectures.py:176-189 differentiated by Keras inside model.fit). */
int orcai_sepconv_planes_u(const float* in, int B, int Cin, int H, int W, int ksize_planes, int ktap, int relu_in, const float* dw, const float* pw,
                           const float* scale, const float* shift, int Cout, int relu_out, int out_layout, int H2, int W2, float* out, float* u_out,
                           void* stream);

/* BatchNormalization (training) on padded channel-quad planes: batch mean / biased variance (scratch: f64[8*ceil(C/4)*32] for
 * orcai_bn_planes_stats -- 32 accumulator copies for its small-plane pass --, f64[8*ceil(C/4)] for the other entry points),
 * y = [relu](v*s + t) at interior pixels, backward (dbeta, dgamma, dv) with the optional ReLU folded in. */
int orcai_bn_planes_stats(const float* v, int B, int C, int H, int W, int ksize, double* scratch, float* mean, float* var, void* stream);

/* Training forward of a k = 3 separable convolution with the BatchNorm batch statistics of its output reduced in the kernel's epilogue
 * (train.py:155-219 runs keras' SeparableConv2D -> BatchNormalization(training=True)): = orcai_sepconv_planes_u(ksize_planes = ktap = 3,
 * relu_out = 0, out_layout = 0) followed by the sums of orcai_bn_planes_stats, without the read pass over `out`.
 *   shards   f64[32][ceil(Cout/4)][8] accumulator copies (zeroed here); orcai_bn_finish_sharded turns them into mean / biased variance
 * Only the shapes the LDS-tile kernels take (every plane up to ~500 pixels wide with orcai_sepconv_tile_mode != 0): ORCAI_E_UNSUPPORTED
 * otherwise, with nothing touched, and the caller runs the two calls above.  Per-workgroup partial sums are f32 (512 pixels), accumulated
 * in f64. */
int orcai_sepconv_planes_stats(const float* in, int B, int Cin, int H, int W, int relu_in, const float* dw, const float* pw, const float* scale, const float* shift,
                               int Cout, float* out, float* u_out, double* shards, void* stream);
/* orcai_sepconv_planes_stats whose input planes hold the PRE-normalisation tensor v of the BatchNorm (+ ReLU) in front of the conv
 * (architectures.py:176-183: bn_a + ReLU feed the second separable conv of a block): y = max(fma(v, gamma * inv, beta - mean * gamma * inv), 0)
 * is formed on load -- the arithmetic of orcai_bn_planes_apply, bit for bit -- and zero outside the image, so the normalised tensor is never
 * written or re-read.  Same refusal rule (ORCAI_E_UNSUPPORTED before anything is touched). */
int orcai_sepconv_planes_stats_bn(const float* v_in, int B, int Cin, int H, int W, const float* in_mean, const float* in_var, const float* in_gamma, const float* in_beta,
                                  float in_eps, const float* dw, const float* pw, const float* scale, const float* shift, int Cout, float* out, float* u_out, double* shards,
                                  void* stream);
/* The input-gradient pass of a k = 3 separable conv (flipped depthwise taps, identity pointwise factor, plane output) with an epilogue that
 * reads a reference tensor `ref` of the OUTPUT's layout where it stores (train.py:201-219: inside Keras' backward):
 *   epi 2: the output is the gradient dy of a BatchNorm whose pre-normalisation input is ref: the BatchNorm backward sums
 *          dbeta = sum g, dgamma = sum g * xhat (g = relu ? dy * [gamma * xhat + beta > 0] : dy) are reduced where dy is stored and left in
 *          `shards` as scratch2C = dbeta[4 CQo] | dgamma[4 CQo] doubles (shards: 32 * 8 * CQo doubles of workspace): the next
 *          orcai_bn_bwd_pointwise(_wgrad) call takes sums_ready = 1 and the read pass over (dy, v) is gone;
 *   epi 3: out = ref > 0 ? out : 0 (the ReLU in front of the conv, folded: no separate orcai_planes_relu_bwd pass).
 * ORCAI_E_UNSUPPORTED (before anything is touched) for shapes the LDS-tile kernels do not take: the caller runs the separate passes. */
int orcai_sepconv_planes_epi(const float* in, int B, int Cin, int H, int W, const float* dw, const float* pw, const float* scale, const float* shift, int Cout, float* out,
                             int epi, const float* ref, const float* mean, const float* var, const float* gamma, const float* beta, float eps, int relu, double* shards,
                             void* stream);
int orcai_bn_finish_sharded(const double* shards, int B, int C, int H, int W, float* mean, float* var, void* stream);
/* The f16 twins (octet planes; shards f64[32][ceil(Cout/8)][16]; flat-tile kernel, any plane width and channel count it handles; the sums are
 * taken on the f32 values before their rounding to f16). */
int orcai_h_sepconv_stats(const void* in, int B, int Cin, int H, int W, int relu_in, const void* dw, const void* pwf, const float* scale, const float* shift, int Cout,
                          void* out, void* u_out, double* shards, void* stream);
/* orcai_h_sepconv_stats whose input planes hold the PRE-normalisation tensor v of a BatchNorm + ReLU (the f16 twin of orcai_sepconv_planes_stats_bn):
 * y = f16(relu(fma(v, gamma * rsqrt(var + eps), beta - mean * ...))) is formed on load -- the value orcai_h_bn_planes_apply stores, bit for bit -- and zero
 * outside the image, so the training forward never materialises y_a (architectures.py:172-189 in training mode). */
int orcai_h_sepconv_stats_bn(const void* v_in, int B, int Cin, int H, int W, const float* in_mean, const float* in_var, const float* in_gamma, const float* in_beta,
                             float in_eps, const void* dw, const void* pwf, const float* scale, const float* shift, int Cout, void* out, void* u_out, double* shards,
                             void* stream);
int orcai_h_bn_finish_sharded(const double* shards, int B, int C, int H, int W, float* mean, float* var, void* stream);
int orcai_bn_planes_apply(const float* v, int B, int C, int H, int W, int ksize, const float* mean, const float* var, const float* gamma, const float* beta,
                          float eps, int relu, float* y, void* stream);
int orcai_bn_planes_bwd(const float* dy, const float* v, int B, int C, int H, int W, int ksize, const float* mean, const float* var, const float* gamma,
                        const float* beta, float eps, int relu, double* scratch, float* dbeta, float* dgamma, float* dv, void* stream);
/* Kernel-layout copies of trunk weights from the flat parameter buffer in one launch: desc = n_desc x {type, src offset, dst offset,
 * C, aux} (device int32): type 0 depthwise (k,k,C,1) -> [ceil(C/4)][k*k][4] (aux = k*k), type 1 the same with reversed taps,
 * type 2 pointwise (1,1,C,aux) -> transposed [aux][C]. */
int orcai_pack_weights(const float* w, const int* desc, int n_desc, float* out, void* stream);
/* Backward of the entry block Conv2D(16) -> BatchNormalization -> ReLU (architectures.py:162-168): dbeta / dgamma of bn0 and the
 * conv weight gradient dW0[tap][16] (accumulated) from dy = gradient at the ReLU output and v = pre-BN conv output.  The BN input
 * gradient is formed on the fly and never written (the entry conv has no input gradient). */
int orcai_conv0_bn_bwd(const float* in, int64_t snippet_stride, const float* dy, const float* v, int B, int H, int W, int ksize, const float* mean,
                       const float* var, const float* gamma, const float* beta, float eps, double* scratch, float* dbeta, float* dgamma, float* dW,
                       void* stream);
/* The entry conv + bn0 of the training forward WITHOUT the pre-normalisation tensor v0 in HBM (architectures.py:164-168; train.py:201-219):
 *   orcai_conv0_stats     first pass: v0 = fma(conv, scale, shift) is formed as orcai_conv0_affine forms it, only its batch statistics leave
 *                         the kernel (shards: 32 * 4 * 8 doubles, the layout orcai_bn_finish_sharded(C = 16) reads);
 *   orcai_conv0_affine_bn second pass: the conv recomputed from the 1-channel input, then y0 = [relu](fma(v0, gamma * inv, beta - mean * gamma * inv))
 *                         -- the two roundings of orcai_conv0_affine + orcai_bn_planes_apply, so y0 is bit for bit what those launches wrote;
 *   orcai_conv0_bn_bwd_x  orcai_conv0_bn_bwd with v0 rebuilt per pixel from the input taps (w0 [k*k][16], bias [16]) instead of read; dW accumulates. */
int orcai_conv0_stats(const float* in, int64_t snippet_stride, int B, int H, int W, int ksize, const float* w, const float* scale, const float* shift, double* shards,
                      void* stream);
int orcai_conv0_affine_bn(const float* in, int64_t snippet_stride, int B, int H, int W, int ksize, const float* w, const float* scale, const float* shift,
                          const float* bn_mean, const float* bn_var, const float* bn_gamma, const float* bn_beta, float bn_eps, int relu, float* out, void* stream);
int orcai_conv0_bn_bwd_x(const float* in, int64_t snippet_stride, const float* dy, int B, int H, int W, int ksize, const float* w0, const float* bias, const float* mean,
                         const float* var, const float* gamma, const float* beta, float eps, double* scratch2C /*>= 1024 doubles*/, float* dbeta, float* dgamma, float* dW,
                         float* workspace /*per-workgroup partial weight gradients*/, int64_t workspace_floats, void* stream);
/* Training forward / backward of the pooling with the BatchNormalization in front of it applied on the fly (architectures.py:
 * 189-196): `s` / `ybn` hold the PRE-BN tensor v; BN(v) = fma(v, gamma*rsqrt(var+eps), beta - mean*that) is monotone per channel,
 * so the maximum (and its position) is taken on v and transformed once.  BN(v) is never materialised. */
int orcai_pool_res_add_bn(const float* s, const float* prev, int B, int C, int Cp, int H, int W, int ksize, const float* wr, const float* br, float* out,
                          int xpooled, const float* bn_mean, const float* bn_var, const float* bn_gamma, const float* bn_beta, float bn_eps, void* stream);
/* bn_sums != NULL (device f64[8*ceil(C/4)], zeroed by the call): additionally accumulates sum dy and sum dy*xhat per channel, the two
 * reductions of that BatchNorm's backward, so orcai_bn_bwd_pointwise(sums_ready = 1) need not read dy and v for them. */
int orcai_pool_bwd_bn(const float* dout, const float* v, int B, int C, int H, int W, int ksize, float* dy, const float* bn_gamma, const float* bn_mean,
                      const float* bn_var, float bn_eps, double* bn_sums, void* stream);
/* orcai_pool_bwd_bn that also reduces dbias[c] = sum over snippets and pixels of dout[c] -- the bias gradient of the block's residual 1x1
 * convolution, whose output gradient dout is (architectures.py:190-196) -- where the pooling backward reads every dout value exactly once;
 * dout_sums: 4 * ceil(C/4) doubles of workspace.  Replaces an orcai_planes_sum pass over dout. */
int orcai_pool_bwd_bn_bias(const float* dout, const float* ybn, int B, int C, int H, int W, int ksize, float* dy, const float* bn_gamma, const float* bn_mean,
                           const float* bn_var, float bn_eps, double* bn_sums, double* dout_sums, float* dbias, void* stream);
/* orcai_bn_planes_bwd fused with the input gradient through the pointwise weights of the separable conv that produced v:
 * dbeta / dgamma as above, dv (may alias dy) = BN input gradient, du = Wpw dv with wt = pointwise^T [C][Cin] (planes of Cin
 * channels).  One pass over dy and v instead of bn apply + a pointwise conv pass that re-reads dv. */
int orcai_bn_bwd_pointwise(const float* dy, const float* v, int B, int C, int H, int W, int ksize, const float* mean, const float* var, const float* gamma,
                           const float* beta, float eps, int relu, double* scratch, int sums_ready, float* dbeta, float* dgamma, const float* wt, int Cin,
                           float* dv, float* du, void* stream);
/* orcai_bn_bwd_pointwise + the pointwise weight gradient of the same separable conv in ONE pass (train.py:201-219 computes these inside Keras):
 * dv is formed per pixel, multiplied by the transposed pointwise weights (du) AND contracted with the depthwise output u of the forward
 * pass (dWpw[ci][co] += sum_pixels u[ci] dv[co]); dv itself is never written.  workspace: per-wave partial products
 * (>= 4 * Cin * C floats per workgroup).  ORCAI_E_UNSUPPORTED (before anything is touched) when ceil(Cin/16) + ceil(C/16) > 6: the caller
 * then runs orcai_bn_bwd_pointwise and orcai_outer_reduce. */
int orcai_pw_wgrad_tiles(int tiles); /* experiments: widest layer (in 16-channel tiles, both operands) the fused entry accepts, 0 = never; < 0 queries; returns the previous value */
int orcai_bn_bwd_pointwise_wgrad(const float* dy, const float* v, const float* u, int B, int C, int H, int W, int ksize, const float* mean, const float* var,
                                 const float* gamma, const float* beta, float eps, int relu, double* scratch2C, int sums_ready, float* dbeta, float* dgamma,
                                 const float* wt, int Cin, float* du, float* dWpw, float* workspace, int64_t workspace_floats, void* stream);
/* out[c] (=|+=) sum over snippets and pixels of x[c] (bias gradients); scratch: f64[4*ceil(C/4)] */
int orcai_planes_sum(const float* x, int B, int C, int H, int W, int ksize, double* scratch, float* out, int accumulate, void* stream);
/* gradient of MaxPooling2D((3,2), 2, "same"): dy[y][x] = sum of dout over the windows whose maximum is ybn[y][x] */
int orcai_pool_bwd(const float* dout, const float* ybn, int B, int C, int H, int W, int ksize, float* dy, void* stream);
/* D[ca][cb] += sum over snippets and pixels of A[ca][p] * Bq[cb][p] (pointwise / residual weight gradients); with a_stride2 the A planes
 * are an Ha x Wa image sampled at (2i, 2j) for pixel (i, j) of the H x W image of Bq.  workspace: device scratch for the per-workgroup
 * partial products (up to 512 * Ca * Cb floats are used; fewer workgroups run if it is smaller). */
int orcai_outer_reduce(const float* A, int Ca, const float* Bq, int Cb, int B, int H, int W, int ksize, int a_stride2, int Ha, int Wa, float* D,
                       float* workspace, int64_t workspace_floats, void* stream);
/* Pixels per pass of orcai_outer_reduce's LDS image: 0 (default) = 256 up to 32 channels per operand, 128 beyond (the LDS per workgroup
 * decides this kernel's occupancy); 128 / 256 force one.  Same sums in a different order.  Returns the previous value. */
int orcai_outer_reduce_pixels(int pixels);
/* dW[tap][c] += sum r[c][p + off(tap)] * du[c][p], r = relu_in ? relu(x) : x  (depthwise weight gradient, written in the Keras
 * kernel layout (k, k, C, 1), i.e. straight into the flat gradient buffer) */
int orcai_dw_wgrad(const float* x, const float* du, int B, int C, int H, int W, int ksize_planes, int ktap, int relu_in, float* dW, void* stream);
int orcai_dw_wgrad_march(int on); /* experiments: 1 (default) = k = 3 launches on planes >= 100 pixels wide run the row-marching kernel (every byte requested once); 0 = the flat-window kernel everywhere; < 0 queries; returns the previous value */
/* the same (k = 3) with r = the BatchNorm + ReLU of the pre-normalisation tensor v, formed on load (see orcai_sepconv_planes_stats_bn) */
int orcai_dw_wgrad_bn(const float* v, const float* du, int B, int C, int H, int W, const float* in_mean, const float* in_var, const float* in_gamma, const float* in_beta,
                      float in_eps, float* dW, void* stream);
/* Depthwise backward of a k = 3 separable conv in ONE pass over (du, x) (replaces orcai_sepconv_planes_epi + orcai_dw_wgrad[_bn]; Keras: the
 * DepthwiseConv2D half of SeparableConv2D's gradient, reference architectures.py:66-86 trained by train.py:201-219):
 *   dr = du (*) reversed depthwise taps (dw_rev: [ceil(C/4)][9][4], tap t = forward tap 8 - t)    -- the gradient w.r.t. the conv's input
 *   dW[tap][c] += sum_p r[c][p + off(tap)] * du[c][p]                                              -- as orcai_dw_wgrad (Keras layout (3, 3, C, 1))
 * with r = relu_in ? relu(x) : x, or, when bn_mean != NULL, r = relu(BatchNorm(x)) formed on load (x = pre-normalisation tensor, as
 * orcai_dw_wgrad_bn).  epi 0: nothing else.  epi 2 (needs the BatchNorm arguments): dr is the gradient of that BatchNorm's output; its backward
 * sums sum g | sum g * xhat (g = dr gated by the ReLU when bn_relu) are left in `shards` as dbeta[4 CQ] | dgamma[4 CQ] doubles (>= 8 * CQ * 32
 * doubles of scratch) for orcai_bn_bwd_pointwise[_wgrad](sums_ready = 1).  epi 3 (relu_in = 1, no BatchNorm): dr is masked by x > 0.  Only the
 * interior of dr is written.  ORCAI_E_UNSUPPORTED (nothing touched): C > 64, misaligned planes. */
int orcai_dw_bwd_fused(const float* x, const float* du, int B, int C, int H, int W, int relu_in, const float* dw_rev, float* dr, float* dW, int epi, const float* bn_mean,
                       const float* bn_var, const float* bn_gamma, const float* bn_beta, float bn_eps, int bn_relu, double* shards, void* stream);
/* orcai_dw_bwd_fused(epi 2) for block 1's first separable conv when the training forward did not keep the entry conv's pre-normalisation tensor
 * (orcai_conv0_stats + orcai_conv0_affine_bn): the conv input y0 = relu(bn0(conv0(snippet))) is rebuilt per pixel from the snippet's nine taps
 * (w0 [9][16], bias0 [16], bn0's batch statistics) instead of read, and the sums left in `shards` (dbeta[16] | dgamma[16] doubles; >= 32 * 32 doubles
 * of scratch) are bn0's backward sums over dr -- what the first pass of orcai_conv0_bn_bwd_x computes; orcai_conv0_bn_bwd_x_ready is that entry
 * point without its first pass.  resq (optional): the residual branch's gradient w.r.t. y0 at the even pixels (2i, 2j) as planes of 16 channels at
 * the pooled resolution [B][4][ceil(H/2) + 2][padded_width(ceil(W/2))][4] (a plain pointwise pass W_res^T dout): added to dr -- and so part of bn0's
 * sums -- instead of a scatter-add pass over dr afterwards.  k = 3 planes. */
int orcai_dw_bwd_fused_conv0(const float* in, int64_t snippet_stride, const float* du, int B, int H, int W, const float* w0, const float* bias0, const float* dw_rev, float* dr,
                             float* dW, const float* bn_mean, const float* bn_var, const float* bn_gamma, const float* bn_beta, float bn_eps, double* shards, const float* resq,
                             void* stream);
int orcai_conv0_bn_bwd_x_ready(const float* in, int64_t snippet_stride, const float* dy, int B, int H, int W, int ksize, const float* w0, const float* bias, const float* mean,
                               const float* var, const float* gamma, const float* beta, float eps, double* scratch2C, float* dbeta, float* dgamma, float* dW, float* workspace,
                               int64_t workspace_floats, void* stream);
/* The entry conv's two reduction passes on a marching kernel (k = 3; csrc/train_trunk.hip conv0_march_kernel): orcai_conv0_stats_march is
 * orcai_conv0_stats (shards: [32][4][8] sums | sums of squares for orcai_bn_finish_sharded) without tiles, LDS or barriers; orcai_conv0_march(1)
 * (default) lets orcai_conv0_bn_bwd_x_ready run its weight-gradient pass the same way (0: the tile kernel; < 0 queries; returns the previous value). */
int orcai_conv0_stats_march(const float* in, int64_t snippet_stride, int B, int H, int W, const float* w, const float* scale, const float* shift, double* shards, void* stream);
int orcai_conv0_march(int on);
/* dW0[tap][c] += sum in[p + off(tap)] * dv[c][p]  (entry conv weight gradient; `in` is the unpadded snippet view) */
int orcai_conv0_wgrad(const float* in, int64_t snippet_stride, const float* dv, int B, int H, int W, int ksize, float* dW, void* stream);
/* Keras-Reshape layout f32[B][H][W*C] -> padded channel-quad planes (gradient entering the final separable conv) */
int orcai_feat_to_planes(const float* f, int B, int C, int H, int W, int ksize, float* out, void* stream);
/* dx = (y > 0) ? dy : 0 on whole plane buffers (n_floats % 4 == 0) */
int orcai_planes_relu_bwd(const float* dy, const float* y, int64_t n_floats, float* dx, void* stream);

/* Step state in DEVICE memory, so that a whole training step can be captured into a hipGraph and replayed (kernel arguments are baked
 * into a captured graph; what changes from step to step must be read from memory):
 *   counter          uint64[1], the number of optimisation steps applied so far; orcai_counter_advance adds 1 at the end of a step;
 *   orcai_dropout_mask_dev   as orcai_dropout_mask with seed = seed_add + counter[0] * 0xD1B54A32D192ED03;
 *   orcai_adam_step_dev      as orcai_adam_step with step = counter[0] + 1 and the learning rate lr[0] (callbacks change it between steps). */
int orcai_dropout_mask_dev(float* mask, int64_t n, const uint64_t* counter, uint64_t seed_add, float keep, void* stream);
int orcai_adam_step_dev(float* w, const float* g, float* m, float* v, int64_t n, const float* lr, float b1, float b2, float eps, const uint64_t* counter, float gscale,
                        void* stream);
int orcai_counter_advance(uint64_t* counter, void* stream);
/* One clear per training step for every reduction scratch buffer: orcai_scratch_arena zeroes `bytes` (rounded down to 32-KiB slots, at most 1024) at `base`
 * with ONE launch on `stream` and registers the range; until the next call, a launcher whose accumulator argument (the `scratch*` / `shards` pointers of the
 * BatchNorm, pooling and weight-gradient entry points) lies inside a slot that no launcher has taken since skips its own zero-fill launch.  Any other pointer,
 * a slot handed to a second launcher, or no registered arena: the launcher clears its accumulator itself, as before.  The caller hands every accumulating
 * launcher of a step its own slot (orcai_amd/training.py: TrunkTrainer._fresh); base = NULL unregisters.  Host-side state: one stream, one step at a time.
 * orcai_arena_take is the query the launchers use (1 = skip the fill; marks the slot taken). */
int orcai_scratch_arena(void* base, size_t bytes, void* stream);

/* Measurement hook (bench.py): the next call of orcai_bn_bwd_pointwise_wgrad / orcai_h_bn_bwd_pointwise_wgrad records the two HIP events (hipEvent_t handles)
 * on its stream around its MAIN kernel only -- the launcher enqueues two small kernels after it (partial-sum fold, f64 -> f32 gradients), so an event pair
 * around the whole call reads ~35 us above the kernel duration rocprofv3 --kernel-trace lists.  The registration is consumed by that call; NULLs clear it.
 * Host-side state, one thread. */
int orcai_profile_bracket(void* ev_start, void* ev_stop);
/* the events for it, for callers without a HIP binding of their own: timing enabled; elapsed needs both events completed */
int orcai_event_create(void** ev);
int orcai_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms);
int orcai_event_destroy(void* ev);
int orcai_arena_take(const void* p, size_t bytes);

/* A voided step (the f16 path under its static loss scale: Keras' LossScaleOptimizer skips the update when a gradient is not finite).
 *   orcai_step_ok: ok[0] = 1 if every value of g[0, ng) and stats[0, ns) (this step's BatchNorm batch statistics; ns may be 0) is
 *   finite, else 0, and skipped[0] += 1; nothing returns to the host, so the decision stays inside a captured graph.
 *   The *_guarded twins do nothing when ok[0] == 0: weights, Adam moments, moving statistics and the step counter keep their values. */
int orcai_step_ok(const float* g, int64_t ng, const float* stats, int64_t ns, int32_t* ok, int64_t* skipped, void* stream);
/* Data parallel replicas must reach the SAME verdict (the reference's MirroredStrategy applies or skips an update on all replicas together,
 * hpsearch.py:186-205): before the gradient all-reduce, g[0] = NaN when any of this rank's stats[0, ns) is not finite -- the summed bucket is then
 * non-finite on every rank and each rank's orcai_step_ok voids the step.  Nothing is written when all statistics are finite. */
int orcai_poison_if_nonfinite(const float* stats, int64_t ns, float* g, void* stream);
int orcai_adam_step_guarded(float* w, const float* g, float* m, float* v, int64_t n, const float* lr, float b1, float b2, float eps, const uint64_t* counter,
                            float gscale, const int32_t* ok, void* stream);
int orcai_ema_update_guarded(float* moving, const float* batch, int n, float momentum, const int32_t* ok, void* stream);
int orcai_counter_advance_guarded(uint64_t* counter, const int32_t* ok, void* stream);

/* Keras LSTM variables <-> the gate-column order of the recurrence kernels (orcai_lstm_recurrent), on the device, once per training
 * step (replaces the per-step framework index / cat / stack kernels; architectures.py:210-229 define the variables).
 * orcai_pack_lstm: desc int32[n_desc][7] = {src offset in w (floats), dst offset (elements), rows, units, ld_dst, col_off, mode};
 *   mode 0: out32[dst + r*ld_dst + col_off + p] = w[src + r*4u + perm(p)];  mode 1: out16[dst + (col_off + p)*ld_dst + r] = (f16) the same
 *   (transposed, zero-padded copy for orcai_h_gemm_bias_act).  perm(p), p = 32w + 16nt + j: Keras column (2nt + (j>>3))*u + 8w + (j&7).
 * orcai_unpack_lstm_grad: G[r*4u + perm(p)] = src[r*ld_src + col_off + p] + l2g * W[r*4u + perm(p)]   (W may be NULL).
 * orcai_ema_update: moving = moving*momentum + batch*(1 - momentum)  (BatchNormalization moving statistics, one flat buffer). */
int orcai_pack_lstm(const float* w, const int* desc, int n_desc, float* out32, void* out16, void* stream);
int orcai_unpack_lstm_grad(const float* src, int ld_src, int col_off, int rows, int units, float* G, const float* W, float l2g, void* stream);
/* The same for up to 16 (source, destination) pairs in ONE launch -- the 12 of both BiLSTM layers of a training step -- and the value of the L2 penalty over up
 * to 8 slices of the flat weight buffer in one launch (train.py:201-219: kernel_regularizer=l2(0.001) on the LSTM input kernels and Dense-128).  `descs_host`,
 * `off_host`, `n_host` are HOST arrays, read during the call (the descriptors travel by value in the kernel arguments: nothing to keep alive, capturable). */
typedef struct {
  const float* src; /* kernel-order gradient [rows][ld_src] */
  int ld_src, col_off, rows;
  float* G;         /* Keras-layout destination [rows][4 * units] */
  const float* W;   /* Keras-layout weights for the L2 term, or NULL */
  float l2g;
} orcai_unpack_desc;
int orcai_unpack_lstm_grads(const orcai_unpack_desc* descs_host, int n, int units, void* stream);
int orcai_l2_values(const float* base, const int64_t* off_host, const int64_t* n_host, int count, float lambda, double* out, void* stream);
int orcai_ema_update(float* moving, const float* batch, int n, float momentum, void* stream);

/* ------------------------------------------------------------------------------------------
 * f16 path (BASELINE configs[4]: the hyper-parameter sweep's width variants on f16 MFMA; hpsearch.py:186-205 x
 * defaults/default_hps_parameter.json:2-25).  Activations are f16 channel-OCTET planes
 *   [snippet][CO = ceil(C/8)][H + 2R][orcai_padded_width(W, k)][8]   (zero pads, written once by the host),
 * contractions run on v_mfma_f32_16x16x32_f16 with f32 accumulation; weights are f16 COPIES of the f32 master weights in these
 * layouts (orcai_amd/half.py packs them on the host, orcai_h_pack_weights on the device once per training step):
 *   depthwise taps      f16[CO][k*k][8]                      element (o, tap, e) = Keras depthwise kernel [tap/k][tap%k][8o+e][0]
 *   A fragments of W    f16[KG = ceil(Cin/32)][MT = ceil(Cout/16)][64][8]
 *                       element (kg, m, lane, e) = W[cin = 32kg + 8(lane>>4) + e][cout = 16m + (lane&15)], 0 outside W[Cin][Cout]
 *   transposed dense    f16 Wt[N][roundup32(K)] of W[K][N]
 * Pointers to f16 data are void*.  Same conventions as above (device pointers, caller-owned, stream-ordered, capturable).
 * ------------------------------------------------------------------------------------------ */

/* Conv2D(16, k, same)(f32 snippet) * scale + shift [ReLU] -> f16 octet planes of 16 channels (architectures.py:164-168);
 * arguments as orcai_conv0_affine. */
int orcai_h_conv0_affine(const float* in, int64_t snippet_stride, int B, int H, int W, int ksize, const float* w, const float* scale, const float* shift, int relu,
                         void* out, void* stream);
/* The same with the entry BatchNorm + ReLU in the same pass (training forward): v_out = the pre-normalisation conv (as orcai_h_conv0_affine with relu = 0),
 * y_out = relu(BatchNorm(v_out)) formed from the f16 value just stored -- bit-identical to orcai_h_conv0_affine + orcai_h_bn_planes_apply(relu = 1) -- with the
 * batch statistics the caller took from the snippet (orcai_conv0_stats_march + orcai_bn_finish_sharded; architectures.py:164-168). */
int orcai_h_conv0_affine_bn(const float* in, int64_t snippet_stride, int B, int H, int W, int ksize, const float* w, const float* scale, const float* shift,
                            const float* bn_mean, const float* bn_var, const float* bn_gamma, const float* bn_beta, float bn_eps, void* v_out, void* y_out, void* stream);

/* [ReLU] -> depthwise ktap x ktap -> pointwise -> * scale + shift -> [ReLU] on f16 octet planes padded for ksize_planes
 * (architectures.py:174-189, 198-206): orcai_sepconv_planes_u of the f32 path.  out_layout 0: f16 octet planes; 1: f32
 * [B][H][W*Cout] Keras Reshape layout; 2: x-pooled f16 [B][CO][H][roundup4(ceil(W/2))][8]; 3: scatter-add into pixel (2y, 2x) of
 * f16 planes of an (H2, W2) image.  u_out (may be NULL): the depthwise output, f16 octet planes of Cin channels.  Cin, Cout <= 64. */
int orcai_h_sepconv(const void* in, int B, int Cin, int H, int W, int ksize_planes, int ktap, int relu_in, const void* dw, const void* pwf, const float* scale,
                    const float* shift, int Cout, int relu_out, int out_layout, int H2, int W2, void* out, void* u_out, void* stream);

/* MaxPooling2D((3,2), 2, "same")(s) + Conv2D(C, 1, strides 2)(prev) + bias (architectures.py:190-196) on f16 octet planes.
 * xpooled 1: s is the x-pooled tensor of orcai_h_sepconv(out_layout 2); 0: s are planes.  bn_mean..bn_beta (NULL, or all four with
 * xpooled 0): s is the pre-BatchNorm tensor and the pooling runs on BN(s) without materialising it (training forward). */
int orcai_h_pool_res_add(const void* s, const void* prev, int B, int C, int Cp, int H, int W, int ksize, const void* wrf, const float* br, void* out, int xpooled,
                         const float* bn_mean, const float* bn_var, const float* bn_gamma, const float* bn_beta, float bn_eps, void* stream);

/* C[M][N] = act(A[M][K] Wt^T + bias) [* scale + shift]: A f32 (converted to f16 on load), Wt f16[N][roundup32(K)], f32 accumulate and
 * output; act 1 = ReLU.  LSTM input projections and Dense-128 (architectures.py:210-237).  K % 4 == 0. */
int orcai_h_gemm_bias_act(const float* A, const void* Wt, const float* bias, const float* scale, const float* shift, float* C, int64_t M, int N, int K, int act,
                          void* stream);

/* f16 path, training: the twins of the f32 training-trunk entry points on f16 octet planes.  Argument lists are IDENTICAL to
 * orcai_bn_planes_stats, orcai_planes_sum, orcai_bn_planes_apply, orcai_bn_bwd_pointwise, orcai_pool_bwd_bn, orcai_outer_reduce,
 * orcai_dw_wgrad, orcai_conv0_bn_bwd, orcai_pack_weights, orcai_feat_to_planes, orcai_planes_relu_bwd (see those for the arithmetic and
 * the reference lines); plane tensors are f16 octet planes, statistics / weight gradients / scratch stay f32 / f64, and matrix operands
 * are f16 A fragments: wtf of orcai_h_bn_bwd_pointwise = fragments of the TRANSPOSED pointwise matrix (row = conv-input channel,
 * k = conv-output channel).  Gradient planes carry the caller's static loss scale; nothing here knows its value.
 * scratch2C: f64[16 * ceil(C/8)] (orcai_h_bn_planes_stats: x 32 accumulator copies).  orcai_h_planes_relu_bwd counts f16 elements (a multiple of 8).
 * orcai_h_pack_weights descriptors {type, src offset (floats), dst offset (halves), C, aux}: 0 / 1 depthwise octets forward / reversed
 * (aux = k*k), 2 A fragments of W[C][aux], 3 A fragments of its transpose, 4 identity fragments of C channels, 5 all-ones taps. */
int orcai_h_bn_planes_stats(const void* v, int B, int C, int H, int W, int ksize, double* scratch2C, float* mean, float* var, void* stream);
int orcai_h_planes_sum(const void* x, int B, int C, int H, int W, int ksize, double* scratchC, float* out, int accumulate, void* stream);
int orcai_h_bn_planes_apply(const void* v, int B, int C, int H, int W, int ksize, const float* mean, const float* var, const float* gamma, const float* beta,
                            float eps, int relu, void* y, void* stream);
int orcai_h_bn_bwd_pointwise(const void* dy, const void* v, int B, int C, int H, int W, int ksize, const float* mean, const float* var, const float* gamma,
                             const float* beta, float eps, int relu, double* scratch2C, int sums_ready, float* dbeta, float* dgamma, const void* wtf, int Cin,
                             void* dv, void* du, void* stream);
/* orcai_h_bn_bwd_pointwise + orcai_h_outer_reduce(u, dv) in one pass (f16 twin of orcai_bn_bwd_pointwise_wgrad; train.py:201-219): dv is formed per pixel,
 * rounded to f16 as the two-launch path stores it, used for du = Wpw dv AND for the pointwise weight gradient dWpw[Cin][C] += sum_pixels u (x) dv, and never
 * written.  u: the conv's stored depthwise output (octet planes of Cin channels); workspace: per-workgroup partial products (>= Cin * C floats; more = more
 * workgroups, up to 1024).  ORCAI_E_UNSUPPORTED (nothing touched) beyond 32 channels on either side: the caller runs the two launches. */
int orcai_h_bn_bwd_pointwise_wgrad(const void* dy, const void* v, const void* u, int B, int C, int H, int W, int ksize, const float* mean, const float* var,
                                   const float* gamma, const float* beta, float eps, int relu, double* scratch2C, int sums_ready, float* dbeta, float* dgamma,
                                   const void* wtf, int Cin, void* du, float* dWpw, float* workspace, int64_t workspace_floats, void* stream);
int orcai_h_pool_bwd_bn(const void* dout, const void* ybn, int B, int C, int H, int W, int ksize, void* dy, const float* bn_gamma, const float* bn_mean,
                        const float* bn_var, float bn_eps, double* bn_sums, void* stream);
/* orcai_h_pool_bwd_bn that also reduces sum(dout) per channel -- the bias gradient of the block's residual conv, dout being the gradient of the block output
 * (f16 twin of orcai_pool_bwd_bn_bias): dout_sums f64[8 * ceil(C/8)] scratch, dbias f32[C] (overwritten; it carries dout's loss scale). */
int orcai_h_pool_bwd_bn_bias(const void* dout, const void* ybn, int B, int C, int H, int W, int ksize, void* dy, const float* bn_gamma, const float* bn_mean,
                             const float* bn_var, float bn_eps, double* bn_sums, double* dout_sums, float* dbias, void* stream);
int orcai_h_outer_reduce(const void* A, int Ca, const void* Bq, int Cb, int B, int H, int W, int ksize, int a_stride2, int Ha, int Wa, float* D, float* workspace,
                         int64_t workspace_floats, void* stream);
int orcai_h_dw_wgrad(const void* x, const void* du, int B, int C, int H, int W, int ksize_planes, int ktap, int relu_in, float* dW, void* stream);
/* orcai_dw_bwd_fused on f16 octet planes (dw_rev: f16 [ceil(C/8)][9][8] reversed taps; dr f16; dW and the sums f32 / f64): with the BatchNorm
 * arguments x is the pre-normalisation tensor and y = f16(relu(BN(x))) is formed on load -- the value orcai_h_bn_planes_apply stored.  shards: >= 16 *
 * ceil(C/8) * 32 doubles; epi 2 leaves dbeta[8 CO] | dgamma[8 CO] there for orcai_h_bn_bwd_pointwise(sums_ready = 1). */
int orcai_h_dw_bwd_fused(const void* x, const void* du, int B, int C, int H, int W, int relu_in, const void* dw_rev, void* dr, float* dW, int epi, const float* bn_mean,
                         const float* bn_var, const float* bn_gamma, const float* bn_beta, float bn_eps, int bn_relu, double* shards, void* stream);
/* orcai_h_dw_bwd_fused(epi 2) with a gradient that lives on the even pixels only added inside the pass (resq: f16 planes of C channels at the pooled
 * resolution [B][ceil(C/8)][ceil(H/2) + 2][padded_width(ceil(W/2))][8]): for block 1's first conv with x = the entry conv's stored v0 the sums left in
 * `shards` are bn0's backward sums over the TOTAL gradient (orcai_h_conv0_bn_bwd_ready = orcai_h_conv0_bn_bwd without its own sums pass), and the
 * residual branch needs no scatter-add pass over dr. */
int orcai_h_dw_bwd_fused_res(const void* x, const void* du, int B, int C, int H, int W, const void* dw_rev, void* dr, float* dW, const float* bn_mean, const float* bn_var,
                             const float* bn_gamma, const float* bn_beta, float bn_eps, int bn_relu, double* shards, const void* resq, void* stream);
int orcai_h_conv0_bn_bwd_ready(const float* in, int64_t snippet_stride, const void* dy, const void* v, int B, int H, int W, int ksize, const float* mean, const float* var,
                               const float* gamma, const float* beta, float eps, double* scratch2C, float* dbeta, float* dgamma, float* dW, void* stream);
int orcai_h_conv0_bn_bwd(const float* in, int64_t snippet_stride, const void* dy, const void* v, int B, int H, int W, int ksize, const float* mean, const float* var,
                         const float* gamma, const float* beta, float eps, double* scratch2C, float* dbeta, float* dgamma, float* dW, void* stream);
int orcai_h_pack_weights(const float* w, const int* desc, int n_desc, void* out, void* stream);
int orcai_h_feat_to_planes(const float* f, int B, int C, int H, int W, int ksize, void* out, void* stream);
int orcai_h_planes_relu_bwd(const void* dy, const void* y, int64_t n_halves, void* dx, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ORCAI_HIP_H */
