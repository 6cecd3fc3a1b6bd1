"""Device-side front-end engine: PCM on the GPU -> normalised spectrogram on the GPU.

Wraps the C ABI calls of include/orcai_hip.h that replace reference
``src/orcAI/spectrogram.py:34-87``.  Host logic restated here: the crop indices
(spectrogram.py:62-67) and numpy's float32 "nearest" virtual index (spectrogram.py:70-75).
"""

from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from orcai_amd import _native as N

TOP_DB = 80.0  # librosa.amplitude_to_db default, spectrogram.py:51-53


def fft_frequencies(sampling_rate: float, n_fft: int) -> np.ndarray:
    """``librosa.fft_frequencies`` (spectrogram.py:41-43)."""
    return np.fft.rfftfreq(n=n_fft, d=1.0 / sampling_rate)


def frames_to_time(n_frames: int, sampling_rate: float, hop: int) -> np.ndarray:
    """``librosa.frames_to_time(range(T))`` (spectrogram.py:45-49)."""
    return (np.arange(n_frames) * hop).astype(int) / float(sampling_rate)


def crop_indices(frequencies: np.ndarray, freq_range) -> tuple[int, int]:
    """spectrogram.py:62-67: first bin with f <= lo, first bin with f >= hi."""
    lo = int(np.argwhere(frequencies <= freq_range[0])[0][0])
    hi = int(np.argwhere(frequencies >= freq_range[1])[0][0])
    return lo, hi


def nearest_rank_index(n: int, q_fraction: float) -> int:
    """Index into the sorted flattened float32 data that ``np.percentile(a, 100*q, method="nearest")``
    picks: numpy >= 2 divides by float32(100) and rounds (n-1)*q half-to-even in float32."""
    q = np.true_divide(np.float32(100 * q_fraction), np.float32(100))
    return int(np.around(np.float32(n - 1) * q))


class FrontEnd:
    """Owns the selection workspace; one instance per device/stream."""

    def __init__(self, device: torch.device | str = "cuda"):
        self.lib = N.lib()
        self.device = torch.device(device)
        nbytes = self.lib.orcai_frontend_workspace_bytes()
        self.workspace = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)

    @staticmethod
    def _check_nfft(n_fft: int) -> None:
        """The reference reads nfft from the parameter file (spectrogram.py:34-39).  512 -- orcai-V1 and default_orcai_parameter.json -- runs the
        tuned STFT kernel, any other power of two from 32 to 4096 a plain radix-2 kernel, every other size from 2 to 4096 a direct transform (slow,
        exact); say what is not covered instead of a bare error code."""
        n = int(n_fft)
        if n < 2 or n > 4096:
            raise NotImplementedError(f"spectrogram parameter nfft = {n_fft}: the MI355X front end implements transform sizes from 2 to 4096 "
                                      "(512, the value of orcai-V1 and of default_orcai_parameter.json, on the tuned kernel); larger transforms "
                                      "need the reference's CPU path")

    # -- the whole of make_spectrogram after decode (spectrogram.py:90-147) -----------------
    def make_spectrogram(self, pcm: torch.Tensor, spectrogram_parameter: dict) -> torch.Tensor:
        """pcm: f32[N] on the device, already at spectrogram_parameter['sampling_rate'].
        Returns f32[T, K] on the device, values in [0, 1]."""
        n_fft = int(spectrogram_parameter["nfft"])
        hop = int(spectrogram_parameter["n_overlap"])
        self._check_nfft(n_fft)
        freqs = fft_frequencies(spectrogram_parameter["sampling_rate"], n_fft)
        f_lo, f_hi = crop_indices(freqs, spectrogram_parameter["freq_range"])
        if f_lo != 0:
            raise NotImplementedError("HIP front end keeps leading bins only (reference crop start is always bin 0)")
        pcm = self._check_pcm(pcm)
        n = pcm.numel()
        T = 1 + (n - (n_fft & 1)) // hop  # librosa, center=True: 1 + (n + 2 (nfft // 2) - nfft) // hop
        K = f_hi
        out = torch.empty((T, K), dtype=torch.float32, device=self.device)
        total = T * K
        r_lo = nearest_rank_index(total, spectrogram_parameter["quantiles"][0])
        r_hi = nearest_rank_index(total, spectrogram_parameter["quantiles"][1])
        N.check(
            self.lib.orcai_make_spectrogram(N.ptr(pcm), n, n_fft, hop, T, K, r_lo, r_hi, TOP_DB, N.ptr(out), N.ptr(self.workspace), N.stream_ptr()),
            "orcai_make_spectrogram",
        )
        return out

    # -- calculate_spectrogram (spectrogram.py:15-55): dB of all bins, referenced and floored ----
    def calculate_db(self, pcm: torch.Tensor, n_fft: int, hop: int) -> torch.Tensor:
        """Returns f32[T, 1 + n_fft//2] (time-major) = amplitude_to_db(|stft|, ref=max)."""
        self._check_nfft(n_fft)
        pcm = self._check_pcm(pcm)
        n = pcm.numel()
        T = 1 + (n - (n_fft & 1)) // hop  # librosa, center=True: 1 + (n + 2 (nfft // 2) - nfft) // hop
        K = 1 + n_fft // 2
        out = torch.empty((T, K), dtype=torch.float32, device=self.device)
        s = N.stream_ptr()
        ws = N.ptr(self.workspace)
        N.check(self.lib.orcai_frontend_reset(ws, s), "orcai_frontend_reset")
        N.check(self.lib.orcai_stft_db(N.ptr(pcm), n, n_fft, hop, T, K, N.ptr(out), ws, s), "orcai_stft_db")
        N.check(self.lib.orcai_frontend_finalize(1, TOP_DB, ws, s), "orcai_frontend_finalize")
        N.check(self.lib.orcai_db_reference(N.ptr(out), T * K, ws, s), "orcai_db_reference")
        return out

    # -- preprocess_spectrogram on a caller-supplied dB array (spectrogram.py:58-87) ------------
    def preprocess_db(self, db_ft: torch.Tensor, f_lo: int, f_hi: int, quantiles) -> torch.Tensor:
        """db_ft: f32[F, T] on the device (the reference's [freq, time] layout). Returns f32[T, f_hi-f_lo]."""
        assert db_ft.dtype == torch.float32 and db_ft.is_cuda and db_ft.dim() == 2
        db_ft = db_ft.contiguous()
        F, T = db_ft.shape
        K = f_hi - f_lo
        out = torch.empty((T, K), dtype=torch.float32, device=self.device)
        s = N.stream_ptr()
        ws = N.ptr(self.workspace)
        N.check(self.lib.orcai_crop_transpose(N.ptr(db_ft), F, T, f_lo, f_hi, N.ptr(out), s), "orcai_crop_transpose")
        self.normalize_inplace(out, quantiles)
        return out

    def normalize_inplace(self, x: torch.Tensor, quantiles) -> None:
        """Exact percentile clip + min-max normalise of a final-dB array, in place."""
        total = x.numel()
        s = N.stream_ptr()
        ws = N.ptr(self.workspace)
        r_lo = nearest_rank_index(total, quantiles[0])
        r_hi = nearest_rank_index(total, quantiles[1])
        N.check(self.lib.orcai_frontend_reset(ws, s), "orcai_frontend_reset")
        N.check(self.lib.orcai_hist_level1(N.ptr(x), total, ws, s), "orcai_hist_level1")
        N.check(self.lib.orcai_quantile_select(N.ptr(x), total, r_lo, r_hi, ws, s), "orcai_quantile_select")
        N.check(self.lib.orcai_frontend_finalize(0, TOP_DB, ws, s), "orcai_frontend_finalize")
        N.check(self.lib.orcai_clip_normalize(N.ptr(x), total, ws, s), "orcai_clip_normalize")

    def select(self, x: torch.Tensor, rank_lo: int, rank_hi: int) -> tuple[float, float]:
        """Exact rank_lo-th / rank_hi-th smallest of a device f32 array (test hook for the selection kernels)."""
        total = x.numel()
        s = N.stream_ptr()
        ws = N.ptr(self.workspace)
        N.check(self.lib.orcai_frontend_reset(ws, s), "orcai_frontend_reset")
        N.check(self.lib.orcai_hist_level1(N.ptr(x), total, ws, s), "orcai_hist_level1")
        N.check(self.lib.orcai_quantile_select(N.ptr(x), total, rank_lo, rank_hi, ws, s), "orcai_quantile_select")
        N.check(self.lib.orcai_frontend_finalize(0, TOP_DB, ws, s), "orcai_frontend_finalize")
        st = self.stats()
        return st["sel_lo_raw"], st["sel_hi_raw"]

    def stats(self) -> dict:
        """Synchronises; {pmax, ref_db, p_lo, p_hi, sel_lo_raw, sel_hi_raw} of the last run."""
        buf = (C.c_float * 6)()
        N.check(self.lib.orcai_frontend_stats_host(N.ptr(self.workspace), buf, N.stream_ptr()), "orcai_frontend_stats_host")
        keys = ["pmax", "ref_db", "p_lo", "p_hi", "sel_lo_raw", "sel_hi_raw"]
        return {k: float(np.float32(v)) for k, v in zip(keys, buf)}

    def _check_pcm(self, pcm: torch.Tensor) -> torch.Tensor:
        if not (isinstance(pcm, torch.Tensor) and pcm.is_cuda and pcm.dtype == torch.float32 and pcm.dim() == 1):
            raise TypeError("pcm must be a 1-D float32 CUDA tensor")
        return pcm.contiguous()


_FRONTENDS: dict = {}


def get_frontend(device: torch.device | str | None = None) -> FrontEnd:
    if not torch.cuda.is_available():
        raise RuntimeError("orcai_amd needs a ROCm GPU: there is no CPU fallback for the front end")
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    if key not in _FRONTENDS:
        _FRONTENDS[key] = FrontEnd(torch.device("cuda", key[1]))
    return _FRONTENDS[key]
