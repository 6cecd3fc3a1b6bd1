"""CPU restatement (numpy/scipy) of the reference front end.  TEST INFRASTRUCTURE.

Follows ``/root/reference/src/orcAI/spectrogram.py:15-87``.  The librosa 0.11.0
calls made there (``stft``, ``amplitude_to_db``, ``fft_frequencies``,
``frames_to_time``) are restated from librosa's documented behaviour because
librosa is not installed in this image: **parity unpinned at the librosa
boundary**.  ``preprocess_spectrogram_ref`` is pinned bit-exactly against the
reference's own function by ``tests/golden/preprocess_*.npz``.
"""

from __future__ import annotations

import numpy as np
import scipy.signal

AMIN = 1e-5  # librosa.amplitude_to_db default
TOP_DB = 80.0  # librosa.amplitude_to_db default


def hann_window(n_fft: int) -> np.ndarray:
    """Periodic Hann, float64 -- what ``librosa.stft(window="hann")`` builds via
    ``scipy.signal.get_window("hann", n_fft, fftbins=True)`` (spectrogram.py:34-39)."""
    return scipy.signal.get_window("hann", n_fft, fftbins=True)


def num_frames(n_samples: int, hop: int, n_fft: int = 512) -> int:
    """Centred STFT frame count (librosa ``center=True`` pads ``n_fft // 2`` on both sides): ``1 + (N + 2 (n_fft // 2) - n_fft) // hop``,
    i.e. ``1 + N // hop`` for even transform sizes."""
    return 1 + (n_samples - (n_fft & 1)) // hop


def stft_ref(y: np.ndarray, n_fft: int = 512, hop: int = 256, block: int = 8192) -> np.ndarray:
    """Centred STFT, zero ("constant") padding of n_fft//2 either side, periodic Hann,
    rFFT.  spectrogram.py:34-39 -> ``librosa.stft(y, n_fft, hop_length, window="hann")``.

    librosa multiplies the float32 frames by the float64 window and calls
    ``numpy.fft.rfft`` (float64 arithmetic), then stores into a complex64 matrix.
    Returns complex64 ``[1 + n_fft//2, T]``.
    """
    y = np.asarray(y, dtype=np.float32)
    pad = n_fft // 2
    yp = np.pad(y, (pad, pad), mode="constant")
    T = num_frames(len(y), hop, n_fft)
    win = hann_window(n_fft)
    out = np.empty((1 + n_fft // 2, T), dtype=np.complex64)
    frames = np.lib.stride_tricks.sliding_window_view(yp, n_fft)[::hop]
    assert frames.shape[0] == T
    for s in range(0, T, block):
        e = min(s + block, T)
        out[:, s:e] = np.fft.rfft(win[None, :] * frames[s:e], axis=1).T
    return out


def fft_frequencies_ref(sr: float, n_fft: int) -> np.ndarray:
    """``librosa.fft_frequencies`` == ``np.fft.rfftfreq(n_fft, 1/sr)`` (spectrogram.py:41-43)."""
    return np.fft.rfftfreq(n=n_fft, d=1.0 / sr)


def frames_to_time_ref(n_frames: int, sr: float, hop: int) -> np.ndarray:
    """``librosa.frames_to_time(range(T), sr, hop_length)`` (spectrogram.py:45-49)."""
    samples = (np.arange(n_frames) * hop).astype(int)
    return samples / float(sr)


def amplitude_to_db_ref(S: np.ndarray) -> np.ndarray:
    """``librosa.amplitude_to_db(np.abs(S), ref=np.max)`` (spectrogram.py:51-53).

    magnitude = |S| (f32); ref = max(magnitude); power = magnitude**2;
    10*log10(max(amin**2, power)) - 10*log10(max(amin**2, ref**2)); floor at max-80.
    All arithmetic stays float32 (numpy weak-scalar promotion).
    """
    magnitude = np.abs(S).astype(np.float32, copy=False)
    ref_value = np.max(magnitude)
    power = np.square(magnitude)
    amin = AMIN**2
    log_spec = 10.0 * np.log10(np.maximum(amin, power))
    log_spec -= 10.0 * np.log10(np.maximum(amin, ref_value**2))
    log_spec = np.maximum(log_spec, log_spec.max() - TOP_DB)
    assert log_spec.dtype == np.float32
    return log_spec


def calculate_spectrogram_ref(y: np.ndarray, spectrogram_parameter: dict):
    """spectrogram.py:15-55 minus the file decode: takes the already-decoded mono f32 signal."""
    sr = spectrogram_parameter["sampling_rate"]
    n_fft = spectrogram_parameter["nfft"]
    hop = spectrogram_parameter["n_overlap"]  # used as hop_length, spectrogram.py:37
    S = stft_ref(y, n_fft, hop)
    frequencies = fft_frequencies_ref(sr, n_fft)
    times = frames_to_time_ref(S.shape[1], sr, hop)
    return amplitude_to_db_ref(S), frequencies, times


def crop_indices(frequencies: np.ndarray, freq_range) -> tuple[int, int]:
    """spectrogram.py:62-67 -- first bin with f <= lo (always 0) and first bin with f >= hi."""
    lo = int(np.argwhere(frequencies <= freq_range[0])[0][0])
    hi = int(np.argwhere(frequencies >= freq_range[1])[0][0])
    return lo, hi


def nearest_rank_index(n: int, q_fraction: float) -> int:
    """Index into the sorted flattened float32 array that
    ``np.percentile(a, 100*q_fraction, method="nearest")`` selects (numpy >= 2.0).

    numpy divides the percentile by ``float32(100)`` for float32 input and
    evaluates ``around((n-1) * q)`` in float32 (round-half-even).
    """
    q_pct = 100 * q_fraction  # python float, as written at spectrogram.py:70-75
    q = np.true_divide(np.float32(q_pct), np.float32(100))
    idx = np.around(np.float32(n - 1) * q)
    return int(idx)


def preprocess_spectrogram_ref(spectrogram: np.ndarray, frequencies: np.ndarray, spectrogram_parameter: dict) -> np.ndarray:
    """spectrogram.py:58-87 -- crop, percentile clip ("nearest"), min-max normalise, transpose."""
    lo, hi = crop_indices(frequencies, spectrogram_parameter["freq_range"])
    spec = spectrogram[lo:hi, :]
    flat = np.sort(spec, axis=None)
    n = flat.size
    p_lo = flat[nearest_rank_index(n, spectrogram_parameter["quantiles"][0])]
    p_hi = flat[nearest_rank_index(n, spectrogram_parameter["quantiles"][1])]
    spec = np.clip(spec, p_lo, p_hi)
    mn = np.min(spec)
    mx = np.max(spec)
    with np.errstate(invalid="ignore", divide="ignore"):
        spec = (spec - mn) / (mx - mn)
    return spec.T


def make_spectrogram_ref(y: np.ndarray, orcai_parameter: dict):
    """spectrogram.py:90-147 on a decoded signal: returns (f32[T,K], f64[257], f64[T])."""
    sp = orcai_parameter["spectrogram"]
    db, frequencies, times = calculate_spectrogram_ref(y, sp)
    return preprocess_spectrogram_ref(db, frequencies, sp), frequencies, times
