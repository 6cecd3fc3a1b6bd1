"""Sampling-rate conversion on the GPU (the resampling half of ``librosa.load(sr=...)``, reference
``src/orcAI/spectrogram.py:23-27``).

librosa's default ``res_type="soxr_hq"`` cannot be matched bit-for-bit (libsoxr is absent: parity unpinned,
SURVEY 8c).  This is an exact rational polyphase Kaiser-windowed-sinc filter; the table is designed here in
float64 and applied by ``csrc/resample.hip``.
"""

from __future__ import annotations

import math
from functools import lru_cache

import numpy as np
import torch

from orcai_amd import _native as N

NUM_ZEROS = 64
KAISER_BETA = 14.769656459379492
ROLLOFF = 0.9475937167399596


def ratio(sr_in: int, sr_out: int) -> tuple[int, int]:
    g = math.gcd(int(sr_in), int(sr_out))
    return int(sr_out) // g, int(sr_in) // g  # L (up), M (down)


def output_length(n_in: int, sr_in: int, sr_out: int) -> int:
    """``librosa.resample`` output length: ceil(n * sr_out / sr_in)."""
    return int(math.ceil(n_in * sr_out / sr_in))


@lru_cache(maxsize=8)
def design_table(L: int, M: int) -> np.ndarray:
    """float32 [L][ntaps]: tap j of phase p weighs input sample i0 - ntaps/2 + 1 + j for output time i0 + p/L."""
    scale = min(1.0, L / M)
    half = int(math.ceil(NUM_ZEROS / scale))
    ntaps = 2 * half
    ntaps += (-ntaps) % 4
    j = np.arange(ntaps, dtype=np.float64)[None, :]
    p = np.arange(L, dtype=np.float64)[:, None]
    t = j - (ntaps // 2 - 1) - p / L  # input offset relative to the output instant
    fc = scale * ROLLOFF
    u = t * scale / NUM_ZEROS
    inside = np.abs(u) < 1.0
    win = np.where(inside, np.i0(KAISER_BETA * np.sqrt(np.clip(1.0 - u * u, 0.0, 1.0))) / np.i0(KAISER_BETA), 0.0)
    return np.ascontiguousarray((fc * np.sinc(fc * t) * win).astype(np.float32))


def resample_device(pcm: torch.Tensor, sr_in: int, sr_out: int) -> torch.Tensor:
    """f32 cuda [N] at sr_in -> f32 cuda [ceil(N*sr_out/sr_in)] at sr_out."""
    if sr_in == sr_out:
        return pcm
    if not (pcm.is_cuda and pcm.dtype == torch.float32 and pcm.dim() == 1):
        raise TypeError("pcm must be a 1-D float32 CUDA tensor")
    L, M = ratio(sr_in, sr_out)
    table = torch.from_numpy(design_table(L, M)).to(pcm.device)
    n_in = pcm.numel()
    n_out = output_length(n_in, sr_in, sr_out)
    out = torch.empty(n_out, dtype=torch.float32, device=pcm.device)
    x = pcm.contiguous()
    N.check(N.lib().orcai_resample_polyphase(N.ptr(x), n_in, N.ptr(out), n_out, L, M, N.ptr(table), table.shape[1], N.stream_ptr()), "orcai_resample_polyphase")
    return out
