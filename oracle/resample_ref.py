"""CPU restatement of the polyphase resampler (TEST INFRASTRUCTURE).  The reference resamples with libsoxr via
``librosa.load`` (spectrogram.py:23-27); soxr is absent, so this stage is **parity unpinned**: the oracle restates
the build's own filter (same float64 design) in float64 arithmetic and is sanity-checked against
``scipy.signal.resample_poly`` on band-limited signals."""

from __future__ import annotations

import math

import numpy as np

NUM_ZEROS = 64
KAISER_BETA = 14.769656459379492
ROLLOFF = 0.9475937167399596


def resample_ref(x: np.ndarray, sr_in: int, sr_out: int) -> np.ndarray:
    g = math.gcd(sr_in, sr_out)
    L, M = sr_out // g, sr_in // g
    scale = min(1.0, L / M)
    half = int(math.ceil(NUM_ZEROS / scale))
    ntaps = 2 * half
    ntaps += (-ntaps) % 4
    n_out = int(math.ceil(len(x) * sr_out / sr_in))
    fc = scale * ROLLOFF
    xp = np.concatenate([np.zeros(ntaps), x.astype(np.float64), np.zeros(ntaps)])
    out = np.empty(n_out)
    n = np.arange(n_out, dtype=np.int64)
    i0 = (n * M) // L
    ph = (n * M) % L
    j = np.arange(ntaps)
    for p in np.unique(ph):
        sel = np.nonzero(ph == p)[0]
        t = j - (ntaps // 2 - 1) - p / L
        u = t * scale / NUM_ZEROS
        win = np.where(np.abs(u) < 1, np.i0(KAISER_BETA * np.sqrt(np.clip(1 - u * u, 0, 1))) / np.i0(KAISER_BETA), 0.0)
        h = (fc * np.sinc(fc * t) * win).astype(np.float32).astype(np.float64)
        k0 = i0[sel] - ntaps // 2 + 1 + ntaps  # index into the padded signal
        idx = k0[:, None] + j[None, :]
        out[sel] = (xp[idx] * h[None, :]).sum(axis=1)
    return out.astype(np.float32)
