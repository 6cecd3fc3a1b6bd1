"""Seeded synthetic recordings for tests and benchmarks: white noise + linear chirps + clicks."""

from __future__ import annotations

import numpy as np


def synth_recording(seconds: float, sampling_rate: int = 48000, seed: int = 20250620, as_int16: bool = True) -> np.ndarray:
    """0.2*N(0,1) noise, a 1->9 kHz 0.5 s chirp (amplitude 0.5) every 15-20 s, 20 clicks per minute."""
    rng = np.random.default_rng(seed)
    n = int(round(seconds * sampling_rate))
    x = (0.2 * rng.standard_normal(n)).astype(np.float32)
    t_chirp = np.arange(int(0.5 * sampling_rate)) / sampling_rate
    chirp = (0.5 * np.sin(2 * np.pi * (1000.0 * t_chirp + 0.5 * (8000.0 / 0.5) * t_chirp**2))).astype(np.float32)
    start = 5.0
    while start + 0.5 < seconds:
        s = int(start * sampling_rate)
        x[s : s + len(chirp)] += chirp
        start += 15.0 if (int(start) % 2) else 20.0
    n_clicks = max(1, int(20 * seconds / 60))
    for pos in rng.integers(0, max(n - 64, 1), size=n_clicks):
        x[pos : pos + 32] += 0.8 * np.hanning(32).astype(np.float32)[: max(0, min(32, n - pos))]
    x = np.clip(x / 1.6, -1.0, 1.0 - 1.0 / 32768)
    if as_int16:
        return np.round(x * 32767.0).astype(np.int16)
    return x.astype(np.float32)


def pcm16_to_float(x: np.ndarray) -> np.ndarray:
    return x.astype(np.float32) / np.float32(32768.0)
