"""Inputs of the golden vectors that are too large to store: built from integer arithmetic only (a splitmix64 hash of the element
index), so the array is the same on every numpy / libm, and the test never has to skip."""

import numpy as np


def hashed_db(seed: int, T: int, F: int = 257) -> np.ndarray:
    """float32 [F, T] dB-like array on a 0.01 dB grid in [-80, 0]: element i = -80 + (splitmix64(seed, i) mod 8001) / 100, with the
    global maximum 0 planted like amplitude_to_db(ref=max) would."""
    i = np.arange(F * T, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15) * (i + np.uint64(1))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    level = (z % np.uint64(8001)).astype(np.int64)  # 0..8000
    x = ((level - 8000).astype(np.float64) / 100.0).astype(np.float32).reshape(F, T)
    x[7, 3 % T] = 0.0
    return x
