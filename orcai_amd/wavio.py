"""WAV decode for the front end (replaces the soundfile half of ``librosa.load``,
reference ``src/orcAI/spectrogram.py:23-27``).  Host-side, numpy only.

Scaling follows libsndfile's float read: PCM16 / 2**15, PCM24 / 2**23, PCM32 / 2**31,
unsigned 8-bit (x - 128) / 2**7, IEEE float passed through.
"""

from __future__ import annotations

import struct
from pathlib import Path

import numpy as np

WAVE_FORMAT_PCM = 1
WAVE_FORMAT_IEEE_FLOAT = 3
WAVE_FORMAT_EXTENSIBLE = 0xFFFE


def read_wav(path: str | Path) -> tuple[np.ndarray, int]:
    """Returns (float32 array [channels, frames], sampling rate)."""
    data = Path(path).read_bytes()
    if len(data) < 12 or data[0:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    pos = 12
    fmt = None
    payload = None
    while pos + 8 <= len(data):
        cid = data[pos : pos + 4]
        size = struct.unpack_from("<I", data, pos + 4)[0]
        body = data[pos + 8 : pos + 8 + size]
        if cid == b"fmt ":
            tag, channels, rate, _, block_align, bits = struct.unpack_from("<HHIIHH", body, 0)
            if tag == WAVE_FORMAT_EXTENSIBLE and len(body) >= 26:
                tag = struct.unpack_from("<H", body, 24)[0]
            fmt = (tag, channels, rate, block_align, bits)
        elif cid == b"data":
            payload = body
        pos += 8 + size + (size & 1)
    if fmt is None or payload is None:
        raise ValueError(f"{path}: missing fmt or data chunk")
    tag, channels, rate, block_align, bits = fmt
    frame_bytes = channels * (bits // 8)
    n = len(payload) // frame_bytes
    payload = payload[: n * frame_bytes]
    if tag == WAVE_FORMAT_PCM:
        if bits == 16:
            x = np.frombuffer(payload, dtype="<i2").astype(np.float32) / np.float32(32768.0)
        elif bits == 8:
            x = (np.frombuffer(payload, dtype=np.uint8).astype(np.float32) - np.float32(128.0)) / np.float32(128.0)
        elif bits == 24:
            b = np.frombuffer(payload, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
            v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
            v = np.where(v & 0x800000, v - 0x1000000, v)
            x = (v.astype(np.float64) / 8388608.0).astype(np.float32)
        elif bits == 32:
            x = (np.frombuffer(payload, dtype="<i4").astype(np.float64) / 2147483648.0).astype(np.float32)
        else:
            raise ValueError(f"{path}: unsupported PCM width {bits}")
    elif tag == WAVE_FORMAT_IEEE_FLOAT:
        if bits == 32:
            x = np.frombuffer(payload, dtype="<f4").astype(np.float32)
        elif bits == 64:
            x = np.frombuffer(payload, dtype="<f8").astype(np.float32)
        else:
            raise ValueError(f"{path}: unsupported float width {bits}")
    else:
        raise ValueError(f"{path}: unsupported WAVE format tag {tag}")
    return np.ascontiguousarray(x.reshape(n, channels).T), int(rate)


def write_wav_pcm16(path: str | Path, samples: np.ndarray, rate: int) -> None:
    """samples: int16 [frames] or [channels, frames]."""
    s = np.asarray(samples)
    if s.dtype != np.int16:
        raise TypeError("write_wav_pcm16 wants int16 samples")
    if s.ndim == 1:
        s = s[None, :]
    channels, n = s.shape
    payload = np.ascontiguousarray(s.T).astype("<i2").tobytes()
    header = b"RIFF" + struct.pack("<I", 36 + len(payload)) + b"WAVE"
    header += b"fmt " + struct.pack("<IHHIIHH", 16, WAVE_FORMAT_PCM, channels, rate, rate * channels * 2, channels * 2, 16)
    header += b"data" + struct.pack("<I", len(payload))
    Path(path).write_bytes(header + payload)
