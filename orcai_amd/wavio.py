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


class WavPrefetcher:
    """Decodes the next recordings of a table on background threads while the GPU works on the current one.  The reference's table
    mode is strictly serial (predict.py:729-755); on MI355X one hour of audio is ~60 ms of GPU work but several hundred ms of file
    read + PCM16 -> float32 conversion, so without this the host side bounds table-mode throughput.  Order of results and error
    behaviour are unchanged: a decode error surfaces when THAT recording is requested (and is logged per recording by the caller)."""

    def __init__(self, paths, depth: int = 2, workers: int = 2):
        from concurrent.futures import ThreadPoolExecutor

        self.paths = [str(p) for p in paths]
        self.depth = max(1, int(depth))
        self.pool = ThreadPoolExecutor(max_workers=max(1, int(workers)), thread_name_prefix="orcai-wav")
        self.futures: dict[int, object] = {}
        self.next_to_schedule = 0
        self.index = {}
        for i, p in enumerate(self.paths):
            self.index.setdefault(p, []).append(i)

    def _schedule_up_to(self, i: int) -> None:
        while self.next_to_schedule < len(self.paths) and self.next_to_schedule <= i:
            k = self.next_to_schedule
            self.futures[k] = self.pool.submit(read_wav, self.paths[k])
            self.next_to_schedule += 1

    def get(self, path) -> tuple[np.ndarray, int]:
        """The decoded recording (as read_wav) -- from the prefetch queue when it is one of the scheduled paths."""
        slots = self.index.get(str(path))
        if not slots:
            return read_wav(path)
        i = slots.pop(0)
        # recordings before i that were scheduled but never asked for (the caller skipped them: output exists, bad row, ...):
        # cancel what has not started and drop the decoded audio of the rest -- an hour of 48 kHz mono is 0.7 GB of float32
        for k in [k for k in self.futures if k < i]:
            self.futures.pop(k).cancel()
        self.next_to_schedule = max(self.next_to_schedule, i)  # never decode a recording the caller has already passed
        self._schedule_up_to(i + self.depth)
        fut = self.futures.pop(i)
        return fut.result()

    def skip(self, path) -> None:
        """The caller will not read this recording: release its slot (and its decoded audio, if any) now."""
        slots = self.index.get(str(path))
        if slots:
            fut = self.futures.pop(slots.pop(0), None)
            if fut is not None:
                fut.cancel()

    def close(self) -> None:
        self.pool.shutdown(wait=False, cancel_futures=True)
        self.futures.clear()


_prefetcher: WavPrefetcher | None = None


def set_prefetcher(p: WavPrefetcher | None) -> None:
    global _prefetcher
    if _prefetcher is not None and p is not _prefetcher:
        _prefetcher.close()
    _prefetcher = p


def read_wav_prefetched(path: str | Path) -> tuple[np.ndarray, int]:
    """read_wav through the active WavPrefetcher, if any."""
    return _prefetcher.get(path) if _prefetcher is not None else read_wav(path)
