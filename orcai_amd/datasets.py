"""Snippet datasets for training.  The reference stores tf.data snapshots (``io.py:150-218``), a TensorFlow container
format that cannot be read without TensorFlow (out of scope); this package stores the same elements --
(spectrogram f32[736,171], labels f32[46,7]) -- as two raw ``.npy`` arrays per split and reproduces the loading
semantics of ``load_dataset`` (io.py:174-182): shuffle(buffer 1000, seeded, reshuffled each epoch) -> batch(drop
remainder) -> prefetch (a background thread stages pinned host batches and copies them to the GPU).
"""

from __future__ import annotations

import queue
import threading
from pathlib import Path

import numpy as np
import torch

SHUFFLE_BUFFER_SIZE = 1000  # io.py:13


def reshape_labels(labels: np.ndarray, n_filters: int) -> np.ndarray:
    """Frame labels (T, L) -> (T / 2**n, L): mean over groups of 2**n frames, rounded half to even (io.py:101-126)."""
    f = 2**n_filters
    if labels.shape[0] % f != 0:
        raise ValueError("The number of rows in 'arr' must be divisible by 2**'n_filters'.")
    avg = np.asarray(labels, dtype=np.float32).reshape(labels.shape[0] // f, f, labels.shape[1]).mean(axis=1)
    return np.round(avg).astype(np.float32)


def save_dataset(spectrograms: np.ndarray, labels: np.ndarray, path: Path | str, overwrite: bool = False) -> None:
    """<path>/{spectrogram.npy, labels.npy} (stands where the reference calls Dataset.save, io.py:187-218)."""
    path = Path(path)
    if path.exists() and not overwrite:
        raise FileExistsError(f"File {path} already exists.")
    path.mkdir(parents=True, exist_ok=True)
    np.save(path / "spectrogram.npy", np.ascontiguousarray(spectrograms, dtype=np.float32))
    np.save(path / "labels.npy", np.ascontiguousarray(labels, dtype=np.float32))


class SnippetDataset:
    """Iterable of (spectrogram cuda f32 [B][H][W], labels cuda f32 [B][T][L]) batches."""

    def __init__(self, path: Path | str, batch_size: int, seed=None, shuffle: bool = True, rank: int = 0, world_size: int = 1):
        path = Path(path)
        self.x = np.load(path / "spectrogram.npy", mmap_mode="r")
        self.y = np.load(path / "labels.npy", mmap_mode="r")
        if self.x.ndim == 4:
            self.x = self.x[..., 0]
        self.batch_size = int(batch_size)
        self.shuffle = shuffle
        self.seed = int(np.random.SeedSequence(seed).generate_state(1)[0])  # io.py:176-179
        self.epoch = 0
        self.rank, self.world_size = rank, world_size

    def __len__(self) -> int:
        return (len(self.x) // self.batch_size) // self.world_size

    def _order(self) -> np.ndarray:
        n = len(self.x)
        if not self.shuffle:
            return np.arange(n)
        rng = np.random.default_rng([self.seed, self.epoch])
        buf = list(range(min(SHUFFLE_BUFFER_SIZE, n)))  # tf.data shuffle-buffer semantics
        nxt = len(buf)
        out = np.empty(n, dtype=np.int64)
        for i in range(n):
            j = int(rng.integers(len(buf)))
            out[i] = buf[j]
            if nxt < n:
                buf[j] = nxt
                nxt += 1
            else:
                buf[j] = buf[-1]
                buf.pop()
        return out

    def __iter__(self):
        order = self._order()
        self.epoch += 1
        nb = len(order) // self.batch_size
        batches = [order[i * self.batch_size : (i + 1) * self.batch_size] for i in range(nb)]
        batches = batches[self.rank :: self.world_size][: len(self)]  # each rank its own batches, same count on all ranks
        q: queue.Queue = queue.Queue(maxsize=3)

        def producer():
            for idx in batches:
                srt = np.sort(idx)
                inv = np.argsort(np.argsort(idx))
                xb = torch.from_numpy(np.ascontiguousarray(self.x[srt])[inv]).pin_memory() if torch.cuda.is_available() else torch.from_numpy(np.ascontiguousarray(self.x[srt])[inv])
                yb = torch.from_numpy(np.ascontiguousarray(self.y[srt])[inv])
                q.put((xb, yb))
            q.put(None)

        threading.Thread(target=producer, daemon=True).start()
        while True:
            item = q.get()
            if item is None:
                return
            xb, yb = item
            if torch.cuda.is_available():
                yield xb.cuda(non_blocking=True), yb.cuda(non_blocking=True)
            else:
                yield xb, yb


def load_dataset(path: Path | str, batch_size: int, compression: str = "GZIP", seed=None, rank: int = 0, world_size: int = 1) -> SnippetDataset:
    """Same call shape as the reference's load_dataset (io.py:150-184); `compression` is accepted and ignored (raw .npy)."""
    return SnippetDataset(path, batch_size, seed=seed, shuffle=True, rank=rank, world_size=world_size)


def make_synthetic_dataset(path: Path | str, n: int, seed: int = 4, input_shape=(736, 171), out_steps: int = 46, n_labels: int = 7, overwrite: bool = True) -> None:
    """U(0,1) snippets with smooth blobs; the label of a step is 1 where the blob of that label's band is
    present; one label column is masked (-1) in 30 % of the snippets."""
    rng = np.random.default_rng(seed)
    H, W = input_shape
    x = rng.random((n, H, W), dtype=np.float32) * 0.5
    y = np.zeros((n, out_steps, n_labels), dtype=np.float32)
    step = H // out_steps
    band = W // n_labels
    for i in range(n):
        for lab in rng.choice(n_labels, size=2, replace=False):
            s0 = int(rng.integers(0, out_steps - 6))
            ln = int(rng.integers(3, 7))
            x[i, s0 * step : (s0 + ln) * step, lab * band : (lab + 1) * band] += 0.5
            y[i, s0 : s0 + ln, lab] = 1.0
        if rng.random() < 0.3:
            y[i, :, int(rng.integers(n_labels))] = -1.0
    save_dataset(np.clip(x, 0, 1), y, path, overwrite=overwrite)
