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


def rank_batches(order: np.ndarray, batch_size: int, rank: int, world_size: int, dp_batch: str) -> list:
    """Which snippets of an epoch's shuffled `order` this rank sees, step by step.
    "replicate" (throughput mode): `batch_size` is the PER-RANK batch; the global batch list is dealt round-robin, every rank
        takes len // world full batches -- the optimiser sees batch_size x world snippets per step and 1 / world of the steps.
    "split" (the reference's MirroredStrategy contract, hpsearch.py:170-205): `batch_size` is the GLOBAL batch; every global batch is
        cut into world contiguous slices of batch_size / world snippets, one per rank -- same number of steps per epoch as one GPU,
        same snippets per optimiser step; gradients are averaged over the ranks, BatchNorm statistics stay per replica."""
    nb = len(order) // batch_size
    batches = [order[i * batch_size : (i + 1) * batch_size] for i in range(nb)]
    if dp_batch == "split":
        per = batch_size // world_size
        return [b[rank * per : (rank + 1) * per] for b in batches]
    return batches[rank::world_size][: nb // world_size]


def check_dp_batch(dp_batch: str, batch_size: int, world_size: int) -> str:
    if dp_batch not in ("replicate", "split"):
        raise ValueError(f"dp_batch must be 'replicate' or 'split', got {dp_batch!r}")
    if dp_batch == "split" and batch_size % world_size:
        raise ValueError(f"dp_batch 'split': the global batch size {batch_size} is not divisible by the {world_size} ranks")
    return dp_batch


class SnippetDataset:
    """Iterable of (spectrogram cuda f32 [B][H][W], labels cuda f32 [B][T][L]) batches."""

    def __init__(self, path: Path | str, batch_size: int, seed=None, shuffle: bool = True, rank: int = 0, world_size: int = 1, dp_batch: str = "replicate"):
        path = Path(path)
        self.dp_batch = check_dp_batch(dp_batch, int(batch_size), world_size)
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
        nb = len(self.x) // self.batch_size
        return nb if self.dp_batch == "split" else nb // self.world_size

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
        batches = rank_batches(order, self.batch_size, self.rank, self.world_size, self.dp_batch)  # same count on all ranks
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


def load_dataset(path: Path | str, batch_size: int, compression: str = "GZIP", seed=None, rank: int = 0, world_size: int = 1,
                 dp_batch: str = "replicate") -> SnippetDataset:
    """Same call shape as the reference's load_dataset (io.py:150-184); `compression` is accepted and ignored (raw .npy).
    A directory holding ``snippet_table_dataset.json`` ({"snippet_table": <csv path>, "n_filters": n}: a descriptor, nothing materialised) gives a
    SnippetTableDataset that gathers its batches on the GPU from the recordings' arrays."""
    desc = Path(path) / "snippet_table_dataset.json"
    if desc.exists():
        import json

        import pandas as pd

        d = json.loads(desc.read_text())
        table = pd.read_csv((Path(path) / d["snippet_table"]).resolve())
        return SnippetTableDataset(table, d["n_filters"], batch_size, seed=seed, shuffle=True, rank=rank, world_size=world_size, dp_batch=dp_batch)
    return SnippetDataset(path, batch_size, seed=seed, shuffle=True, rank=rank, world_size=world_size, dp_batch=dp_batch)


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


# ------------------------------------------------------------------------------------------------------------------
# HBM-resident snippet store: the snippet table indexes rows of per-recording arrays; nothing is
# materialised.  Replaces DataLoader (io.py:17-147) + Dataset.save/load (io.py:150-218) for training.
# ------------------------------------------------------------------------------------------------------------------
def snippet_rows(times_meta: dict, t_start: float, snippet_duration: float, n_filters: int) -> tuple[int, int]:
    """Index arithmetic of ``_make_snippet_table`` (snippets.py:98-133): the spectrogram time axis is
    ``linspace(min, max, length)``; a snippet starting at ``t_start`` covers rows
    ``[searchsorted(times, t_start, "left") - 1,  + 2**n * ((snippet_duration / delta_t) // 2**n))``."""
    times = np.linspace(times_meta["min"], times_meta["max"], times_meta["length"])
    delta_t = times[1] - times[0]
    f = 2**n_filters
    n_steps = int(f * ((snippet_duration / delta_t) // f))
    start = int(np.searchsorted(times, t_start, side="left") - 1)
    return start, start + n_steps


class RecordingStore:
    """Spectrogram [T,W] and label [T,L] arrays of many recordings, concatenated along time and resident in HBM.
    On disk (the stand-in for spectrogram.zarr / labels.zarr, io.py:296-331): ``<dir>/spectrogram/spectrogram.npy`` and
    ``<dir>/labels/labels.npy``."""

    def __init__(self, recording_dirs, device=None):
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.offsets: dict[str, int] = {}
        self.lengths: dict[str, int] = {}
        specs, labs = [], []
        total = 0
        for d in dict.fromkeys(str(x) for x in recording_dirs):  # unique, first-seen order
            s = np.load(Path(d) / "spectrogram" / "spectrogram.npy", mmap_mode="r")
            lab = np.load(Path(d) / "labels" / "labels.npy", mmap_mode="r")
            if s.shape[0] != lab.shape[0]:
                raise ValueError(f"{d}: spectrogram has {s.shape[0]} rows, labels {lab.shape[0]}")
            self.offsets[d], self.lengths[d] = total, int(s.shape[0])
            total += int(s.shape[0])
            specs.append(torch.from_numpy(np.array(s, dtype=np.float32)))  # np.array copies out of the read-only memory map
            labs.append(torch.from_numpy(np.array(lab, dtype=np.float32)))
        self.spec = torch.cat(specs).to(self.device)
        self.labels = torch.cat(labs).to(self.device)
        self.total_rows = total

    def global_rows(self, recording_dirs, row_start, row_stop) -> np.ndarray:
        """Store row of the first row of every snippet; bounds are checked here, on the host, once."""
        start = np.asarray(row_start, dtype=np.int64)
        stop = np.asarray(row_stop, dtype=np.int64)
        out = np.empty(len(start), dtype=np.int64)
        for i, d in enumerate(recording_dirs):
            d = str(d)
            if start[i] < 0 or stop[i] > self.lengths[d] or stop[i] <= start[i]:
                raise IndexError(f"snippet {i}: rows [{start[i]}, {stop[i]}) outside recording {d} of {self.lengths[d]} rows")
            out[i] = self.offsets[d] + start[i]
        return out


class SnippetTableDataset:
    """Batches (spectrogram cuda f32 [B][H][W], labels cuda f32 [B][H/2**n][L]) gathered on the GPU from a RecordingStore by the
    kernels of csrc/datapath.hip.  ``snippet_table``: DataFrame with columns recording_data_dir, row_start, row_stop (the table
    DataLoader takes, io.py:20-37).  Same iteration contract as SnippetDataset (shuffle buffer, drop remainder, rank slicing)."""

    def __init__(self, snippet_table, n_filters: int, batch_size: int, seed=None, shuffle: bool = True, rank: int = 0, world_size: int = 1,
                 store: RecordingStore | None = None, dp_batch: str = "replicate"):
        from orcai_amd import _native as N

        self._N = N
        self.lib = N.lib()
        dirs = [str(d) for d in snippet_table["recording_data_dir"]]
        self.store = store if store is not None else RecordingStore(dirs)
        lengths = (np.asarray(snippet_table["row_stop"], dtype=np.int64) - np.asarray(snippet_table["row_start"], dtype=np.int64))
        if len(lengths) and not np.all(lengths == lengths[0]):
            raise ValueError("all snippets must have the same number of rows")
        self.rows = int(lengths[0]) if len(lengths) else 0
        self.factor = 2**n_filters
        if self.rows % self.factor:
            raise ValueError("The number of rows in 'arr' must be divisible by 2**'n_filters'.")  # io.py:123-126
        self.starts = torch.from_numpy(self.store.global_rows(dirs, snippet_table["row_start"], snippet_table["row_stop"])).to(self.store.device)
        self.batch_size = int(batch_size)
        self.shuffle = shuffle
        self.seed = int(np.random.SeedSequence(seed).generate_state(1)[0])
        self.epoch = 0
        self.rank, self.world_size = rank, world_size
        self.dp_batch = check_dp_batch(dp_batch, self.batch_size, world_size)

    def __len__(self) -> int:
        nb = len(self.starts) // self.batch_size
        return nb if self.dp_batch == "split" else nb // self.world_size

    _order = SnippetDataset._order

    @property
    def x(self):  # SnippetDataset._order only needs len(self.x)
        return self.starts

    def batch(self, idx: torch.Tensor):
        N, lib, st = self._N, self.lib, self.store
        B = int(idx.numel())
        W, L = int(st.spec.shape[1]), int(st.labels.shape[1])
        starts = self.starts[idx].contiguous()
        x = torch.empty((B, self.rows, W), dtype=torch.float32, device=st.device)
        y = torch.empty((B, self.rows // self.factor, L), dtype=torch.float32, device=st.device)
        s = N.stream_ptr()
        N.check(lib.orcai_gather_snippets(st.spec.data_ptr(), starts.data_ptr(), B, self.rows, W, x.data_ptr(), s), "orcai_gather_snippets")
        N.check(lib.orcai_downsample_labels(st.labels.data_ptr(), starts.data_ptr(), B, self.rows, L, self.factor, y.data_ptr(), s), "orcai_downsample_labels")
        return x, y

    def __iter__(self):
        order = self._order()
        self.epoch += 1
        for idx in rank_batches(order, self.batch_size, self.rank, self.world_size, self.dp_batch):
            yield self.batch(torch.from_numpy(np.asarray(idx, dtype=np.int64)).to(self.store.device))
