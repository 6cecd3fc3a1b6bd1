"""CPU restatement (numpy/pandas) of the reference's snippet indexing and label
extraction.  TEST INFRASTRUCTURE.

Follows ``/root/reference/src/orcAI/predict.py:235-340,343-364,474-499`` and
``/root/reference/src/orcAI/auxiliary.py:420-440``.  Pinned bit-exactly by
``tests/golden/aggregate_*.npz`` / ``labels_*.json`` which were produced by the
reference's own functions (tests/golden/make_golden.py).
"""

from __future__ import annotations

import numpy as np
import pandas as pd


def snippet_geometry(n_frames: int, snippet_length: int, n_filters: int):
    """predict.py:242-253,276 -- (shift, steps per snippet, n_snippets, total output steps)."""
    shift = snippet_length // 2
    tpo = 2**n_filters
    prediction_length = snippet_length // tpo
    num_snippets = (n_frames - snippet_length) // shift + 1
    total_steps = n_frames // tpo
    return shift, tpo, prediction_length, num_snippets, total_steps


def slice_snippets(spectrogram: np.ndarray, snippet_length: int) -> np.ndarray:
    """predict.py:253-264 -- materialised [n, L, F, 1] float32 view of 50 %-overlapping snippets."""
    shift = snippet_length // 2
    n = (spectrogram.shape[0] - snippet_length) // shift + 1
    snippets = np.array([spectrogram[i * shift : i * shift + snippet_length] for i in range(n)])
    return snippets[..., np.newaxis]


def aggregate_predictions_ref(predictions: np.ndarray, n_frames: int, snippet_length: int, n_filters: int, num_labels: int):
    """predict.py:276-293 -- float64 overlay + overlap count + divide where count > 0."""
    shift, tpo, plen, _, total = snippet_geometry(n_frames, snippet_length, n_filters)
    agg = np.zeros((total, num_labels))
    cnt = np.zeros(total)
    for i, p in enumerate(predictions):
        start = i * (shift // tpo)
        agg[start : start + plen] += p
        cnt[start : start + plen] += 1
    valid = cnt > 0
    agg[valid] /= cnt[valid, np.newaxis]
    return agg, cnt


def find_consecutive_ones_ref(binary_vector: np.ndarray):
    """auxiliary.py:420-440 -- starts and INCLUSIVE stops of runs of ones."""
    diff = np.diff(binary_vector, prepend=0, append=0)
    return np.where(diff == 1)[0], np.where(diff == -1)[0] - 1


def compute_binary_predictions_ref(agg: np.ndarray, cnt: np.ndarray, calls: list[str], threshold: float = 0.5):
    """predict.py:298-317 -- threshold/max(overlap), strict >, per-label runs."""
    thr = threshold / np.max(cnt)
    binary = (agg > thr).astype(int)
    starts, stops, names = [], [], []
    for i, name in enumerate(calls):
        if sum(binary[:, i]) > 0:
            s, e = find_consecutive_ones_ref(binary[:, i])
            starts += list(s)
            stops += list(e)
            names += [name] * len(s)
    return starts, stops, names


def compute_labels_ref(starts, stops, names, tpo: int, label_suffix: str | None) -> pd.DataFrame:
    """predict.py:320-340 -- x tpo, suffix, sort by (start, stop, label)."""
    if (label_suffix is not None) & (label_suffix != ""):
        names = [n + label_suffix for n in names]
    return (
        pd.DataFrame({"start": np.asarray(starts) * tpo, "stop": np.asarray(stops) * tpo, "label": names})
        .sort_values(by=["start", "stop", "label"])
        .reset_index(drop=True)
    )


def labels_to_tsv_ref(labels: pd.DataFrame, delta_t: float) -> str:
    """predict.py:343-364,474-499 -- x delta_t, round(4), tab-separated with header."""
    labels = labels.copy()
    labels["start"] = labels["start"] * delta_t  # same values as the reference's .loc assignment
    labels["stop"] = labels["stop"] * delta_t
    return labels[["start", "stop", "label"]].round(4).to_csv(None, sep="\t", index=False)
