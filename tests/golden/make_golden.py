#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE's own code.

Run in the build container only (``/root/reference`` does not travel):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference's third-party dependencies that are absent from this image
(librosa, keras, tensorflow, zarr, humanize, ...) are replaced by empty stub
modules *before* import; only the reference's own numpy/pandas functions are
executed (preprocess_spectrogram, compute_aggregated_predictions,
compute_binary_predictions, compute_labels, find_consecutive_ones,
save_predictions, filter_predictions, and test.py's compute_confusion_table /
compute_misclassification_tables with the real scikit-learn) plus ``numpy.percentile`` itself for the
virtual-index table.  Outputs are data only -- no reference source or bytecode
is written anywhere.
"""

from __future__ import annotations

import io
import json
import sys
import types
import zlib
from pathlib import Path

import numpy as np
import pandas as pd

HERE = Path(__file__).resolve().parent
REFERENCE_SRC = Path("/root/reference/src")


class _Dummy:
    def __getattr__(self, k):
        return _Dummy()

    def __call__(self, *a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return _Dummy()

    def __mro_entries__(self, bases):
        return (object,)


class _Stub(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return _Dummy()


def import_reference():
    sys.dont_write_bytecode = True
    for name in ["humanize", "librosa", "keras", "tensorflow", "zarr", "tqdm.keras", "rich_click", "keras_tuner", "soundfile"]:
        sys.modules.setdefault(name, _Stub(name))
    sys.path.insert(0, str(REFERENCE_SRC))
    import orcAI.auxiliary as A
    import orcAI.predict as P
    import orcAI.spectrogram as S

    return S, P, A


SPEC_PARAM = {"sampling_rate": 48000, "nfft": 512, "n_overlap": 256, "freq_range": [0, 16000], "quantiles": [0.01, 0.999], "duration": 4}
CALLS = ["BR", "BUZZ", "HERDING", "PHS", "SS", "TAILSLAP", "WHISTLE"]
ORCAI_PARAM = {"name": "orcai-v1", "model": {"filters": [30, 40, 50, 60]}, "calls": CALLS, "spectrogram": SPEC_PARAM}
SHAPE = {"input_shape": [736, 171, 1], "num_labels": 7}


def synthetic_db(seed: int, T: int, kind: str) -> np.ndarray:
    """Seeded float32 dB-like array [257, T] in [-80, 0] (what calculate_spectrogram returns)."""
    rng = np.random.default_rng(seed)
    if kind == "smooth":
        x = -40.0 + 12.0 * rng.standard_normal((257, T))
    elif kind == "ties":  # coarse grid -> many exact ties, heavy mass on the floor
        x = np.round(-50.0 + 25.0 * rng.standard_normal((257, T)))
    elif kind == "constant":
        x = np.full((257, T), -33.25)
    else:
        raise ValueError(kind)
    x = np.clip(x, -80.0, 0.0).astype(np.float32)
    x[7, 3 % T] = 0.0  # the global max of an amplitude_to_db(ref=max) array is exactly 0
    return x


def crc(a: np.ndarray) -> int:
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def gen_preprocess(S):
    freqs = np.fft.rfftfreq(512, 1 / 48000)
    # small cases: full input and output stored
    for name, seed, T, kind in [("smooth_T300", 11, 300, "smooth"), ("ties_T200", 12, 200, "ties"), ("constant_T64", 13, 64, "constant")]:
        x = synthetic_db(seed, T, kind)
        with np.errstate(all="ignore"):
            y = S.preprocess_spectrogram(x.copy(), freqs, SPEC_PARAM)
        np.savez_compressed(HERE / f"preprocess_{name}.npz", db=x, out=np.ascontiguousarray(y), seed=seed, T=T, kind=kind)
        print("preprocess", name, y.shape, y.dtype, float(np.nanmin(y)), float(np.nanmax(y)))
    # large cases: input regenerated from the seed (crc pinned), output summarised
    summary = {}
    for name, seed, T, kind in [("smooth_T11251", 21, 11251, "smooth"), ("ties_T2000", 22, 2000, "ties")]:
        x = synthetic_db(seed, T, kind)
        y = np.ascontiguousarray(S.preprocess_spectrogram(x.copy(), freqs, SPEC_PARAM))
        rows = [0, 1, T // 2, T - 1]
        np.savez_compressed(HERE / f"preprocess_{name}.npz", rows=np.array(rows), out_rows=y[rows], seed=seed, T=T, kind=kind)
        summary[name] = {
            "seed": seed,
            "T": T,
            "kind": kind,
            "input_crc32": crc(x),
            "output_crc32": crc(y),
            "p_lo": float(np.percentile(x[0:171], 1.0, method="nearest")),
            "p_hi": float(np.percentile(x[0:171], 99.9, method="nearest")),
            "out_sum_f64": float(y.astype(np.float64).sum()),
        }
        print("preprocess", name, summary[name])
    # an input made of integer arithmetic only (tests/golden/portable_inputs.py): reproducible without trusting a Generator stream
    from portable_inputs import hashed_db

    x = hashed_db(23, 11251)
    y = np.ascontiguousarray(S.preprocess_spectrogram(x.copy(), freqs, SPEC_PARAM))
    summary["hashed_T11251"] = {"seed": 23, "T": 11251, "kind": "hashed", "input_crc32": crc(x), "output_crc32": crc(y),
                                "p_lo": float(np.percentile(x[0:171], 1.0, method="nearest")), "p_hi": float(np.percentile(x[0:171], 99.9, method="nearest")),
                                "out_sum_f64": float(y.astype(np.float64).sum())}
    print("preprocess hashed_T11251", summary["hashed_T11251"])
    (HERE / "preprocess_large.json").write_text(json.dumps(summary, indent=1))


def gen_virtual_index():
    """(n, q) -> index that numpy.percentile(method='nearest') selects on float32 data."""
    table = []
    for n in [171 * 736, 171 * 11251, 171 * 675001, 171 * 753665, (1 << 24) + 3, 1, 2, 171]:
        a = (np.arange(n, dtype=np.uint32) + np.uint32(0x3F800000)).view(np.float32)  # strictly increasing
        for q in [0.01, 0.999, 0.5, 0.0, 1.0]:
            v = np.percentile(a, 100 * q, method="nearest")
            idx = int(np.float32(v).view(np.uint32)) - 0x3F800000
            table.append({"n": n, "q": q, "index": idx})
        del a
    (HERE / "virtual_index.json").write_text(json.dumps(table, indent=1))
    print("virtual_index", table[:6])


class FakeModel:
    """Deterministic stand-in for keras.Model.predict: (n,736,171,1) -> (n,46,7) float32."""

    def predict(self, snippets, verbose=0):
        s = snippets[..., 0]
        n = s.shape[0]
        m = s.reshape(n, 46, 16, 171).mean(axis=(2,))  # (n,46,171)
        out = np.stack([m[:, :, 13 * l : 13 * l + 24].mean(axis=2) for l in range(7)], axis=2)
        return out.astype(np.float32)


def gen_aggregate(P):
    from orcAI.auxiliary import Messenger

    fake = FakeModel()
    for T in [736, 1103, 1104, 1471, 1472, 11251]:
        rng = np.random.default_rng(1000 + T)
        spec = rng.random((T, 171), dtype=np.float32)
        captured = {}

        class Capture(FakeModel):
            def predict(self, snippets, verbose=0):
                captured["snippet_sums"] = snippets.astype(np.float64).sum(axis=(1, 2, 3))
                captured["shape"] = snippets.shape
                captured["pred"] = super().predict(snippets, verbose)
                return captured["pred"]

        agg, cnt = P.compute_aggregated_predictions(Path("x.wav"), spec, Capture(), ORCAI_PARAM, SHAPE, msgr=Messenger(verbosity=0))
        np.savez_compressed(
            HERE / f"aggregate_T{T}.npz",
            T=T,
            seed=1000 + T,
            spec_crc32=crc(spec),
            snippet_shape=np.array(captured["shape"]),
            snippet_sums=captured["snippet_sums"],
            predictions=captured["pred"],
            aggregated=agg,
            overlap_count=cnt,
        )
        print("aggregate", T, captured["shape"], agg.shape, agg.dtype, {int(k): int(v) for k, v in zip(*np.unique(cnt, return_counts=True))})


def gen_labels(P):
    from orcAI.auxiliary import Messenger

    S_steps = 703
    rng = np.random.default_rng(77)
    agg = rng.random((S_steps, 7)) * 0.2  # all below 0.25
    cnt = np.concatenate([np.ones(23), 2 * np.ones(644), np.ones(23), np.zeros(13)])
    agg[0:5, 0] = 0.9  # run at index 0
    agg[690:703, 1] = 0.8  # run ending at S-1
    agg[100, 2] = 0.3  # single-step run
    agg[102, 2] = 0.3  # another single-step run one apart
    agg[200:210, 3] = 0.25  # exactly the threshold: strict > -> NOT a call
    agg[300:310, 4] = np.nextafter(0.25, 1.0)  # just above
    agg[300:320, 6] = 0.5  # same start as label 4, longer
    agg[300:310, 5] = 0.6  # same start and stop as label 4 -> sorted by label
    starts, stops, names = P.compute_binary_predictions(agg, cnt, CALLS, threshold=0.5)
    cases = {}
    for suffix in ["*", "", None, "_pred"]:
        df = P.compute_labels(starts, stops, names, 16, suffix)
        delta_t = 256 / 48000
        buf = io.StringIO()
        # save_predictions writes to a path; call it with a real temp file to capture bytes
        import tempfile

        with tempfile.TemporaryDirectory() as d:
            p = Path(d) / "o.txt"
            P.save_predictions(df.copy(), p, delta_t, msgr=Messenger(verbosity=0))
            text = p.read_text()
        cases[str(suffix)] = {
            "start": [int(v) for v in df["start"]],
            "stop": [int(v) for v in df["stop"]],
            "label": list(df["label"]),
            "tsv": text,
        }
    limits = {"default": [0.05, None], "BR": [None, 0.2], "WHISTLE": [0.1, 1.0], "SS": [0.3, None]}
    df = P.compute_labels(starts, stops, names, 16, "*")
    kept = P.filter_predictions(df.copy(), delta_t=256 / 48000, call_duration_limits=limits, label_suffix="*", msgr=Messenger(verbosity=0))
    out = {
        "row_starts": [int(v) for v in starts],
        "row_stops": [int(v) for v in stops],
        "label_names": names,
        "cases": cases,
        "filter_limits": limits,
        "filter_kept": {"start": [int(v) for v in kept["start"]], "stop": [int(v) for v in kept["stop"]], "label": list(kept["label"])},
    }
    np.savez_compressed(HERE / "labels_crafted.npz", aggregated=agg, overlap_count=cnt)
    (HERE / "labels_crafted.json").write_text(json.dumps(out, indent=1))
    print("labels", len(starts), cases["*"]["tsv"][:120].replace("\n", "|"))
    # empty case
    s0, e0, n0 = P.compute_binary_predictions(np.zeros((50, 7)), np.ones(50), CALLS)
    df0 = P.compute_labels(s0, e0, n0, 16, "*")
    (HERE / "labels_empty.json").write_text(json.dumps({"n": len(df0), "columns": list(df0.columns)}))


def gen_consecutive(A):
    cases = [[0, 1, 1, 0, 1], [1, 1, 1], [0, 0, 0], [1], [0], [1, 0, 1, 0, 1], [0, 1, 1, 1, 1, 0, 0, 1, 1]]
    out = []
    for c in cases:
        s, e = A.find_consecutive_ones(np.array(c))
        out.append({"input": c, "starts": [int(v) for v in s], "stops": [int(v) for v in e]})
    (HERE / "consecutive_ones.json").write_text(json.dumps({"mask_value": float(A.MASK_VALUE), "cases": out}, indent=1))


def gen_test_tables():
    """orcai test (test.py:37-225): confusion table and both misclassification tables on a seeded label / probability batch."""
    import orcAI.test as Tm

    rng = np.random.default_rng(4242)
    B, T, L = 40, 46, 7
    y_true = np.zeros((B, T, L), dtype=np.float32)
    hot = rng.integers(0, L + 3, size=(B, T))  # values >= L: no label in that row
    for l in range(L):
        y_true[..., l] = hot == l
    extra = rng.random((B, T)) < 0.03  # a few rows with two labels (dropped by the at-most-one-1 mask)
    y_true[..., 1][extra] = 1.0
    for b in range(0, B, 3):  # one label column masked in a third of the snippets
        y_true[b, :, (b // 3) % L] = -1.0
    y_pred = np.clip(0.15 + 0.7 * (y_true == 1) + 0.25 * rng.standard_normal((B, T, L)), 0.0, 1.0).astype(np.float32)
    y_pred[0, 0, 0] = 0.5  # exactly the threshold: counted as predicted (>=)
    conf = Tm.compute_confusion_table(y_true, y_pred, CALLS)
    stacked_true = Tm._stack_batch(y_true)
    stacked_pred = Tm._stack_batch((y_pred >= 0.5).astype(int))
    mis = Tm.compute_misclassification_tables(stacked_true, stacked_pred, "true", "pred", CALLS)
    # values as float64 arrays (exact); row / column labels in the json
    np.savez_compressed(HERE / "test_tables.npz", y_true=y_true, y_pred=y_pred, confusion=conf.to_numpy(dtype=np.float64),
                        **{"mis_" + k: v.to_numpy(dtype=np.float64) for k, v in mis.items()})
    out = {"confusion": {"index": list(conf.index), "columns": list(conf.columns)},
           "misclassification": {k: {"index": list(v.index), "columns": list(v.columns)} for k, v in mis.items()}}
    (HERE / "test_tables.json").write_text(json.dumps(out, indent=1))
    print("test tables", conf.shape, {k: v.shape for k, v in mis.items()})


def gen_label_raster():
    """labels.py:18-123 on a crafted annotation file: intervals touching frame times exactly (inclusive bounds), overlapping
    intervals of one label, a label present without annotations, masked labels, an original label missing from the equivalences."""
    import tempfile

    import orcAI.labels as Lb
    from orcAI.auxiliary import Messenger

    calls = ["BR", "BUZZ", "HERDING", "PHS", "SS"]
    times = {"min": 0.0, "max": 299 * 256 / 48000, "length": 300}
    t = np.linspace(times["min"], times["max"], times["length"])
    rows = [(float(t[10]), float(t[20]), "br1"), (float(t[18]) + 1e-9, float(t[25]) - 1e-9, "br2"), (0.5, 0.52, "buzz"), (float(t[100]), float(t[100]), "ss"),
            (1.2, 1.3, "unknown"), (float(t[298]), 5.0, "buzz"), (-1.0, float(t[0]), "ss")]
    eq = {"br1": "BR", "br2": "BR", "buzz": "BUZZ", "ss": "SS"}
    with tempfile.TemporaryDirectory() as d:
        d = Path(d)
        (d / "rec7" / "spectrogram").mkdir(parents=True)
        (d / "rec7" / "spectrogram" / "times.json").write_text(json.dumps(times))
        ann = d / "rec7.txt"
        ann.write_text("".join(f"{a!r}\t{b!r}\t{c}\n" for a, b, c in rows))
        arr, label_dict = Lb._convert_annotation(ann, d, calls, ["BR", "BUZZ", "SS", "PHS"], ["HERDING"], eq, Messenger(verbosity=0))
    np.savez_compressed(HERE / "labels_raster.npz", array=arr.to_numpy(dtype=np.float64))
    (HERE / "labels_raster.json").write_text(json.dumps({"calls": calls, "times": times, "rows": rows, "equivalences": eq, "present": ["BR", "BUZZ", "SS", "PHS"],
                                                          "masked": ["HERDING"], "columns": list(arr.columns), "label_dict": label_dict}, indent=1))
    print("label raster", arr.shape, arr.sum().to_dict())


def snippet_fixture_inputs(root: Path):
    """Deterministic recording data directories for the snippet-table fixtures (also rebuilt by the test): three recordings
    (640 s, 450 s, 150 s -- the last one shorter than a segment), per-frame labels with random call intervals, one masked label
    column, one recording without labels.  Label arrays go to labels.npy (the test reads these) AND are what the patched
    ``zarr.open`` hands to the reference."""
    calls = CALLS
    rng = np.random.default_rng(20250620)
    dirs = []
    for name, n_rows, with_labels in (("recA", 120000, True), ("recB", 84375, True), ("recC", 28125, True), ("recD", 60000, False)):
        d = root / name
        (d / "spectrogram").mkdir(parents=True)
        (d / "labels").mkdir(parents=True)
        times = {"min": 0.0, "max": (n_rows - 1) * 256 / 48000, "length": n_rows}
        (d / "spectrogram" / "times.json").write_text(json.dumps(times))
        if with_labels:
            lab = np.zeros((n_rows, len(calls)), dtype=np.int16)
            for c in range(len(calls)):
                for _ in range(int(rng.integers(3, 12))):
                    a = int(rng.integers(0, n_rows - 2000))
                    lab[a : a + int(rng.integers(50, 1500)), c] = 1
            if name == "recB":
                lab[:, 2] = -1  # HERDING cannot be annotated in this recording
            np.save(d / "labels" / "labels.npy", lab)
            (d / "labels" / "label_list.json").write_text(json.dumps({c: i for i, c in enumerate(calls)}))
        dirs.append(d)
    return dirs


SNIPPET_PARAM = {"name": "t", "seed": 42, "calls": CALLS,
                 "model": {"filters": [30, 40, 50, 60], "batch_size": 4, "n_batch_train": 6, "n_batch_val": 2, "n_batch_test": 2, "call_weights": None},
                 "snippets": {"segment_duration": 200, "snippets_per_sec": 0.25, "snippet_duration": 4, "fraction_removal": 0.5, "train": 0.8, "val": 0.1, "test": 0.1}}


def gen_snippet_tables():
    """snippets.py:26-556 -- _make_snippet_table, _compute_snippet_stats, _filter_snippet_table and the files written by
    create_snippet_table / create_tvt_snippet_tables, run by the reference itself with ``zarr.open`` patched to return the
    numpy label array (zarr and tensorflow are absent from this image)."""
    import gzip
    import tempfile
    import types as _types

    import orcAI.snippets as Sn
    from orcAI.auxiliary import Messenger

    Sn.zarr = _types.SimpleNamespace(open=lambda path, mode="r": np.load(Path(path).with_suffix(".npy"), mmap_mode="r") if Path(path).with_suffix(".npy").exists()
                                     else (_ for _ in ()).throw(FileNotFoundError(path)))
    out = {}
    with tempfile.TemporaryDirectory() as d:
        root = Path(d) / "data"
        dirs = snippet_fixture_inputs(root)
        rng = np.random.default_rng(seed=[1, SNIPPET_PARAM["seed"]])
        tables = []
        for rd in dirs:
            table, dur, nseg, rec, status = Sn._make_snippet_table(rd, SNIPPET_PARAM, rng=rng, msgr=Messenger(verbosity=0))
            out[f"status_{rec}"] = [float(dur), int(nseg), status]
            if table is not None:
                tables.append(table)
        allt = pd.concat(tables).reset_index(drop=True)
        stats = Sn._compute_snippet_stats(allt, for_calls=CALLS)
        filt = Sn._filter_snippet_table(allt, SNIPPET_PARAM, rng=np.random.default_rng(seed=[2, SNIPPET_PARAM["seed"]]), msgr=Messenger(verbosity=0))
        # the two public entry points, writing files
        rt = Path(d) / "recording_table.csv"
        pd.DataFrame({"recording": [x.name for x in dirs] + ["recE"], "base_dir_annotation": ["a", "a", "a", "a", np.nan]}).to_csv(rt, index=False)
        tv = Path(d) / "tvt"
        Sn.create_snippet_table(rt, root, tv, SNIPPET_PARAM, verbosity=0, msgr=Messenger(verbosity=0))
        Sn.create_tvt_snippet_tables(tv, None, SNIPPET_PARAM, create_unfiltered_test_snippets=True, n_unfiltered_test_snippets=5, verbosity=0, msgr=Messenger(verbosity=0))
        files_out = {}
        for f in sorted(tv.iterdir()):
            raw = gzip.decompress(f.read_bytes()).decode() if f.suffix == ".gz" else f.read_text()
            files_out[f.name] = raw.replace(str(root), "<ROOT>")
    num = ["row_start", "row_stop"] + CALLS
    np.savez_compressed(HERE / "snippet_tables.npz", all_numeric=allt[num].to_numpy(dtype=np.float64), filtered_numeric=filt[num].to_numpy(dtype=np.float64),
                        stats=stats.to_numpy(dtype=np.float64))
    (HERE / "snippet_tables.json").write_text(json.dumps({
        "param": SNIPPET_PARAM, "status": out, "all_recording": list(allt["recording"]), "all_data_type": list(allt["data_type"]),
        "filtered_recording": list(filt["recording"]), "filtered_data_type": list(filt["data_type"]), "stats_index": list(stats.index),
        "stats_columns": list(stats.columns), "files": files_out}, indent=1))
    print("snippet tables", allt.shape, filt.shape, {k: len(v) for k, v in files_out.items()})


def main_only_preprocess():
    """Regenerate the preprocess_* fixtures only (python make_golden.py preprocess)."""
    S, _, _ = import_reference()
    gen_preprocess(S)


def main():
    S, P, A = import_reference()
    gen_preprocess(S)
    gen_virtual_index()
    gen_aggregate(P)
    gen_labels(P)
    gen_consecutive(A)
    gen_test_tables()
    gen_label_raster()
    gen_snippet_tables()


if __name__ == "__main__":
    sys.path.insert(0, str(HERE))
    if sys.argv[1:] == ["preprocess"]:
        main_only_preprocess()
    else:
        main()
