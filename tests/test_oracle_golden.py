"""The CPU oracle against golden vectors produced by the reference's own code
(tests/golden/make_golden.py).  Bit-exact: these are integer / order-statistic /
f64-accumulate rows."""

import json
import zlib

import numpy as np
import pytest

from oracle import frontend_ref as F
from oracle import postprocess_ref as P

SPEC_PARAM = {"sampling_rate": 48000, "nfft": 512, "n_overlap": 256, "freq_range": [0, 16000], "quantiles": [0.01, 0.999], "duration": 4}
CALLS = ["BR", "BUZZ", "HERDING", "PHS", "SS", "TAILSLAP", "WHISTLE"]
FREQS = np.fft.rfftfreq(512, 1 / 48000)


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def synthetic_db(seed, T, kind):
    rng = np.random.default_rng(seed)
    if kind == "smooth":
        x = -40.0 + 12.0 * rng.standard_normal((257, T))
    elif kind == "ties":
        x = np.round(-50.0 + 25.0 * rng.standard_normal((257, T)))
    else:
        x = np.full((257, T), -33.25)
    x = np.clip(x, -80.0, 0.0).astype(np.float32)
    x[7, 3 % T] = 0.0
    return x


@pytest.mark.parametrize("name", ["smooth_T300", "ties_T200", "constant_T64"])
def test_preprocess_small_bit_exact(golden_dir, name):
    g = np.load(golden_dir / f"preprocess_{name}.npz")
    out = np.ascontiguousarray(F.preprocess_spectrogram_ref(g["db"], FREQS, SPEC_PARAM))
    assert out.dtype == np.float32 and out.shape == g["out"].shape
    assert np.array_equal(out, g["out"], equal_nan=True)


@pytest.mark.parametrize("name", ["smooth_T11251", "ties_T2000"])
def test_preprocess_large_bit_exact(golden_dir, name):
    meta = json.loads((golden_dir / "preprocess_large.json").read_text())[name]
    x = synthetic_db(meta["seed"], meta["T"], meta["kind"])
    if crc(x) != meta["input_crc32"]:
        pytest.skip("numpy Generator stream differs from the one the fixture was made with")
    out = np.ascontiguousarray(F.preprocess_spectrogram_ref(x, FREQS, SPEC_PARAM))
    assert crc(out) == meta["output_crc32"]
    g = np.load(golden_dir / f"preprocess_{name}.npz")
    assert np.array_equal(out[g["rows"]], g["out_rows"])


def test_crop_indices():
    assert F.crop_indices(FREQS, [0, 16000]) == (0, 171)
    assert F.crop_indices(FREQS, [500, 16000])[0] == 0  # first bin with f <= lo is always bin 0


def test_virtual_index_table(golden_dir):
    for row in json.loads((golden_dir / "virtual_index.json").read_text()):
        assert F.nearest_rank_index(row["n"], row["q"]) == row["index"], row


@pytest.mark.parametrize("T", [736, 1103, 1104, 1471, 1472, 11251])
def test_aggregate_bit_exact(golden_dir, T):
    g = np.load(golden_dir / f"aggregate_T{T}.npz")
    spec = np.random.default_rng(int(g["seed"])).random((T, 171), dtype=np.float32)
    if crc(spec) == int(g["spec_crc32"]):
        snippets = P.slice_snippets(spec, 736)
        assert list(snippets.shape) == list(g["snippet_shape"])
        assert np.array_equal(snippets.astype(np.float64).sum(axis=(1, 2, 3)), g["snippet_sums"])
    shift, tpo, plen, n, total = P.snippet_geometry(T, 736, 4)
    assert (shift, tpo, plen) == (368, 16, 46)
    assert n == g["predictions"].shape[0] and total == g["aggregated"].shape[0]
    agg, cnt = P.aggregate_predictions_ref(g["predictions"], T, 736, 4, 7)
    assert agg.dtype == np.float64
    assert np.array_equal(agg, g["aggregated"]) and np.array_equal(cnt, g["overlap_count"])


def test_labels_crafted(golden_dir):
    g = np.load(golden_dir / "labels_crafted.npz")
    j = json.loads((golden_dir / "labels_crafted.json").read_text())
    s, e, n = P.compute_binary_predictions_ref(g["aggregated"], g["overlap_count"], CALLS)
    assert [int(v) for v in s] == j["row_starts"] and [int(v) for v in e] == j["row_stops"] and n == j["label_names"]
    for suffix_key, case in j["cases"].items():
        suffix = None if suffix_key == "None" else suffix_key
        df = P.compute_labels_ref(s, e, n, 16, suffix)
        assert [int(v) for v in df["start"]] == case["start"]
        assert [int(v) for v in df["stop"]] == case["stop"]
        assert list(df["label"]) == case["label"]
        assert P.labels_to_tsv_ref(df, 256 / 48000) == case["tsv"]


def test_consecutive_ones(golden_dir):
    j = json.loads((golden_dir / "consecutive_ones.json").read_text())
    for c in j["cases"]:
        s, e = P.find_consecutive_ones_ref(np.array(c["input"]))
        assert list(s) == c["starts"] and list(e) == c["stops"]


def test_orcai_test_tables_match_reference(golden_dir):
    """test.py:37-225 (confusion table, both misclassification tables) against tables the reference's own functions produced
    (scikit-learn confusion_matrix included) on a seeded batch with masked columns, double labels and a probability of exactly 0.5."""
    from orcai_amd.test import _stack_batch, compute_confusion_table, compute_misclassification_tables

    with np.load(golden_dir / "test_tables.npz") as z:
        g = {k: z[k] for k in z.files}
    want = json.loads((golden_dir / "test_tables.json").read_text())
    conf = compute_confusion_table(g["y_true"], g["y_pred"], CALLS)
    assert list(conf.index) == want["confusion"]["index"] and list(conf.columns) == want["confusion"]["columns"]
    assert np.array_equal(conf.to_numpy(dtype=np.float64), g["confusion"], equal_nan=True)
    mis = compute_misclassification_tables(_stack_batch(g["y_true"]), _stack_batch((g["y_pred"] >= 0.5).astype(int)), "true", "pred", CALLS)
    assert sorted(mis) == sorted(want["misclassification"])
    for key, table in mis.items():
        assert list(table.index) == want["misclassification"][key]["index"] and list(table.columns) == want["misclassification"][key]["columns"]
        assert np.array_equal(table.to_numpy(dtype=np.float64), g["mis_" + key], equal_nan=True), key


def test_label_rasterisation_matches_reference(golden_dir, tmp_path):
    """labels.py:18-123 (`t_vec >= start & t_vec <= stop`, masked columns, call equivalences) against the array the reference's
    own `_convert_annotation` produced; then the table driver writes the arrays the GPU data path loads."""
    from orcai_amd.labels import _convert_annotation, create_label_arrays

    g = json.loads((golden_dir / "labels_raster.json").read_text())
    want = np.load(golden_dir / "labels_raster.npz")["array"]
    (tmp_path / "rec7" / "spectrogram").mkdir(parents=True)
    (tmp_path / "rec7" / "spectrogram" / "times.json").write_text(json.dumps(g["times"]))
    ann = tmp_path / "rec7.txt"
    ann.write_text("".join(f"{a!r}\t{b!r}\t{c}\n" for a, b, c in g["rows"]))
    arr, label_dict = _convert_annotation(ann, tmp_path, g["calls"], g["present"], g["masked"], g["equivalences"])
    assert list(arr.columns) == g["columns"] and label_dict == g["label_dict"]
    assert np.array_equal(arr.to_numpy(dtype=np.float64), want)
    with pytest.raises(KeyError):  # the reference only creates the label column when equivalences are given (labels.py:66-77)
        _convert_annotation(ann, tmp_path, g["calls"], g["present"], g["masked"], None)
    import pandas as pd

    table = pd.DataFrame({"recording": ["rec7"], "base_dir_annotation": [str(tmp_path)], "rel_annotation_path": ["rec7.txt"],
                          **{c: [c in g["present"]] for c in g["calls"]}})
    table.to_csv(tmp_path / "table.csv", index=False)
    create_label_arrays(tmp_path / "table.csv", tmp_path, orcai_parameter={"calls": g["calls"]}, call_equivalences=g["equivalences"], verbosity=0)
    saved = np.load(tmp_path / "rec7" / "labels" / "labels.npy")
    assert saved.dtype == np.float32 and np.array_equal(saved.astype(np.float64), want)
    assert json.loads((tmp_path / "rec7" / "labels" / "label_list.json").read_text()) == g["label_dict"]


def test_stft_oracle_agrees_with_independent_implementations():
    """The STFT stage has no reference fixture (librosa is absent: parity unpinned).  As a cross-check, the numpy restatement is
    compared with two independent implementations of the same definition (centred, zero padding, periodic Hann, hop 256):
    torch.stft in float64 and scipy.signal.stft rescaled; plus the frame-count formula T = 1 + N // hop (spectrogram.py:34-39)."""
    import scipy.signal
    import torch

    rng = np.random.default_rng(3)
    for n in (256 * 40, 256 * 40 + 1, 256 * 40 + 255, 700, 100):
        y = (rng.standard_normal(n) * 0.2).astype(np.float32)
        S = F.stft_ref(y)
        assert S.shape == (257, 1 + n // 256) and S.dtype == np.complex64
        win = torch.hann_window(512, periodic=True, dtype=torch.float64)
        St = torch.stft(torch.from_numpy(y.astype(np.float64)), 512, hop_length=256, window=win, center=True, pad_mode="constant", return_complex=True).numpy()
        assert St.shape == S.shape
        assert np.abs(S - St).max() <= 1e-5 * max(1.0, np.abs(St).max())
        if n >= 512:
            _, _, Ss = scipy.signal.stft(y.astype(np.float64), window="hann", nperseg=512, noverlap=256, boundary="zeros", padded=False)
            Ss = Ss * scipy.signal.get_window("hann", 512).sum()  # scipy normalises by the window sum
            k = min(Ss.shape[1], S.shape[1])
            assert np.abs(S[:, :k] - Ss[:, :k]).max() <= 1e-5 * max(1.0, np.abs(Ss).max())


def test_lstm_oracle_agrees_with_torch_nn_lstm():
    """The Keras layers have no fixture here (parity unpinned).  Cross-check of the oracle's Bidirectional(LSTM) restatement
    (gate order i, f, c, o; one bias vector; backward direction returned in input time order; concat [fwd, bwd],
    architectures.py:210-229) against torch.nn.LSTM, an independent implementation with the same gate order."""
    import torch

    from oracle import model_ref as M

    rng = np.random.default_rng(9)
    B, T, Fin, u = 3, 11, 20, 16
    p = {}
    for d in ("fwd", "bwd"):
        p[f"l/{d}/kernel"] = (rng.standard_normal((Fin, 4 * u)) * 0.3).astype(np.float32)
        p[f"l/{d}/recurrent"] = (rng.standard_normal((u, 4 * u)) * 0.3).astype(np.float32)
        p[f"l/{d}/bias"] = (rng.standard_normal(4 * u) * 0.1).astype(np.float32)
    x = rng.standard_normal((B, T, Fin)).astype(np.float32)
    got = M._bilstm(torch.tensor(x, dtype=torch.float64), p, "l", torch.float64).numpy()
    ref = torch.nn.LSTM(Fin, u, batch_first=True, bidirectional=True).double()
    with torch.no_grad():
        for d, suffix in (("fwd", ""), ("bwd", "_reverse")):
            getattr(ref, "weight_ih_l0" + suffix).copy_(torch.tensor(p[f"l/{d}/kernel"].T, dtype=torch.float64))
            getattr(ref, "weight_hh_l0" + suffix).copy_(torch.tensor(p[f"l/{d}/recurrent"].T, dtype=torch.float64))
            getattr(ref, "bias_ih_l0" + suffix).copy_(torch.tensor(p[f"l/{d}/bias"], dtype=torch.float64))
            getattr(ref, "bias_hh_l0" + suffix).zero_()
        want = ref(torch.tensor(x, dtype=torch.float64))[0].numpy()
    assert got.shape == want.shape == (B, T, 2 * u)
    assert np.abs(got - want).max() <= 1e-12


@pytest.mark.parametrize("cfg", [
    dict(input_shape=(16, 9, 1), filters=(5, 6), kernel_size=3, lstm_units=8, num_labels=3),
    dict(input_shape=(8, 12, 1), filters=(7,), kernel_size=5, lstm_units=6, num_labels=2),
    dict(input_shape=(24, 7, 1), filters=(4, 5, 6), kernel_size=7, lstm_units=4, num_labels=4),
])
def test_two_independent_model_restatements_agree(cfg):
    """the torch-functional oracle (model_ref) and the explicit-loop numpy restatement (model_ref_loops, written from the
    Keras / TF documentation without a convolution library) agree on ResNetLSTM and ResNet1DConv for odd widths (asymmetric "same"
    pooling pads), kernel sizes 3 / 5 / 7 and 1-3 blocks, in float64."""
    import torch

    from oracle import model_ref as M
    from oracle import model_ref_loops as Lp

    p = M.random_params(seed=4, **cfg)
    rng = np.random.default_rng(4)
    x = rng.random((2, *cfg["input_shape"]), dtype=np.float32)
    ref = M.forward_ref(p, x, dtype=torch.float64)
    for b in range(2):
        assert np.abs(Lp.forward_one(p, x[b]) - ref[b]).max() <= 1e-10
    q = {k: v for k, v in p.items() if not k.startswith(("lstm", "dense", "bn_d"))}
    q["conv1d/kernel"] = (0.2 * rng.standard_normal((36, 36, cfg["num_labels"]))).astype(np.float32)
    q["conv1d/bias"] = (0.1 * rng.standard_normal(cfg["num_labels"])).astype(np.float32)
    ref1 = M.forward_ref_1dconv(q, x, dtype=torch.float64)
    for b in range(2):
        assert np.abs(Lp.forward_one_1dconv(q, x[b]) - ref1[b]).max() <= 1e-10


def _golden_generator_module(golden_dir):
    import importlib.util

    spec = importlib.util.spec_from_file_location("make_golden", golden_dir / "make_golden.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)  # defines helpers only; the reference is not touched at import
    return mod


def test_reference_snippet_tables_are_valid_product_inputs(golden_dir):
    """tests/golden/snippet_tables.* were written by the reference's own snippets.py (create_snippet_table /
    create_tvt_snippet_tables; make_golden.gen_snippet_tables).  The product does not rebuild those tables (out of
    scope); it CONSUMES them (orcai_amd.datasets.SnippetTableDataset, GPU test in test_datapath_gpu.py).  Here: every row of the
    reference-made train / val / test tables has the length orcai_amd.datasets.snippet_rows derives for the same parameters, and
    stays inside the fixture recordings."""
    import io
    import json

    import pandas as pd

    from orcai_amd.datasets import snippet_rows

    G = _golden_generator_module(golden_dir)
    gold = json.loads((golden_dir / "snippet_tables.json").read_text())
    param = gold["param"]
    lengths = {}
    for name in ("train.csv.gz", "val.csv.gz", "test.csv.gz", "all_snippets.csv.gz"):
        t = pd.read_csv(io.StringIO(gold["files"][name]))
        assert len(t) > 0
        lengths[name] = t
        assert ((t["row_stop"] - t["row_start"]) == 736).all()
        assert (t["row_start"] >= -1).all()
    meta = {"min": 0.0, "max": 11250 * 256 / 48000, "length": 11251}
    a, b = snippet_rows(meta, 1.0, param["snippets"]["snippet_duration"], len(param["model"]["filters"]))
    assert b - a == 736
    assert len(lengths["train.csv.gz"]) == param["model"]["batch_size"] * param["model"]["n_batch_train"]
    assert hasattr(G, "snippet_fixture_inputs")  # the generator of the recordings these tables index (used by the GPU test)



def test_training_oracle_forced_branches_reproduce_its_own_gradient():
    """oracle.train_ref with `forced` branches (ReLU masks, max-pool window elements): forcing the branches an unforced forward records gives
    that forward's loss and gradients again (the mechanism tests/test_half_gpu.py uses to compare the f16 path with the float64 gradient of
    the SAME piecewise-linear branch), and forcing OTHER branches changes them."""
    from oracle import model_ref as M
    from oracle import train_ref as T

    cfg = dict(input_shape=(32, 12, 1), filters=(10, 20), kernel_size=3, lstm_units=8, num_labels=3)
    p = M.calibrated_params(seed=3, **cfg)
    rng = np.random.default_rng(3)
    B, steps = 2, 32 // 4
    x = rng.random((B, 32, 12, 1), dtype=np.float32)
    y = (rng.random((B, steps, 3)) > 0.5).astype(np.float32)
    masks = {k: (rng.random((B, steps, d)) > 0.5).astype(np.float32) for k, d in (("drop1", 16), ("drop2", 16), ("drop3", 128))}
    rec = {}
    base = T.loss_and_grads(p, x, y, masks, 0.5, forced_np={"record": rec})
    assert {"relu/bn0", "relu/b1/in", "relu/b2/bn_a", "relu/bn_f", "relu/dense1", "pool/b1", "pool/b2"} <= set(rec)
    forced = {k: v.numpy() for k, v in rec.items()}
    again = T.loss_and_grads(p, x, y, masks, 0.5, forced_np=forced)
    assert abs(again["loss"] - base["loss"]) <= 1e-12
    for k, g in base["grads"].items():
        assert np.abs(again["grads"][k] - g).max() <= 1e-10 * max(1.0, np.abs(g).max()), k
    forced["relu/b1/bn_a"] = 1.0 - forced["relu/b1/bn_a"]
    other = T.loss_and_grads(p, x, y, masks, 0.5, forced_np=forced)
    assert np.abs(other["grads"]["b1/sep_a/pointwise"] - base["grads"]["b1/sep_a/pointwise"]).max() > 1e-6
