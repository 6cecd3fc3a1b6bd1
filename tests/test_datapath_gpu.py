"""Training-data path on the GPU: HBM-resident recording store + snippet-table gather and label down-sampling
(csrc/datapath.hip) against numpy restatements of DataLoader.__getitem__ / reshape_labels (io.py:101-147).  Bit-exact: the
gather is a copy, and the label mean is a sum of at most 16 small integers (exact in float32) followed by round-half-even."""

import numpy as np
import pandas as pd
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _recordings(tmp_path, n=3, W=20, L=4, seed=0):
    rng = np.random.default_rng(seed)
    dirs = []
    for r in range(n):
        T = int(rng.integers(300, 700))
        d = tmp_path / f"rec{r}"
        (d / "spectrogram").mkdir(parents=True)
        (d / "labels").mkdir()
        spec = rng.random((T, W), dtype=np.float32)
        lab = (rng.random((T, L)) < 0.4).astype(np.float32)
        lab[:, r % L][rng.random(T) < 0.3] = -1.0  # partially masked column: groups mixing -1 with 0/1
        lab[100:164, (r + 1) % L] = -1.0  # fully masked groups stay -1
        np.save(d / "spectrogram" / "spectrogram.npy", spec)
        np.save(d / "labels" / "labels.npy", lab)
        dirs.append(d)
    return dirs


def _table(dirs, rows, n, seed):
    rng = np.random.default_rng(seed)
    recs, starts = [], []
    for _ in range(n):
        d = dirs[int(rng.integers(len(dirs)))]
        T = np.load(d / "spectrogram" / "spectrogram.npy", mmap_mode="r").shape[0]
        s = int(rng.integers(0, T - rows + 1))
        recs.append(str(d))
        starts.append(s)
    starts[0] = 0  # first and last possible snippet of a recording
    T0 = np.load(dirs[0] / "spectrogram" / "spectrogram.npy", mmap_mode="r").shape[0]
    recs[0] = recs[1] = str(dirs[0])
    starts[1] = T0 - rows
    return pd.DataFrame({"recording_data_dir": recs, "row_start": starts, "row_stop": [s + rows for s in starts]})


def test_gather_and_label_downsampling_bit_exact(tmp_path):
    from oracle.train_ref import reshape_labels_ref
    from orcai_amd.datasets import SnippetTableDataset

    dirs = _recordings(tmp_path)
    rows, nf = 64, 4
    table = _table(dirs, rows, 37, seed=1)
    ds = SnippetTableDataset(table, nf, batch_size=8, shuffle=False)
    assert len(ds) == 4
    seen = 0
    for bi, (x, y) in enumerate(ds):
        assert x.shape == (8, rows, 20) and y.shape == (8, rows // 16, 4)
        for j in range(8):
            row = table.iloc[bi * 8 + j]
            spec = np.load(row["recording_data_dir"] + "/spectrogram/spectrogram.npy")
            lab = np.load(row["recording_data_dir"] + "/labels/labels.npy")
            assert np.array_equal(x[j].cpu().numpy(), spec[row["row_start"] : row["row_stop"]])
            assert np.array_equal(y[j].cpu().numpy(), reshape_labels_ref(lab[row["row_start"] : row["row_stop"]], nf))
            seen += 1
    assert seen == 32  # drop_remainder
    ymix = np.concatenate([y.cpu().numpy().ravel() for _, y in ds])
    assert set(np.unique(ymix)) <= {-1.0, 0.0, 1.0} and (ymix == -1.0).any()


def test_table_dataset_equals_materialised_dataset_and_trains(tmp_path):
    """The same snippets, once materialised to disk (SnippetDataset, the tf.data stand-in) and once gathered from the store, give
    identical batches in the same seeded order; a training step accepts the gathered batch directly."""
    from orcai_amd.architectures import ResNetLSTM
    from orcai_amd.datasets import SnippetDataset, SnippetTableDataset, reshape_labels, save_dataset
    from orcai_amd.training import Trainer

    dirs = _recordings(tmp_path, W=12, L=3, seed=3)
    rows, nf = 32, 2
    table = _table(dirs, rows, 40, seed=2)
    xs, ys = [], []
    for _, row in table.iterrows():
        spec = np.load(row["recording_data_dir"] + "/spectrogram/spectrogram.npy")
        lab = np.load(row["recording_data_dir"] + "/labels/labels.npy")
        xs.append(spec[row["row_start"] : row["row_stop"]])
        ys.append(reshape_labels(lab[row["row_start"] : row["row_stop"]], nf))
    save_dataset(np.stack(xs), np.stack(ys), tmp_path / "mat")
    a = SnippetDataset(tmp_path / "mat", 8, seed=[7, 1])
    b = SnippetTableDataset(table, nf, 8, seed=[7, 1])
    for (xa, ya), (xb, yb) in zip(a, b):
        assert torch.equal(xa, xb) and torch.equal(ya, yb)
    model = ResNetLSTM((32, 12, 1), 3, [10, 20], 3, 0.0, 64, seed=3)
    tr = Trainer(model, learning_rate=1e-3)
    xb, yb = next(iter(b))
    out = tr.train_step(xb.contiguous().view(-1), 32 * 12, 8, yb)
    assert np.isfinite(out["acc"].cpu().numpy()).all()
    with pytest.raises(ValueError):
        SnippetTableDataset(table, 6, 8)  # 32 rows are not divisible by 2**6
    bad = table.copy()
    bad.loc[0, "row_stop"] = 10**6
    bad.loc[0, "row_start"] = 10**6 - rows
    with pytest.raises(IndexError):
        SnippetTableDataset(bad, nf, 8)


def test_reference_made_snippet_tables_feed_training(tmp_path):
    """The training-data path from the reference's own tables to `orcai train`: tests/golden/snippet_tables.json holds the train / val / test
    tables the reference's create_tvt_snippet_tables wrote for the fixture recordings (make_golden.gen_snippet_tables); the product reads
    them through a dataset descriptor (nothing is materialised: the reference's create_tvt_data writes ~GBs of TF datasets here) and
    gathers its batches on the GPU from the recordings' arrays.  Rows gathered == rows the table names, bit for bit."""
    import importlib.util
    import json

    from click.testing import CliRunner

    from orcai_amd.cli import cli
    from orcai_amd.datasets import SnippetTableDataset, load_dataset
    from orcai_amd.io import read_json

    golden = __import__("pathlib").Path(__file__).resolve().parent / "golden"
    spec_ = importlib.util.spec_from_file_location("make_golden", golden / "make_golden.py")
    G = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(G)  # helpers only; the reference is not touched at import
    gold = json.loads((golden / "snippet_tables.json").read_text())
    root = tmp_path / "data"
    dirs = G.snippet_fixture_inputs(root)  # recA..recD: labels.npy + times.json, as the reference saw them
    W = 16
    rng = np.random.default_rng(5)
    for d in dirs:  # the spectrogram arrays the tables index (narrow: the model below takes any width)
        T = json.loads((d / "spectrogram" / "times.json").read_text())["length"]
        np.save(d / "spectrogram" / "spectrogram.npy", rng.random((T, W), dtype=np.float32))
    tvt = tmp_path / "tvt"
    tvt.mkdir()
    nf = 4
    for kind in ("train", "val", "test"):
        text = gold["files"][f"{kind}.csv.gz"].replace("<ROOT>", str(root))
        pd.read_csv(__import__("io").StringIO(text)).to_csv(tvt / f"{kind}.csv.gz", index=False)
        (tvt / f"{kind}_dataset").mkdir()
        (tvt / f"{kind}_dataset" / "snippet_table_dataset.json").write_text(json.dumps({"snippet_table": f"../{kind}.csv.gz", "n_filters": nf}))
    calls = gold["param"]["calls"]
    (tvt / "dataset_shapes.json").write_text(json.dumps({"spectrogram": [736, W, 1], "labels": [46, len(calls)]}))
    ds = load_dataset(tvt / "train_dataset", 4, seed=[7, 3])
    assert isinstance(ds, SnippetTableDataset) and len(ds) == gold["param"]["model"]["n_batch_train"]
    table = pd.read_csv(tvt / "train.csv.gz")
    plain = SnippetTableDataset(table, nf, 4, shuffle=False)
    from oracle.train_ref import reshape_labels_ref

    for bi, (x, y) in enumerate(plain):
        assert tuple(x.shape) == (4, 736, W) and tuple(y.shape) == (4, 46, len(calls))
        for j in range(4):
            row = table.iloc[bi * 4 + j]
            spec = np.load(row["recording_data_dir"] + "/spectrogram/spectrogram.npy", mmap_mode="r")
            lab = np.load(row["recording_data_dir"] + "/labels/labels.npy", mmap_mode="r")
            assert np.array_equal(x[j].cpu().numpy(), spec[row["row_start"] : row["row_stop"]])
            assert np.array_equal(y[j].cpu().numpy(), reshape_labels_ref(np.asarray(lab[row["row_start"] : row["row_stop"]], dtype=np.float32), nf))
    param = read_json(__import__("importlib.resources", fromlist=["files"]).files("orcai_amd.defaults").joinpath("default_orcai_parameter.json"))
    param.update({"name": "tiny", "seed": 3, "calls": calls})
    param["model"].update({"filters": [8, 12, 12, 12], "lstm_units": 64, "batch_size": 4, "n_batch_train": 6, "n_batch_val": 2, "n_batch_test": 2, "epochs": 1,
                           "call_weights": None})
    pfile = tmp_path / "param.json"
    pfile.write_text(json.dumps(param))
    out = tmp_path / "out"
    out.mkdir()
    res = CliRunner().invoke(cli, ["train", str(tvt), str(out), "-p", str(pfile), "-v", "0"], catch_exceptions=False)
    assert res.exit_code == 0, res.output
    hist = read_json(out / "tiny" / "training_history.json")
    assert len(hist["loss"]) == 1 and np.isfinite(hist["loss"][0]) and np.isfinite(hist["val_MBA"][0])
