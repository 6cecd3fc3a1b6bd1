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
