"""Randomised hyper-parameter / shape sweep (seeded): odd and even heights and widths, widths around the 60/62/64-column
window sizes, filter counts that are not multiples of 4 or 16, kernel sizes 3/5/7, batch sizes around the 16-snippet LSTM tile.
Inference forward and one full training step are compared with the CPU oracle; this is the safety net for indexing bugs in
the padded-plane kernels that fixed test shapes would not reach."""

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import model_ref as M  # noqa: E402
from oracle import train_ref as T  # noqa: E402


def _random_cfg(rng, train: bool):
    nb = int(rng.integers(1, 4 if train else 5))
    k = int(rng.choice([3, 3, 5, 7] if not train else [3, 3, 5]))
    H = int(rng.integers(2, 7)) * 2**nb  # the reference requires H divisible by 2**n_filters (labels are averaged over 2**n rows)
    W = int(rng.choice([5, 11, 12, 20, 31, 59, 60, 61, 62, 63, 64, 65, 121, 124]))
    if train:
        W = min(W, 65)
    filters = tuple(int(rng.integers(3, 64 if not train else 40)) for _ in range(nb))
    return dict(input_shape=(H, W, 1), filters=filters, kernel_size=k, lstm_units=int(rng.choice([64, 128])), num_labels=int(rng.integers(1, 9)))


@pytest.mark.parametrize("seed", range(10))
def test_random_inference_configs(seed):
    from orcai_amd.architectures import ResNetLSTM

    rng = np.random.default_rng(1000 + seed)
    cfg = _random_cfg(rng, train=False)
    p = M.calibrated_params(seed=seed, **cfg)
    model = ResNetLSTM(cfg["input_shape"], cfg["num_labels"], list(cfg["filters"]), cfg["kernel_size"], 0.0, cfg["lstm_units"])
    model.set_weights_dict(p)
    B = int(rng.choice([1, 2, 15, 16, 17, 33]))
    x = rng.random((B, *cfg["input_shape"]), dtype=np.float32)
    ref = M.forward_ref(p, x)
    out = model.predict(x, batch_size=int(rng.choice([B, max(1, B // 2), 7])))
    assert out.shape == ref.shape, cfg
    assert np.abs(out - ref).max() <= 1e-5, (cfg, B, float(np.abs(out - ref).max()))


@pytest.mark.parametrize("seed", range(6))
def test_random_training_configs(seed):
    from orcai_amd.architectures import ResNetLSTM
    from orcai_amd.training import Trainer

    rng = np.random.default_rng(2000 + seed)
    cfg = _random_cfg(rng, train=True)
    p = M.calibrated_params(seed=seed, **cfg)
    for k in p:
        if k.endswith(("gamma", "beta")):
            p[k] = (p[k] + 0.2 * rng.standard_normal(p[k].shape)).astype(np.float32)
    H, W, _ = cfg["input_shape"]
    steps = H // 2 ** len(cfg["filters"])
    L, u = cfg["num_labels"], cfg["lstm_units"]
    B = int(rng.choice([2, 3, 5]))
    x = rng.random((B, H, W, 1), dtype=np.float32)
    y = (rng.random((B, steps, L)) > 0.5).astype(np.float32)
    y[0, :, 0] = -1.0
    ref = T.loss_and_grads(p, x, y, None, 0.0)
    model = ResNetLSTM(cfg["input_shape"], L, list(cfg["filters"]), cfg["kernel_size"], 0.0, u)
    model.set_weights_dict(p)
    tr = Trainer(model, learning_rate=1e-3)
    out = tr.forward_backward(torch.from_numpy(np.ascontiguousarray(x[..., 0])).cuda().view(-1), H * W, B, torch.from_numpy(y).cuda(), masks=None)
    acc = out["acc"].cpu().numpy()
    assert np.abs(out["probs"].cpu().numpy() - ref["probs"]).max() <= 5e-6, cfg
    assert abs(acc[0] / acc[1] + acc[3] - ref["loss"]) <= 2e-6 * max(1.0, abs(ref["loss"])), cfg
    bad = {}
    for name, g in ref["grads"].items():
        got = tr.P.G(name).cpu().numpy()
        zero_mean_bias = name.endswith("/bias") and not name.startswith(("dense2", "lstm", "dense1")) and "res" not in name
        scale = max(1e-3, float(np.abs(g).max())) if not zero_mean_bias else 1.0
        err = float(np.abs(got - g).max()) / scale
        if err > (5e-4 if not zero_mean_bias else 1e-4):
            bad[name] = err
    assert not bad, (cfg, B, bad)
