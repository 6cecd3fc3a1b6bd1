"""GPU parity of the training-mode head (BN+ReLU of the final conv, 2 BiLSTM + dropout, Dense+BN+dropout, Dense+sigmoid,
masked BCE + L2) forward AND backward against torch autograd on the CPU oracle (float64).

Tolerance: fp32 kernels vs fp64 autograd: probabilities 2e-6; loss 1e-6 relative; every gradient tensor
max|delta| <= 2e-5 * max(1e-3, max|ref|) (relative to the tensor's scale)."""

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import model_ref as M  # noqa: E402
from oracle import train_ref as T  # noqa: E402


def _setup(seed, n, units, labels=3, rate=0.5):
    from orcai_amd.architectures import ResNetLSTM
    from orcai_amd.training import FlatParams, HeadTrainer

    cfg = dict(input_shape=(32, 12, 1), filters=(10, 20), kernel_size=3, lstm_units=units, num_labels=labels)
    p = M.calibrated_params(seed=seed, **cfg)
    rng = np.random.default_rng(seed)
    for k in p:  # make every head parameter non-trivial
        if k.startswith(("bn_f", "bn_d")) and k.endswith(("gamma", "beta")):
            p[k] = (p[k] + 0.3 * rng.standard_normal(p[k].shape)).astype(np.float32)
    model = ResNetLSTM(cfg["input_shape"], labels, list(cfg["filters"]), 3, rate, units)
    model.set_weights_dict(p)
    P = FlatParams(model, torch.device("cuda"))
    return model, p, P, HeadTrainer(model, P), rng


def _oracle_head(p, featv, y, masks, rate, W):
    dt = torch.float64
    tp = {k: torch.tensor(v, dtype=dt, requires_grad=T.is_trainable(k)) for k, v in p.items()}
    f = torch.tensor(featv, dtype=dt, requires_grad=True)
    n, Tn, cols = featv.shape
    stats = {}
    x = T._bn_train(f.view(n, Tn, W, cols // W), tp, "bn_f", stats, axis=3)
    x = torch.relu(x).reshape(n, Tn, cols)
    keep = 1 - rate
    x = T._bilstm(x, tp, "lstm1")
    if masks:
        x = x * torch.tensor(masks["drop1"], dtype=dt) / keep
    x = T._bilstm(x, tp, "lstm2")
    if masks:
        x = x * torch.tensor(masks["drop2"], dtype=dt) / keep
    x = torch.relu(x @ tp["dense1/kernel"] + tp["dense1/bias"])
    x = T._bn_train(x, tp, "bn_d", stats, axis=2)
    if masks:
        x = x * torch.tensor(masks["drop3"], dtype=dt) / keep
    probs = torch.sigmoid(x @ tp["dense2/kernel"] + tp["dense2/bias"])
    bce = T.masked_bce(torch.tensor(y, dtype=dt), probs)
    loss = bce + T.l2_penalty(tp)
    loss.backward()
    grads = {k: v.grad.numpy() for k, v in tp.items() if v.grad is not None}
    return probs.detach().numpy(), float(bce.detach()), float(loss.detach()), grads, f.grad.numpy(), {k: v.detach().numpy() for k, v in stats.items()}


@pytest.mark.parametrize("n,units,use_dropout", [(5, 64, True), (18, 128, True), (3, 64, False)])
def test_head_forward_backward_vs_autograd(n, units, use_dropout):
    model, p, P, head, rng = _setup(3 + n, n, units)
    Tn, W, C = 8, 3, 36
    featv = (rng.standard_normal((n, Tn, W * C)) * 1.5 + 0.3).astype(np.float32)
    y = (rng.random((n, Tn, 3)) > 0.5).astype(np.float32)
    y[0, :, 1] = -1.0
    y[n - 1, 3:, :] = -1.0
    rate = 0.5
    masks = None
    if use_dropout:
        masks = {k: (rng.random((n, Tn, d)) > rate).astype(np.float32) for k, d in (("drop1", 2 * units), ("drop2", 2 * units), ("drop3", 128))}
    dmasks = None if masks is None else {k: torch.from_numpy(v).cuda() for k, v in masks.items()}
    probs = head.forward(torch.from_numpy(featv).cuda(), dmasks, rate)
    out = head.loss_and_backward(torch.from_numpy(y).cuda())
    acc = out["acc"].cpu().numpy()
    ref_p, ref_bce, ref_loss, ref_g, ref_df, ref_stats = _oracle_head(p, featv, y, masks, rate, W)
    assert np.abs(probs.cpu().numpy() - ref_p).max() <= 2e-6
    bce = acc[0] / acc[1]
    assert abs(bce - ref_bce) <= 1e-6 * max(1.0, abs(ref_bce))
    assert abs((bce + acc[3]) - ref_loss) <= 1e-6 * max(1.0, abs(ref_loss))
    assert acc[1] == float((y != -1).sum())
    assert acc[2] == float((((ref_p > 0.5).astype(np.float32) == y) & (y != -1)).sum())
    worst = {}
    for name, g in ref_g.items():
        if name not in P.offsets or not name.startswith(("lstm", "dense", "bn_d", "bn_f")):
            continue
        got = P.G(name).cpu().numpy()
        scale = max(1e-3, float(np.abs(g).max()))
        worst[name] = float(np.abs(got - g).max()) / scale
    assert len(worst) == 20
    bad = {k: v for k, v in worst.items() if v > 2e-5}
    assert not bad, bad
    df = out["dfeatv"].cpu().numpy()
    assert np.abs(df - ref_df).max() <= 2e-5 * max(1e-3, float(np.abs(ref_df).max()))
    head.update_moving_stats()
    for k in ("bn_f/mean", "bn_f/var", "bn_d/mean", "bn_d/var"):
        assert np.abs(P.stats[k].cpu().numpy() - ref_stats[k]).max() <= 1e-5 * max(1.0, float(np.abs(ref_stats[k]).max()))


def test_adam_matches_keras_form():
    from orcai_amd.training import FlatParams, adam_step

    model, p, P, head, rng = _setup(11, 2, 64)
    w0 = P.w.cpu().numpy().astype(np.float64)
    m = np.zeros_like(w0)
    v = np.zeros_like(w0)
    w = w0.copy()
    for step in range(1, 4):
        g = rng.standard_normal(w0.shape) * 10.0 ** rng.integers(-6, 1, size=w0.shape)
        P.g.copy_(torch.from_numpy(g.astype(np.float32)))
        adam_step(P, 1e-4, step)
        w, m, v = T.adam_step_ref(w, g.astype(np.float32).astype(np.float64), m, v, step, 1e-4)
    assert np.abs(P.w.cpu().numpy() - w).max() <= 2e-7
    assert np.abs(P.m.cpu().numpy() - m).max() <= 1e-6 * np.abs(m).max()


def test_dropout_mask_statistics():
    from orcai_amd import _native as N

    n = 1 << 20
    mask = torch.empty(n, device="cuda")
    N.check(N.lib().orcai_dropout_mask(mask.data_ptr(), n, 1234, 0.5, N.stream_ptr()), "dropout_mask")
    m = mask.cpu().numpy()
    assert set(np.unique(m)) == {0.0, 1.0} and abs(m.mean() - 0.5) < 5e-3
    mask2 = torch.empty(n, device="cuda")
    N.check(N.lib().orcai_dropout_mask(mask2.data_ptr(), n, 1234, 0.5, N.stream_ptr()), "dropout_mask")
    assert torch.equal(mask, mask2)  # counter based: reproducible
    N.check(N.lib().orcai_dropout_mask(mask2.data_ptr(), n, 1235, 0.7, N.stream_ptr()), "dropout_mask")
    assert abs(mask2.mean().item() - 0.7) < 5e-3 and not torch.equal(mask, mask2)


@pytest.mark.parametrize("M_,cols,C", [(2944, 396, 36), (77, 128, 128), (600, 1100, 50), (33, 36, 36), (40, 2304, 36)])
def test_row_batchnorm_and_column_sum_kernels_vs_float64(M_, cols, C):
    """The row-tensor kernels of the head on their own (round 4: re-parallelised -- rows over threads without index division, 2-D apply grids, 32-column
    workgroups for the column sums, row slabs with an ordered fold up to 2 048 columns and the one-workgroup-per-channel kernels beyond): statistics, apply, backward sums + apply and column sums against float64 numpy on shapes with ragged column blocks."""
    from orcai_amd import _native as N

    lib, st = N.lib(), N.stream_ptr()
    rng = np.random.default_rng(M_ + cols)
    x = (rng.standard_normal((M_, cols)) * 1.5 + 0.3).astype(np.float32)
    dy = rng.standard_normal((M_, cols)).astype(np.float32)
    gamma, beta = (1 + 0.3 * rng.standard_normal(C)).astype(np.float32), (0.2 * rng.standard_normal(C)).astype(np.float32)
    dev = lambda a: torch.from_numpy(a).cuda()  # noqa: E731
    xd, dyd, gd, bd = dev(x), dev(dy), dev(gamma), dev(beta)
    mean, var = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    N.check(lib.orcai_bn_rows_stats(N.ptr(xd), M_, cols, C, N.ptr(mean), N.ptr(var), st), "bn_rows_stats")
    xr = x.astype(np.float64).reshape(M_, cols // C, C)
    mu, vv = xr.mean(axis=(0, 1)), xr.var(axis=(0, 1))
    torch.cuda.synchronize()
    assert np.abs(mean.cpu().numpy() - mu).max() <= 2e-6 and np.abs(var.cpu().numpy() - vv).max() <= 5e-6
    eps = 1e-3
    for relu in (0, 1):
        y = torch.empty_like(xd)
        N.check(lib.orcai_bn_rows_apply(N.ptr(xd), M_, cols, C, N.ptr(mean), N.ptr(var), N.ptr(gd), N.ptr(bd), eps, relu, N.ptr(y), st), "bn_rows_apply")
        m32, v32 = mean.cpu().numpy().astype(np.float64), var.cpu().numpy().astype(np.float64)
        inv = 1.0 / np.sqrt(v32 + eps)
        xh = (xr - m32) * inv
        yr = xh * gamma + beta
        want = np.maximum(yr, 0) if relu else yr
        assert np.abs(y.cpu().numpy().reshape(xr.shape) - want).max() <= 5e-6 * max(1.0, np.abs(want).max())
        dbeta, dgamma, dx = torch.empty(C, device="cuda"), torch.empty(C, device="cuda"), torch.empty_like(xd)
        N.check(lib.orcai_bn_rows_bwd(N.ptr(dyd), N.ptr(xd), M_, cols, C, N.ptr(mean), N.ptr(var), N.ptr(gd), N.ptr(bd), eps, relu, N.ptr(dbeta), N.ptr(dgamma), N.ptr(dx), st), "bn_rows_bwd")
        got_y = y.cpu().numpy().reshape(xr.shape)
        de = dy.astype(np.float64).reshape(xr.shape) * ((got_y > 0) if relu else 1.0)  # the mask the kernel's own forward value gives
        cnt = M_ * (cols // C)
        db, dg = de.sum(axis=(0, 1)), (de * xh).sum(axis=(0, 1))
        dxr = gamma * inv * (de - db / cnt - xh * dg / cnt)
        torch.cuda.synchronize()
        assert np.abs(dbeta.cpu().numpy() - db).max() <= 2e-5 * max(1.0, np.abs(db).max()) and np.abs(dgamma.cpu().numpy() - dg).max() <= 2e-5 * max(1.0, np.abs(dg).max())
        assert np.abs(dx.cpu().numpy().reshape(xr.shape) - dxr).max() <= 2e-5 * max(1.0, np.abs(dxr).max())
    out = torch.full((cols,), 2.0, device="cuda")
    N.check(lib.orcai_colsum(N.ptr(dyd), M_, cols, N.ptr(out), 0, st), "colsum")
    N.check(lib.orcai_colsum(N.ptr(dyd), M_, cols, N.ptr(out), 1, st), "colsum")
    want = 2 * dy.astype(np.float64).sum(axis=0)
    assert np.abs(out.cpu().numpy() - want).max() <= 1e-5 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("M_,K,N_", [(2944, 128, 7), (33, 128, 3), (100, 6, 8), (7, 36, 1)])
def test_dense_sigmoid_rows_vs_float64(M_, K, N_):
    """orcai_dense_sigmoid: the eight-lanes-per-row kernel (K a multiple of 4) and the one-thread-per-row kernel it falls back to, against float64."""
    from orcai_amd import _native as N

    lib, st = N.lib(), N.stream_ptr()
    rng = np.random.default_rng(K + N_)
    x, w, b = rng.standard_normal((M_, K)).astype(np.float32), (rng.standard_normal((K, N_)) / np.sqrt(K)).astype(np.float32), rng.standard_normal(N_).astype(np.float32)
    xd, wd, bd = (torch.from_numpy(a).cuda() for a in (x, w, b))
    out = torch.full((M_ + 1, N_), -1.0, device="cuda")
    N.check(lib.orcai_dense_sigmoid(N.ptr(xd), N.ptr(wd), N.ptr(bd), M_, K, N_, N.ptr(out), st), "dense_sigmoid")
    want = 1.0 / (1.0 + np.exp(-(x.astype(np.float64) @ w.astype(np.float64) + b)))
    got = out.cpu().numpy()
    assert np.abs(got[:M_] - want).max() <= 1e-6 and (got[M_] == -1.0).all()  # nothing written past the last row
