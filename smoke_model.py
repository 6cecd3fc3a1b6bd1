"""smoke(): one small invocation of the model hot path on cuda:0, checked against the CPU oracle."""

from __future__ import annotations


def run() -> None:
    import numpy as np
    import torch

    from oracle import model_ref as M
    from oracle import postprocess_ref as P
    from orcai_amd.architectures import ResNetLSTM
    from orcai_amd.predict import aggregate_predictions_device

    p = M.calibrated_params(seed=11, calib_batch=1)
    model = ResNetLSTM((736, 171, 1), 7, [30, 40, 50, 60], 3, 0.0, 128)
    model.set_weights_dict(p)
    T = 736 + 368 * 2 + 50
    spec = np.random.default_rng(5).random((T, 171), dtype=np.float32)
    pred = model.predict_spectrogram(torch.from_numpy(spec).cuda())
    ref = M.forward_ref(p, P.slice_snippets(spec, 736))
    err = float(np.abs(pred.cpu().numpy() - ref).max())
    assert pred.shape == (3, 46, 7) and err <= 1e-5, err
    agg, cnt = aggregate_predictions_device(pred, T, 736, 4)
    agg_ref, cnt_ref = P.aggregate_predictions_ref(pred.cpu().numpy(), T, 736, 4, 7)
    assert np.array_equal(agg, agg_ref) and np.array_equal(cnt, cnt_ref)
    print(f"smoke: ResNetLSTM forward on 3 snippets max|delta p| vs oracle = {err:.2e}; overlap average bit-exact")
    # one tiny training step (forward in training mode + masked BCE + L2 + full backward) against torch autograd on the CPU
    from oracle import train_ref as T
    from orcai_amd.training import Trainer

    cfg = dict(input_shape=(32, 12, 1), filters=(10, 20), kernel_size=3, lstm_units=64, num_labels=3)
    p = M.calibrated_params(seed=5, **cfg)
    rng = np.random.default_rng(5)
    x = rng.random((3, 32, 12, 1), dtype=np.float32)
    y = (rng.random((3, 8, 3)) > 0.5).astype(np.float32)
    y[0, :, 0] = -1.0
    ref = T.loss_and_grads(p, x, y, None, 0.0)
    small = ResNetLSTM(cfg["input_shape"], 3, [10, 20], 3, 0.0, 64)
    small.set_weights_dict(p)
    tr = Trainer(small, learning_rate=1e-3)
    out = tr.forward_backward(torch.from_numpy(np.ascontiguousarray(x[..., 0])).cuda().view(-1), 32 * 12, 3, torch.from_numpy(y).cuda(), masks=None)
    acc = out["acc"].cpu().numpy()
    loss = acc[0] / acc[1] + acc[3]
    assert abs(loss - ref["loss"]) <= 2e-6 * max(1.0, abs(ref["loss"])), (loss, ref["loss"])
    g = tr.P.G("conv0/kernel").cpu().numpy()  # the gradient that has travelled through every layer of the backward pass
    gref = ref["grads"]["conv0/kernel"]
    gerr = float(np.abs(g - gref).max())
    assert gerr <= 5e-4 * max(1e-3, float(np.abs(gref).max())), gerr
    print(f"smoke: training step loss {loss:.6f} vs autograd oracle {ref['loss']:.6f}; d loss / d conv0 kernel max|delta| = {gerr:.2e}")
    # the f16 path (BASELINE configs[4]) on the same small network: forward against the fp32 oracle
    half = ResNetLSTM(cfg["input_shape"], 3, [10, 20], 3, 0.0, 64, precision="f16")
    half.set_weights_dict(p)
    herr = float(np.abs(half.predict(x) - M.forward_ref(p, x)).max())
    assert herr <= 5e-3, herr
    print(f"smoke: f16 path forward max|delta p| vs fp32 oracle = {herr:.2e}")
