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
