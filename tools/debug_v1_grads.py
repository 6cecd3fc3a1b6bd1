"""Which kernel deviates at the orcai-V1 shape?  The whole-step gradient comparison of tests/test_train_full_gpu.py at 736 x 171, B = 2, under
different TrunkTrainer switches; prints every gradient tensor whose error exceeds 2e-4 of the tensor's largest element."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from oracle import model_ref as M, train_ref as T
from orcai_amd.architectures import ResNetLSTM
from orcai_amd.training import Trainer

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (736, 171)
cfg = dict(input_shape=(H, W, 1), filters=(30, 40, 50, 60), kernel_size=3, lstm_units=128, num_labels=7)
B, seed, rate = 2, 11, 0.5
p = M.calibrated_params(seed=seed, **cfg)
rng = np.random.default_rng(seed)
for k in p:
    if k.endswith(("gamma", "beta")):
        p[k] = (p[k] + 0.2 * rng.standard_normal(p[k].shape)).astype(np.float32)
steps, L, u = H // 16, 7, 128
x = rng.random((B, H, W, 1), dtype=np.float32)
y = (rng.random((B, steps, L)) > 0.5).astype(np.float32)
y[0, :, 0] = -1.0
masks = {k: (rng.random((B, steps, d)) > rate).astype(np.float32) for k, d in (("drop1", 2 * u), ("drop2", 2 * u), ("drop3", 128))}
ref = T.loss_and_grads(p, x, y, masks, rate)
ref32 = T.loss_and_grads(p, x, y, masks, rate, dtype=torch.float32)
print("f32 CPU autograd vs f64 (what plain f32 arithmetic gives):", {n: f"{float(np.abs(ref32['grads'][n] - g).max()) / max(1e-3, float(np.abs(g).max())):.1e}" for n, g in ref["grads"].items()
      if float(np.abs(ref32['grads'][n] - g).max()) / max(1e-3, float(np.abs(g).max())) > 2e-4})
xd = torch.from_numpy(np.ascontiguousarray(x[..., 0])).cuda().view(-1)
yd = torch.from_numpy(y).cuda()
md = {k: torch.from_numpy(v).cuda() for k, v in masks.items()}
configs = [{}, {"fused_dw_bwd": False}, {"fused_pw_wgrad": False}, {"apply_on_load": False}, {"stats_in_epilogue": False}, {"conv0_two_pass": False}, {"bias_in_pool": False},
           {"dgrad_epilogues": False, "fused_dw_bwd": False},
           {"fused_dw_bwd": False, "fused_pw_wgrad": False, "apply_on_load": False, "stats_in_epilogue": False, "conv0_two_pass": False, "bias_in_pool": False, "dgrad_epilogues": False, "conv0_march": False, "conv0_in_dgrad": False}]
for flags in configs:
    model = ResNetLSTM(cfg["input_shape"], L, list(cfg["filters"]), 3, rate, u)
    model.set_weights_dict(p)
    tr = Trainer(model, learning_rate=1e-3)
    for k, v in flags.items():
        setattr(tr.trunk, k, v)
    out = tr.forward_backward(xd, H * W, B, yd, masks=md)
    acc = out["acc"].cpu().numpy()
    errs = {}
    for name, g in ref["grads"].items():
        got = tr.P.G(name).cpu().numpy()
        zb = name.endswith("/bias") and not name.startswith(("dense2", "lstm", "dense1")) and "res" not in name
        scale = max(1e-3, float(np.abs(g).max())) if not zb else 1.0
        errs[name] = float(np.abs(got - g).max()) / scale
    bad = {k: f"{v:.1e}" for k, v in sorted(errs.items(), key=lambda kv: -kv[1]) if v > 2e-4}
    st = {}
    tr.trunk.update_moving_stats()
    print(f"flags {flags}: dprobs {np.abs(out['probs'].cpu().numpy() - ref['probs']).max():.1e} loss err {abs(acc[0] / acc[1] + acc[3] - ref['loss']):.1e}  tensors over 2e-4: {bad}", flush=True)
    del tr, model
