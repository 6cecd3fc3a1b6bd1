"""Where the f16 training step departs from the f32 one: per-buffer and per-gradient relative differences (same weights, inputs, masks)."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from oracle import model_ref as M  # noqa: E402
from orcai_amd.architectures import ResNetLSTM  # noqa: E402
from orcai_amd.training import Trainer  # noqa: E402

cfg = dict(input_shape=(32, 12, 1), filters=(10, 20), kernel_size=3, lstm_units=64, num_labels=3)
B, rate = 3, 0.5
p = M.calibrated_params(seed=5, **cfg)
rng = np.random.default_rng(5)
H, W, _ = cfg["input_shape"]
steps = H // 4
x = rng.random((B, H, W), dtype=np.float32)
y = (rng.random((B, steps, 3)) > 0.5).astype(np.float32)
masks = {k: torch.from_numpy((rng.random((B, steps, d)) > rate).astype(np.float32)).cuda() for k, d in (("drop1", 128), ("drop2", 128), ("drop3", 128))}
res = {}
for prec in ("f32", "f16"):
    m = ResNetLSTM(cfg["input_shape"], 3, list(cfg["filters"]), 3, rate, 64, precision=prec)
    m.set_weights_dict(p)
    tr = Trainer(m, 1e-3)
    out = tr.forward_backward(torch.from_numpy(x).cuda().view(-1), H * W, B, torch.from_numpy(y).cuda(), masks=masks)
    torch.cuda.synchronize()
    G = tr.trunk.G
    bufs = {}
    for k, t in tr.trunk.buf.items():
        Bq, CG, HP, WP, _ = t.shape
        bufs[k] = t.float().permute(0, 1, 4, 2, 3).reshape(Bq, CG * G, HP, WP).cpu().numpy()
    res[prec] = (bufs, {n: tr.P.G(n).cpu().numpy() / tr.grad_scale for n in tr.P.offsets}, tr.grad_scale)
b32, g32, _ = res["f32"]
b16, g16, S = res["f16"]
print("buffers (f16 vs f32), max|d| / max|ref|; gradient planes carry the loss scale", S)
for k in b32:
    a, b = b32[k], b16[k]
    c = min(a.shape[1], b.shape[1])
    a, b = a[:, :c], b[:, :c]
    sc = S if k.startswith(("d", "du")) else 1.0
    print(f"  {k:10s} {np.abs(b / sc - a).max() / max(1e-12, np.abs(a).max()):.2e}   max|ref| {np.abs(a).max():.3e}")
print("gradients")
for n in g32:
    print(f"  {n:28s} {np.abs(g16[n] - g32[n]).max() / max(1e-12, np.abs(g32[n]).max()):.2e}")
