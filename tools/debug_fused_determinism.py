"""One training step from identical state in fresh Trainers: run-to-run spread of loss, batch statistics and gradients with the BatchNorm
statistics in the conv epilogue on / off (float atomics in the weight-gradient sums give ~1e-7 relative either way)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from orcai_amd.architectures import ResNetLSTM
from orcai_amd.training import Trainer

rng = np.random.default_rng(0)
x = torch.from_numpy(rng.random((8, 32, 12), dtype=np.float32)).cuda().view(-1)
y = torch.from_numpy((rng.random((8, 8, 3)) > 0.5).astype(np.float32)).cuda()

def one(fused, steps):
    tr = Trainer(ResNetLSTM((32, 12, 1), 3, [10, 20], 3, 0.3, 64, seed=1), learning_rate=3e-3, seed=5)
    tr.trunk.stats_in_epilogue = fused
    for _ in range(steps - 1):
        tr.train_step(x, 32 * 12, 8, y)
    out = tr.forward_backward(x, 32 * 12, 8, y)
    torch.cuda.synchronize()
    a = out["acc"].cpu().numpy()
    return a[0] / a[1], tr.P.g.cpu().numpy().copy(), tr.P.stats_flat.cpu().numpy().copy() if hasattr(tr.P, "stats_flat") else None, tr.P.batch_flat.cpu().numpy().copy()

for steps in (1, 4, 8):
    for fused in (True, False):
        runs = [one(fused, steps) for _ in range(6)]
        L = np.array([r[0] for r in runs]); G = np.stack([r[1] for r in runs]); S = np.stack([r[3] for r in runs])
        gs = np.abs(G).max()
        print(f"steps={steps} fused={fused}: loss spread {L.max() - L.min():.2e}; grad spread {np.abs(G - G[0]).max() / gs:.2e} (rel. to max |g| {gs:.2e}); "
              f"batch-stat spread {np.abs(S - S[0]).max():.2e}", flush=True)
    # fused vs unfused
