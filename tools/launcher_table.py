"""Per-launcher time of the f32 training step (batch 64, orcai-V1) with every orcai_* launcher bracketed by HIP events, for a TrunkTrainer switch
off and on: usage: launcher_table.py flag  (the brackets slow the step; only the differences between the two columns mean anything)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench_predict import TrainWorkload

flag = sys.argv[1]
w = TrainWorkload(torch.device("cuda", 0), 0)
cols = {}
for val in (False, True):
    setattr(w.trainer.trunk, flag, val)
    for _ in range(3):
        w.step(False)
    w.timed.mode, w.timed.events = "all", {}
    for _ in range(4):
        w.trainer.train_step(w.x, 736 * 171, w.B, w.y, world_size=1)
    torch.cuda.synchronize()
    cols[val] = {k: (sum(a.elapsed_time(b) for a, b, _ in v) / 4, len(v) // 4) for k, v in w.timed.events.items()}
    w.timed.mode, w.timed.events = "dominant", None
names = sorted(set(cols[False]) | set(cols[True]), key=lambda k: -(cols[True].get(k, (0, 0))[0] + cols[False].get(k, (0, 0))[0]))
print(f"{'launcher':40s} {flag}=False  calls | {flag}=True  calls")
for n in names:
    a, b = cols[False].get(n, (0.0, 0)), cols[True].get(n, (0.0, 0))
    print(f"{n:40s} {a[0]:8.3f} {a[1]:4d} | {b[0]:8.3f} {b[1]:4d}")
print(f"{'sum':40s} {sum(v[0] for v in cols[False].values()):8.3f}      | {sum(v[0] for v in cols[True].values()):8.3f}")
