"""Every launcher call of one f32 training step (batch 64, orcai-V1) in launch order with its HIP-event time (mean of 4 bracketed steps) and its
integer arguments (shapes): where a launcher's total in tools/launcher_table.py comes from.  usage: launcher_calls.py [min_us]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench_predict import TrainWorkload

min_us = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
w = TrainWorkload(torch.device("cuda", 0), 0)
for _ in range(3):
    w.step(False)
w.timed.mode, w.timed.events = "all", {}
order = []
orig = dict(w.timed.events)
for _ in range(4):
    w.trainer.train_step(w.x, 736 * 171, w.B, w.y, world_size=1)
torch.cuda.synchronize()
rows = []
for name, ev in w.timed.events.items():
    n = len(ev) // 4
    for i in range(n):
        t = sum(ev[s * n + i][0].elapsed_time(ev[s * n + i][1]) for s in range(4)) / 4
        ints = [a for a in ev[i][2] if isinstance(a, int) and not isinstance(a, bool) and abs(a) < 100000]
        rows.append((t, name, i, ints[:8]))
tot = sum(r[0] for r in rows)
print(f"sum of bracketed launcher time {tot:.3f} ms per step ({len(rows)} calls)")
for t, name, i, ints in sorted(rows, key=lambda r: -r[0]):
    if t * 1e3 >= min_us:
        print(f"{t * 1e3:9.1f} us  {name:36s} #{i:<2d} {ints}")
