#!/usr/bin/env python3
"""Every launcher call of ONE training step in launch order: time (HIP events around each call), algorithmic bytes and the rate against 8 TB/s.
usage: step_calls.py f32 | f16 [set3]     (the brackets slow the step; per-call times are what matters here)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench_predict as bp  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "f32"
dev = torch.device("cuda", 0)
if kind == "f32":
    w = bp.TrainWorkload(dev, 0)
    tl, tr, sym, nbytes = w.timed, w.trainer, bp._train_call_symbol, bp._train_call_bytes
else:
    os.environ["ORCAI_HPS_VARIANTS"] = sys.argv[2] if len(sys.argv) > 2 else "set3"
    w = bp.HpsearchWorkload(dev, 0)
    v = w.variants[0]
    tl, tr, sym, nbytes = w.timed[v], w.trainers[v], bp._h_call_symbol, bp._h_call_bytes
for _ in range(3):
    tr.train_step(w.x, 736 * 171, w.B, w.y, world_size=1)
torch.cuda.synchronize()
tl.mode, tl.events, tl.order = "all", {}, []
REP = 3
for _ in range(REP):
    tr.train_step(w.x, 736 * 171, w.B, w.y, world_size=1)
torch.cuda.synchronize()
# launch order = call order of the first repetition (tl.order); the time of a call = its fastest repetition
per_rep = len(tl.order) // REP
calls = []
for name, idx in tl.order[:per_rep]:
    n = len(tl.events[name]) // REP
    reps = [tl.events[name][idx + r * n] for r in range(REP)]
    calls.append((name, reps[0], min(a.elapsed_time(b) for a, b, _ in reps)))
tot = 0.0
print(f"{'launcher':34s} {'symbol':58s} {'ms':>7s} {'MB':>8s} {'TB/s':>6s}  shape")
for name, (e0, e1, a), ms in calls:
    by = nbytes(name, a)
    tot += ms
    ints = [x for x in a if isinstance(x, int) and 0 < x < 4096][:6]
    print(f"{name[6:40]:34s} {str(sym(name, a))[:58]:58s} {ms:7.3f} {'' if by is None else f'{by / 1e6:8.1f}'} {'' if by is None else f'{by / ms / 1e9:6.2f}'}  {ints}")
print(f"sum {tot:.3f} ms over {len(calls)} calls")
