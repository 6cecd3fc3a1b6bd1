#!/usr/bin/env python3
"""Host-side duration of each phase of PredictWorkload.step (no synchronisation added): a phase that takes as long as its GPU work has a hidden
device synchronisation in it, and the next recording's kernels are then queued late (the gap in front of its first kernel)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench_predict as bp  # noqa: E402
from orcai_amd.predict import aggregate_predictions_device  # noqa: E402

w = bp.PredictWorkload(torch.device("cuda", 0), 0)
for _ in range(3):
    w.step(False)
w.drain()
torch.cuda.synchronize()
acc = {}
N = 6
t_all = time.perf_counter()
for _ in range(N):
    t0 = time.perf_counter(); spec = w.fe.make_spectrogram(w.pcm, bp.SPEC_PARAM); t1 = time.perf_counter()
    pred = w.model.predict_spectrogram(spec, chunk=w.chunk); t2 = time.perf_counter()
    pending = aggregate_predictions_device(pred, w.T, 736, 4, wait=False); t3 = time.perf_counter()
    w.drain(); t4 = time.perf_counter()
    w.in_flight = pending
    for k, v in (("front end", t1 - t0), ("model", t2 - t1), ("aggregate", t3 - t2), ("drain (previous recording's host half)", t4 - t3)):
        acc[k] = acc.get(k, 0.0) + v
w.drain()
torch.cuda.synchronize()
total = (time.perf_counter() - t_all) / N * 1e3
print({k: round(v / N * 1e3, 3) for k, v in acc.items()}, "ms of HOST time per step; wall per step", round(total, 3))
