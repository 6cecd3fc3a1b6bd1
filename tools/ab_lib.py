"""Same-process A/B of a library-global experiment switch on the f32 training step: usage: ab_lib.py <setter> v1,v2,... [reps]
(e.g. ab_lib.py orcai_pw_wgrad_tiles 0,4,6)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench_predict import TrainWorkload
from orcai_amd import _native as N

setter = getattr(N.lib(), sys.argv[1])
vals = [int(v) for v in sys.argv[2].split(",")]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
w = TrainWorkload(torch.device("cuda", 0), 0)
for r in range(reps):
    for v in vals:
        setter(v)
        for _ in range(3):
            w.step(False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            w.step(False)
        torch.cuda.synchronize()
        print(f"{sys.argv[1]}({v}): {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms/step", flush=True)
