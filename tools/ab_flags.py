"""Same-process A/B of TrunkTrainer switches on the f32 training step (batch 64, orcai-V1): usage: ab_flags.py flag[,flag...] [reps]
Each flag is toggled False / True alternately (reps times); prints ms per step for every setting.  Box-to-box spread is larger than the effects,
so only same-process comparisons count."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench_predict import TrainWorkload

flags = sys.argv[1].split(",")
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
w = TrainWorkload(torch.device("cuda", 0), 0)
tr = w.trainer.trunk


def timed(n=20):
    for _ in range(3):
        w.step(False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        w.step(False)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for flag in flags:
    for r in range(reps):
        for val in (False, True):
            setattr(tr, flag, val)
            print(f"{flag} = {val}: {timed():.3f} ms/step", flush=True)
    setattr(tr, flag, True)
