"""A/B of TrunkTrainer switches on the f16 sweep (three width variants) in one process.  usage: ab_sweep_flags.py [flag]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench_predict import HpsearchWorkload

w = HpsearchWorkload(torch.device("cuda", 0), 0)
flag = sys.argv[1] if len(sys.argv) > 1 else "stats_in_epilogue"
for mode in (0, 1, 0, 1, 0, 1):
    for tr in w.trainers.values():
        setattr(tr.trunk, flag, bool(mode))
    for _ in range(3):
        w.step(False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        w.step(False)
    torch.cuda.synchronize()
    print(f"{flag}={mode}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms per sweep step", flush=True)
