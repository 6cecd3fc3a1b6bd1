"""A/B of TrunkTrainer switches on the f32 training step in one process.  usage: ab_train_order.py [dgrad_first|stats_in_epilogue]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench_predict import TrainWorkload

w = TrainWorkload(torch.device("cuda", 0), 0)
flag = sys.argv[1] if len(sys.argv) > 1 else "dgrad_first"
for mode in (0, 1, 0, 1, 0, 1):
    setattr(w.trainer.trunk, flag, bool(mode))
    for _ in range(3):
        w.step(False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        w.step(False)
    torch.cuda.synchronize()
    print(f"{flag}={mode}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms/step", flush=True)
