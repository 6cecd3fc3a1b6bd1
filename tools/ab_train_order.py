"""A/B of the order of a separable conv's backward kernels in the training step (TrunkTrainer.dgrad_first), f32 and f16, one process.
usage: ab_train_order.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench_predict import TrainWorkload

w = TrainWorkload(torch.device("cuda", 0), 0)
for mode in (0, 1, 0, 1, 0, 1):
    w.trainer.trunk.dgrad_first = bool(mode)
    for _ in range(3):
        w.step(False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        w.step(False)
    torch.cuda.synchronize()
    print(f"dgrad_first={mode}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms/step", flush=True)
