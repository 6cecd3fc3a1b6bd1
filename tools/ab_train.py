"""A/B of separable-conv launcher modes on the f32 training step, in one process (box-to-box spread is larger than the effects):
usage: ab_train.py [mode ...]   (orcai_sepconv_tile_mode values 0..2; 1000 + pixels: tile mode 1 with orcai_outer_reduce_pixels forced)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench_predict import TrainWorkload
from orcai_amd import _native as N

modes = [int(a) for a in sys.argv[1:]] or [0, 1, 3]
w = TrainWorkload(torch.device("cuda", 0), 0)
lib = N.lib()
for mode in modes + modes:
    lib.orcai_sepconv_tile_mode(mode if mode < 1000 else 1)
    lib.orcai_outer_reduce_pixels(mode - 1000 if mode >= 1000 else 0)
    for _ in range(3):
        w.step(False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        w.step(False)
    torch.cuda.synchronize()
    print(f"mode {mode}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms/step", flush=True)
