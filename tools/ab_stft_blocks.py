"""Persistent-grid size of the STFT kernel (orcai_stft_blocks) against its kernel time on configs[1] (1 024 snippets), same process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from orcai_amd import _native as N

lib = N.lib()
print("occupancy (workgroups per compute unit, runtime):", lib.orcai_stft_occupancy(), flush=True)
w = bench.FrontendWorkload(torch.device("cuda", 0), 0)
for rep in range(2):
    for blocks in (512, 640, 768, 1024, 1280, 1536):
        lib.orcai_stft_blocks(blocks)
        w.ev = []
        for _ in range(3):
            w.step(False)
        torch.cuda.synchronize()
        import time
        t0 = time.perf_counter()
        for _ in range(20):
            w.step(True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20 * 1e3
        ms = sum(a.elapsed_time(b) for a, b in w.ev) / len(w.ev)
        print(f"blocks {blocks}: stft_db_kernel {ms:.4f} ms, pipeline {dt:.4f} ms", flush=True)
