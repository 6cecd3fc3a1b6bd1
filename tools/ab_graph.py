"""Eager launches against hipGraph replay of the whole training step (Trainer.train_step_graphed), f32 orcai-V1 and the three f16
width variants of the hyper-parameter sweep, in one process.  usage: ab_graph.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench_predict import HPS_FILTER_SETS
from orcai_amd.architectures import ResNetLSTM
from orcai_amd.training import Trainer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(4)
B = 64
x = torch.rand((B, 736, 171), device=dev, generator=g).view(-1)
y = (torch.rand((B, 46, 7), device=dev, generator=g) > 0.7).float()
cases = [("f32 orcai-V1", [30, 40, 50, 60], "f32")] + [(f"f16 {k}", v, "f16") for k, v in HPS_FILTER_SETS.items()]
for name, filters, prec in cases:
    res = {}
    for mode in ("eager", "graph", "eager", "graph"):
        model = ResNetLSTM((736, 171, 1), 7, filters, 3, 0.5, 128, seed=1, precision=prec)
        tr = Trainer(model, 1e-4, seed=0)
        fn = (lambda: tr.train_step(x, 736 * 171, B, y)) if mode == "eager" else (lambda: tr.train_step_graphed(x, 736 * 171, B, y))
        for _ in range(4):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        res.setdefault(mode, []).append((time.perf_counter() - t0) / steps * 1e3)
        del tr, model
    print(name, {k: [round(v, 3) for v in vs] for k, vs in res.items()}, flush=True)
