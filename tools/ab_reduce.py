"""Read-only training passes in isolation: orcai_bn_planes_stats (and its backward twin's sums) on a block-1-sized tensor of random data,
(a) launched back to back, (b) each launch behind a kernel that has just WRITTEN 1 GB (what the step does: the statistics pass follows
the convolution that produced the tensor).  HIP events around the measured launch only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from orcai_amd import _native as N

lib = N.lib()
dev = torch.device("cuda", 0)
B, C, H, W, k = 64, 30, 736, 171, 3
WP = lib.orcai_padded_width(W, k)
CQ = (C + 3) // 4
x = torch.randn((B, CQ, H + 2, WP, 4), device=dev)
other = torch.empty_like(x)
scratch = torch.zeros(8 * 16 * 32, dtype=torch.float64, device=dev)
mean, var = torch.empty(64, device=dev), torch.empty(64, device=dev)
gb = x.numel() * 4 / 1e9
st = N.stream_ptr()


def stats():
    N.check(lib.orcai_bn_planes_stats(x.data_ptr(), B, C, H, W, k, scratch.data_ptr(), mean.data_ptr(), var.data_ptr(), st), "stats")


def timed(pre, n=10):
    ts = []
    for _ in range(n):
        pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); stats(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


for _ in range(3):
    stats()
torch.cuda.synchronize()
for name, pre in (("back to back (previous launch was the same read)", lambda: stats()),
                  ("behind a 1 GB write to ANOTHER buffer", lambda: other.copy_(x)),
                  ("behind a rewrite of the tensor itself", lambda: x.mul_(1.0)),
                  ("after an idle device", lambda: torch.cuda.synchronize())):
    ms = timed(pre)
    print(f"bn_planes_stats on {gb:.2f} GB, {name}: {ms * 1e3:.0f} us = {gb / ms:.2f} TB/s", flush=True)
