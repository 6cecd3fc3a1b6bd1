"""orcai_sepconv_planes_stats + orcai_bn_finish_sharded eager against three replays of the same two calls captured in a hipGraph.  With the
accumulators cleared by hipMemsetAsync (the first version of the launcher) replays 1 and 2 came back with every odd double of the
accumulators 1.0 too low; with the kernel fill of csrc/zero_fill.h all replays are bit-identical to the eager launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from orcai_amd import _native as N
lib = N.lib(); dev = torch.device("cuda", 0)
for (Cin, Cout, H, W) in ((16, 10, 32, 12), (16, 30, 21, 171), (10, 20, 16, 6)):
    B, CQ, CQo, WP = 8, (Cin + 3) // 4, (Cout + 3) // 4, lib.orcai_padded_width(W, 3)
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.zeros(B, CQ * 4, H + 2, WP); x[:, :Cin, 1:H + 1, :W] = torch.randn(B, Cin, H, W, generator=g)
    planes = x.view(B, CQ, 4, H + 2, WP).permute(0, 1, 3, 4, 2).contiguous().to(dev)
    dw = torch.randn(CQ, 9, 4, generator=g).to(dev); pw = (torch.randn(Cin, Cout, generator=g) / Cin ** 0.5).to(dev)
    scale, shift = torch.ones(64).to(dev), torch.randn(64, generator=g).to(dev)
    out, u = torch.zeros((B, CQo, H + 2, WP, 4), device=dev), torch.zeros((B, CQ, H + 2, WP, 4), device=dev)
    shards = torch.zeros(8 * 16 * 32, dtype=torch.float64, device=dev)
    mean, var = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
    def step():
        st = N.stream_ptr()
        rc = lib.orcai_sepconv_planes_stats(N.ptr(planes), B, Cin, H, W, 1, N.ptr(dw), N.ptr(pw), N.ptr(scale), N.ptr(shift), Cout, N.ptr(out), N.ptr(u), N.ptr(shards), st)
        assert rc == 0, rc
        assert lib.orcai_bn_finish_sharded(N.ptr(shards), B, Cout, H, W, N.ptr(mean), N.ptr(var), st) == 0
    step(); torch.cuda.synchronize()
    m0, v0 = mean.clone(), var.clone(); s0 = shards.clone()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step(); side.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=side):
            step()
    torch.cuda.current_stream().wait_stream(side)
    for r in range(3):
        mean.zero_(); var.zero_()
        gr.replay(); torch.cuda.synchronize()
        d = (shards - s0); idx = d.nonzero().flatten().tolist()
        if r == 1: print("   diff entries", [(i, i // (CQo * 8), (i // 8) % CQo, i % 8, float(d[i]), float(s0[i])) for i in idx[:12]], len(idx))
        print((Cin, Cout, H, W), "replay", r, "max|dmean|", float((mean - m0).abs().max()), "max|dvar|", float((var - v0).abs().max()), "shards sum", float(shards.sum()))
