"""The hipMemsetAsync-node probe of tools/microbench/graph_memset.hip inside a torch process: memory from torch's caching allocator, capture
through torch.cuda.graph (global capture mode, private pool), the memset issued through ctypes like the library's launchers are.  Graph =
{ hipMemsetAsync(acc, 0) ; acc += vals ; snap = acc }.  Prints every replay whose snapshot differs from vals, with bit patterns."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

hip = C.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
hip.hipMemsetAsync.restype = C.c_int
dev = torch.device("cuda", 0)
bad_total = 0
for n in (3 * 8 * 32, 8 * 8 * 32, 15 * 8 * 32, 4096):
    for off in (0, 1):
        base = torch.full((n + 2,), 7.0, dtype=torch.float64, device=dev)
        acc = base[off:off + n]
        vals = (torch.arange(n, device=dev, dtype=torch.float64) % 7 + 1) * 0.37
        snap = torch.zeros(n, dtype=torch.float64, device=dev)
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        def step():
            rc = hip.hipMemsetAsync(acc.data_ptr(), 0, n * 8, torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rc
            acc.add_(vals)
            snap.copy_(acc)
        with torch.cuda.stream(side):
            step(); side.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                step()
        torch.cuda.current_stream().wait_stream(side)
        for r in range(5):
            junk = torch.full((1 << 16,), -1.0, dtype=torch.float64, device=dev)  # eager work between replays, incl. a fill with double -1.0
            junk2 = torch.zeros(1 << 16, device=dev); junk2.zero_()
            g.replay(); torch.cuda.synchronize()
            d = (snap - vals)
            idx = d.nonzero().flatten().tolist()
            if idx:
                bad_total += len(idx)
                print(f"n {n} off {off} replay {r}: {len(idx)} wrong; first", [(i, float(d[i]), hex(np.float64(float(snap[i])).view(np.uint64))) for i in idx[:6]], flush=True)
print("torch graph memset probe:", "MEMSET NODE REPLAYS WRONG" if bad_total else "all replays exact", bad_total)
