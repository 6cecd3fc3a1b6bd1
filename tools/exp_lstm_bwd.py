"""Times orcai_lstm_bwd in the variant libraries under build/variants (debug experiment)."""
import ctypes, glob, os, sys
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B, T, U = 64, 46, 128
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(0)
dH = torch.randn((B, T, 2 * U), device=dev, generator=g) * 0.1
gates = torch.rand((B, T, 2, 4 * U), device=dev, generator=g)
cs = torch.randn((B, T, 2, U), device=dev, generator=g) * 0.5
Uw = torch.randn((2, U, 4 * U), device=dev, generator=g) * 0.05
dxz = torch.empty((B, T, 2, 4 * U), device=dev)
for path in sorted(glob.glob(os.path.join(root, "build/variants/th_*.so"))):
    lib = ctypes.CDLL(path)
    f = lib.orcai_lstm_bwd
    f.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 3 + [ctypes.c_void_p, ctypes.c_void_p]
    f.restype = ctypes.c_int
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        rc = f(dH.data_ptr(), gates.data_ptr(), cs.data_ptr(), Uw.data_ptr(), B, T, U, dxz.data_ptr(), st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f(dH.data_ptr(), gates.data_ptr(), cs.data_ptr(), Uw.data_ptr(), B, T, U, dxz.data_ptr(), st)
    e1.record(); torch.cuda.synchronize()
    print(os.path.basename(path), rc, "us per call", e0.elapsed_time(e1) * 100, "per step", e0.elapsed_time(e1) * 100 / T, flush=True)
