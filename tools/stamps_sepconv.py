"""Diagnostic: build model_fwd.hip with -DORCAI_STAMPS into a separate .so and print the share of block time per phase
of sepconv_kernel for the b1/sep_b shape.  Never part of the product build."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
so = os.path.join(ROOT, "gpurun_out", "libstamps.so")
if not os.path.exists(so):
    subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17", "-DORCAI_STAMPS", f"-I{ROOT}/include",
                    f"{ROOT}/orcai_amd/csrc/model_fwd.hip", "-o", so], check=True)
lib = C.CDLL(so)
B, Cin, H, W, Cout = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 30, 736, 171, 30
WP, HP = 172, 738
x = torch.zeros((B, Cin, HP, WP), device="cuda"); x[:, :, 1:-1, :W] = torch.randn((B, Cin, H, W), device="cuda")
y = torch.zeros((B, Cout, HP, WP), device="cuda")
dw = torch.randn((Cin, 9), device="cuda"); pw = torch.randn((Cin, Cout), device="cuda"); sc = torch.ones(Cout, device="cuda"); sh = torch.zeros(Cout, device="cuda")
vp = lambda t: C.c_void_p(t.data_ptr())
def run():
    return lib.orcai_sepconv_bn(vp(x), B, Cin, H, W, 3, 0, vp(dw), vp(pw), vp(sc), vp(sh), Cout, 0, 0, vp(y), None)
import numpy as np
nblk = min(184 * B, 16384)
buf = (C.c_ulonglong * (8 * nblk))()
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
lib.orcai_debug_stamps(buf, nblk)
a = np.frombuffer(buf, dtype=np.uint64).reshape(nblk, 8).astype(np.float64)
names = ["barrier0(prev mfma done)", "stage(load+store)", "wdw/afrag+barrier1", "depthwise", "barrier2", "mfma", "epilogue"]
m = a.mean(axis=0); tot = m[:7].sum()
print(f"kernel {e0.elapsed_time(e1):.3f} ms for {B} snippets; wave1 cycles per block: {tot:.0f}")
for n, v in zip(names, m[:7]):
    print(f"  {n:28s} {v:9.0f} cycles/block  {100.0 * v / tot:5.1f} %")
