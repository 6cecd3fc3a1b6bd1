"""Does the training step issue any hipMemset*?  Run with LD_PRELOAD=tools/bin/libcount_memsets.so (tools/microbench/count_memsets.c): the
interposer counts every hipMemset* entry point; this script reads the counter around N eager steps (a captured step issues the same calls).
A hipMemsetAsync node replayed on the default stream re-reads a stale fill pattern under the HIP 7.0.51831 runtime of the torch wheel
(profiles/r03_graph_memset_probe.log), so the step must not contain any."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from orcai_amd.architectures import ResNetLSTM
from orcai_amd.training import Trainer

probe = ctypes.CDLL(None).probe_memset_calls
probe.restype = ctypes.c_long
for prec in ("f32", "f16"):
    c0 = probe()
    tr = Trainer(ResNetLSTM((64, 171, 1), 3, [30, 40], 3, 0.3, 64, seed=1, precision=prec), learning_rate=1e-3, seed=5)
    rng = np.random.default_rng(0)
    x = torch.from_numpy(rng.random((4, 64, 171), dtype=np.float32)).cuda().view(-1)
    y = torch.from_numpy((rng.random((4, 16, 3)) > 0.5).astype(np.float32)).cuda()
    tr.train_step(x, 64 * 171, 4, y)
    torch.cuda.synchronize()
    c1 = probe()
    for _ in range(10):
        tr.train_step(x, 64 * 171, 4, y)
    torch.cuda.synchronize()
    c2 = probe()
    t = torch.zeros(1000, device="cuda"); t.zero_(); u = torch.zeros(7, dtype=torch.float64, device="cuda")
    c3 = probe()
    print(f"{prec}: hipMemset* calls: construction + first step {c1 - c0}, ten further steps {c2 - c1}, torch.zeros / zero_ x3 {c3 - c2}", flush=True)
