import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from orcai_amd.architectures import ResNetLSTM
from orcai_amd.training import Trainer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
model = ResNetLSTM((736, 171, 1), 7, [30, 40, 50, 60], 3, 0.5, 128, seed=1)
tr = Trainer(model, 1e-4)
g = torch.Generator(device="cuda"); g.manual_seed(0)
x = torch.rand((B, 736, 171), device="cuda", generator=g).view(-1)
y = (torch.rand((B, 46, 7), device="cuda", generator=g) > 0.5).float()
for it in range(4):
    torch.cuda.synchronize(); t0 = time.time()
    out = tr.train_step(x, 736 * 171, B, y)
    torch.cuda.synchronize(); dt = time.time() - t0
    a = out["acc"].cpu().numpy()
    print(f"step {it}: {dt*1e3:.1f} ms  {B/dt:.1f} snippets/s  bce {a[0]/a[1]:.4f}  mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
