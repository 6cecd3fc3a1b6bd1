"""Per-step values of the dominant kernel's HIP-event bracket in the training workload (why does roofline.kernel_ms depend on the warm-up length?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_predict as bp

w = bp.TrainWorkload(torch.device("cuda", 0), 0)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    w.step(False)
torch.cuda.synchronize()
for _ in range(30):
    w.step(True)
torch.cuda.synchronize()
for name, calls in w.timed.events.items():
    print(name, [round(a.elapsed_time(b), 3) for a, b, _ in calls])
print("steps", [round(a.elapsed_time(b), 3) for a, b in w.ev])
