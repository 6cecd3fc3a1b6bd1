"""A/B of block 1's front: conv0+sep_a and sep_b as two launches (mode 0) against orcai_block_front (mode = rows * 100 + groups, e.g.
804 = 8-row tiles, 4 tiles per workgroup): bit equality of the model output and per-layer times (HIP events) on the same spectrogram.
usage: ab_front.py [seconds] [chunk] [mode ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth_pcm_device, SPEC_PARAM
from orcai_amd import _native as N
from orcai_amd.frontend import FrontEnd
from orcai_amd.architectures import ResNetLSTM

dev = torch.device("cuda", 0)
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 1200.0
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 128
modes = [int(a) for a in sys.argv[3:]] or [0, 804]
pcm = synth_pcm_device(int(secs * 48000), 3, dev)
spec = FrontEnd(dev).make_spectrogram(pcm, SPEC_PARAM)
model = ResNetLSTM((736, 171, 1), 7, [30, 40, 50, 60], 3, 0.0, 128, seed=1)
model.prepare()
lib = N.lib()
ref = None
for mode in modes + [modes[0]]:
    model.fuse_front = mode != 0
    if mode:
        lib.orcai_block_front_config(mode // 100, mode % 100)
    for it in range(3):
        model.kernel_events = {}
        pred = model.predict_spectrogram(spec, chunk=chunk)
        torch.cuda.synchronize()
    tot = {k: sum(a.elapsed_time(b) for a, b in v) for k, v in model.kernel_events.items()}
    if ref is None:
        ref = pred.clone()
    same = bool(torch.equal(pred, ref))
    print(f"mode={mode:5d} bit-identical={same} maxdiff={float((pred - ref).abs().max()):.3g} total={sum(tot.values()):.2f} ms",
          {k: round(v, 2) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:6]}, flush=True)
