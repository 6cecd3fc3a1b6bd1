"""A/B of the entry kernel: strip tiles (orcai_entry_tile 10) against the marching form (100 + tiles per workgroup): bit equality of the predictions and per-layer
times.  usage: ab_entry_march.py [seconds] [mode,mode,...] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth_pcm_device, SPEC_PARAM
from orcai_amd import _native as N
from orcai_amd.frontend import FrontEnd
from orcai_amd.architectures import ResNetLSTM

dev = torch.device("cuda", 0)
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 1200.0
modes = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [10, 108]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
pcm = synth_pcm_device(int(secs * 48000), 3, dev)
spec = FrontEnd(dev).make_spectrogram(pcm, SPEC_PARAM)
model = ResNetLSTM((736, 171, 1), 7, [30, 40, 50, 60], 3, 0.0, 128, seed=1)
model.prepare()
ref = None
before = N.lib().orcai_entry_tile(-1)
for r in range(reps):
    for mode in modes:
        N.lib().orcai_entry_tile(mode)
        for it in range(2):
            model.kernel_events = {}
            pred = model.predict_spectrogram(spec)
            torch.cuda.synchronize()
        tot = {k: sum(a.elapsed_time(b) for a, b in v) for k, v in model.kernel_events.items()}
        if ref is None:
            ref = pred.clone()
        print(f"entry_tile={mode} bit-identical={bool(torch.equal(pred, ref))} model={sum(tot.values()):.2f} ms", {k: round(v, 2) for k, v in tot.items() if k.startswith(("b1", "conv0"))}, flush=True)
N.lib().orcai_entry_tile(before)
