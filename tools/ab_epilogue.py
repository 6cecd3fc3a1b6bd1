"""A/B of the x-pooled strip-tile epilogue (orcai_fast_epilogue 0 / 1) in one process: bit equality of the model output and per-layer
times (HIP events) on the same spectrogram.  usage: ab_epilogue.py [seconds] [chunk]"""
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
pcm = synth_pcm_device(int(secs * 48000), 3, dev)
spec = FrontEnd(dev).make_spectrogram(pcm, SPEC_PARAM)
model = ResNetLSTM((736, 171, 1), 7, [30, 40, 50, 60], 3, 0.0, 128, seed=1)
model.prepare()
lib = N.lib()
ref = None
for mode in (0, 1, 0, 1, 0, 1):
    lib.orcai_fast_epilogue(mode)
    for it in range(3):
        model.kernel_events = {}
        pred = model.predict_spectrogram(spec, chunk=chunk)
        torch.cuda.synchronize()
    tot = {k: sum(a.elapsed_time(b) for a, b in v) for k, v in model.kernel_events.items()}
    if ref is None:
        ref = pred.clone()
    print(f"fast_epilogue={mode} bit-identical={bool(torch.equal(pred, ref))} total={sum(tot.values()):.2f} ms",
          {k: round(v, 3) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:4]}, flush=True)
