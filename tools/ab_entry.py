"""A/B of the fused entry convolution (orcai_conv0_sepconv) against conv0 + sepconv: bit equality and per-layer times.
usage: ab_entry.py [seconds] [chunk] [windows-per-wave ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth_pcm_device, SPEC_PARAM
from orcai_amd import _native as N
from orcai_amd.frontend import FrontEnd
from orcai_amd.architectures import ResNetLSTM

dev = torch.device("cuda", 0)
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 128
nws = [int(a) for a in sys.argv[3:]] or [1, 2, 4, 8]
pcm = synth_pcm_device(int(secs * 48000), 3, dev)
spec = FrontEnd(dev).make_spectrogram(pcm, SPEC_PARAM)
model = ResNetLSTM((736, 171, 1), 7, [30, 40, 50, 60], 3, 0.0, 128, seed=1)
model.prepare()
ref = None
for fuse, nw in [(False, 1)] + [(True, n) for n in nws]:
    model.fuse_entry = fuse
    N.lib().orcai_entry_windows(nw)
    for it in range(3):
        model.kernel_events = {}
        pred = model.predict_spectrogram(spec, chunk=chunk)
        torch.cuda.synchronize()
    tot = {k: sum(a.elapsed_time(b) for a, b in v) for k, v in model.kernel_events.items()}
    if ref is None:
        ref = pred.clone()
    print(f"fuse_entry={fuse} nw={nw} bit-identical={bool(torch.equal(pred, ref))} total={sum(tot.values()):.2f} ms",
          {k: round(v, 2) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:5]}, flush=True)
