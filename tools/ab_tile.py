"""A/B of the separable-conv launcher modes (orcai_sepconv_tile_mode: 0 one-window kernel, 1 strip + flat-range LDS tiles, 2 flat-range
tiles only): bit equality of the model output and per-layer times (HIP events) on the same spectrogram.
usage: ab_tile.py [seconds] [chunk] [mode ...]"""
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
nws = [int(a) for a in sys.argv[3:]] or [0, 1, 2]
pcm = synth_pcm_device(int(secs * 48000), 3, dev)
spec = FrontEnd(dev).make_spectrogram(pcm, SPEC_PARAM)
model = ResNetLSTM((736, 171, 1), 7, [30, 40, 50, 60], 3, 0.0, 128, seed=1)
model.prepare()
lib = N.lib()
ref = None
for nw in nws + [nws[0]]:
    lib.orcai_sepconv_tile_mode(nw % 10)
    lib.orcai_entry_tile({0: 10, 1: 0, 2: 16}[nw // 10])  # + 10: entry kernel without tiles, + 20: 16-wave entry tiles
    for it in range(3):
        model.kernel_events = {}
        pred = model.predict_spectrogram(spec, chunk=chunk)
        torch.cuda.synchronize()
    tot = {k: sum(a.elapsed_time(b) for a, b in v) for k, v in model.kernel_events.items()}
    if ref is None:
        ref = pred.clone()
    same = bool(torch.equal(pred, ref))
    print(f"mode={nw:2d} bit-identical={same} maxdiff={float((pred - ref).abs().max()):.3g} total={sum(tot.values()):.2f} ms",
          {k: round(v, 2) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:14]}, flush=True)
