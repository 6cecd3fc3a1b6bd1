"""A/B of the row-marching separable-conv kernel (sepconv_rows_kernel) against the shipped streaming kernel on block 1 of orcai-V1:
agreement of the model output and per-layer times (HIP events) on the same spectrogram.  usage: ab_rows.py [seconds] [chunk]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth_pcm_device, SPEC_PARAM
from orcai_amd import _native as N
from orcai_amd.frontend import FrontEnd
from orcai_amd.architectures import ResNetLSTM

dev = torch.device("cuda", 0)
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 900.0
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 128
pcm = synth_pcm_device(int(secs * 48000), 3, dev)
spec = FrontEnd(dev).make_spectrogram(pcm, SPEC_PARAM)
model = ResNetLSTM((736, 171, 1), 7, [30, 40, 50, 60], 3, 0.0, 128, seed=1)
model.prepare()
lib = N.lib()
ref = None
for ring, nr in [(8, 0), (8, 16), (8, 23), (8, 32), (8, 46), (4, 23), (4, 46), (8, 92), (8, 0)]:
    lib.orcai_sepconv_rows(-ring)
    lib.orcai_sepconv_rows(nr)
    for it in range(3):
        model.kernel_events = {}
        pred = model.predict_spectrogram(spec, chunk=chunk)
        torch.cuda.synchronize()
    tot = {k: sum(a.elapsed_time(b) for a, b in v) for k, v in model.kernel_events.items()}
    if ref is None:
        ref = pred.clone()
    print(f"ring={ring} rows_per_wave={nr:3d} max|dp| vs stream kernel={float((pred - ref).abs().max()):.3g} b1/sep_b={tot['b1/sep_b']:.3f} ms total={sum(tot.values()):.2f} ms", flush=True)
lib.orcai_sepconv_rows(0)
