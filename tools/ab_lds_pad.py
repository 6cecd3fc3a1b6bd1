"""EXPERIMENT: occupancy sensitivity of b1/sep_b (sepconv_tile_kernel, x-pooled): unused dynamic LDS (env ORCAI_EXP_LDS_PAD, read once per process) lowers
the workgroups per compute unit (24.5 KB static: 4 per CU; +16 KB: 3; +36 KB: 2).  usage: ab_lds_pad.py [seconds]  (run once per pad value)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth_pcm_device, SPEC_PARAM
from orcai_amd.frontend import FrontEnd
from orcai_amd.architectures import ResNetLSTM

dev = torch.device("cuda", 0)
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 1200.0
pcm = synth_pcm_device(int(secs * 48000), 3, dev)
spec = FrontEnd(dev).make_spectrogram(pcm, SPEC_PARAM)
model = ResNetLSTM((736, 171, 1), 7, [30, 40, 50, 60], 3, 0.0, 128, seed=1)
model.prepare()
for it in range(4):
    model.kernel_events = {}
    pred = model.predict_spectrogram(spec)
    torch.cuda.synchronize()
tot = {k: sum(a.elapsed_time(b) for a, b in v) for k, v in model.kernel_events.items()}
print(f"pad={os.environ.get('ORCAI_EXP_LDS_PAD', '0')} model={sum(tot.values()):.2f} ms", {k: round(v, 2) for k, v in tot.items() if k.startswith("b1") or "conv0" in k}, flush=True)
