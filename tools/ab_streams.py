import os, sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT','/root/repo'))
import torch
from bench import synth_pcm_device, SPEC_PARAM
from orcai_amd.frontend import FrontEnd
from orcai_amd.architectures import ResNetLSTM
dev = torch.device("cuda", 0)
pcm = synth_pcm_device(int(3600 * 48000), 3, dev)
spec = FrontEnd(dev).make_spectrogram(pcm, SPEC_PARAM)
model = ResNetLSTM((736, 171, 1), 7, [30, 40, 50, 60], 3, 0.0, 128, seed=1)
model.prepare()
ref = None
for ns in (1, 2, 3, 4, 1, 2):
    model.head_streams = ns
    for _ in range(3):
        pred = model.predict_spectrogram(spec, chunk=128)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        pred = model.predict_spectrogram(spec, chunk=128)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10 * 1e3
    if ref is None: ref = pred.clone()
    print(f"streams {ns}: {dt:.2f} ms per hour of audio, bit-identical {bool(torch.equal(pred, ref))}", flush=True)
