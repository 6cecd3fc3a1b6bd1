import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from bench import synth_pcm_device, SPEC_PARAM
from orcai_amd.frontend import FrontEnd
from orcai_amd.architectures import ResNetLSTM
from orcai_amd.predict import aggregate_predictions_device
log = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "debug_predict.log"), "a")
def P(*a):
    print(*a, file=log, flush=True); print(*a, flush=True)
if os.environ.get("ORCAI_POOL_FUSED"):  # A/B of the fused block tail under a profiler (0 = the two launches)
    from orcai_amd import _native as N
    N.lib().orcai_pool_fused(int(os.environ["ORCAI_POOL_FUSED"]))
dev = torch.device("cuda", 0)
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 64
t0 = time.time(); pcm = synth_pcm_device(int(secs * 48000), 3, dev); torch.cuda.synchronize(); P("synth", time.time() - t0)
fe = FrontEnd(dev)
model = ResNetLSTM((736, 171, 1), 7, [30, 40, 50, 60], 3, 0.0, 128, seed=1); model.prepare()
for it in range(3):
    t0 = time.time(); spec = fe.make_spectrogram(pcm, SPEC_PARAM); torch.cuda.synchronize(); P("frontend", time.time() - t0, tuple(spec.shape))
    model.kernel_events = {}
    t0 = time.time(); pred = model.predict_spectrogram(spec, chunk=chunk); torch.cuda.synchronize(); P("model", time.time() - t0, tuple(pred.shape))
    tot = {k: sum(a.elapsed_time(b) for a, b in v) for k, v in model.kernel_events.items()}
    P({k: round(v, 2) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])})
    t0 = time.time(); agg, cnt = aggregate_predictions_device(pred, spec.shape[0], 736, 4); P("aggregate", time.time() - t0, agg.shape)
