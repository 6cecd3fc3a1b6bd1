"""Where a predict step spends its time outside the model kernels: front end, model, overlap average, threshold + runs, labels
(every part followed by a device synchronisation; the sum is therefore a little above the un-instrumented step)."""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT','/root/repo'))
import torch
from bench_predict import PredictWorkload, SPEC_PARAM, CALLS
from orcai_amd.predict import aggregate_predictions_device, compute_binary_predictions, compute_labels

w = PredictWorkload(torch.device("cuda", 0), 0)
for _ in range(3): w.step(False)
w.drain()
torch.cuda.synchronize()
acc = {}
def tick(name, t0):
    torch.cuda.synchronize(); acc[name] = acc.get(name, 0) + time.perf_counter() - t0
N = 10
for _ in range(N):
    t0 = time.perf_counter(); spec = w.fe.make_spectrogram(w.pcm, SPEC_PARAM); tick("front end", t0)
    t0 = time.perf_counter(); pred = w.model.predict_spectrogram(spec, chunk=w.chunk); tick("model", t0)
    t0 = time.perf_counter(); agg, cnt = aggregate_predictions_device(pred, w.T, 736, 4); tick("overlap average", t0)
    t0 = time.perf_counter(); s, e, n = compute_binary_predictions(agg, cnt, CALLS, 0.5); tick("threshold + runs", t0)
    t0 = time.perf_counter(); lab = compute_labels(s, e, n, 16, "*"); tick("labels", t0)
print({k: round(v / N * 1e3, 3) for k, v in acc.items()}, "ms per step; total", round(sum(acc.values()) / N * 1e3, 3))
