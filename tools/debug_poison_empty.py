"""Does any kernel of the training step read memory that nothing has written?  Every torch.empty / empty_like of the step is handed out
pre-filled with a poison value (NaN, then 3e38, then -7.0); a step on poisoned workspaces must give the same loss, batch statistics and
gradients as a step on clean ones.  (Found in round 3 through a graph-vs-eager trajectory that diverged only when another test had left
stale blocks in torch's caching allocator.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from orcai_amd.architectures import ResNetLSTM
from orcai_amd.training import Trainer

_empty, _empty_like = torch.empty, torch.empty_like
POISON = [None]


def p_empty(*a, **k):
    t = _empty(*a, **k)
    if POISON[0] is not None and t.is_cuda and t.is_floating_point():
        t.fill_(POISON[0])
    return t


def p_empty_like(x, **k):
    t = _empty_like(x, **k)
    if POISON[0] is not None and t.is_cuda and t.is_floating_point():
        t.fill_(POISON[0])
    return t


torch.empty, torch.empty_like = p_empty, p_empty_like
CFGS = [dict(shape=(32, 12, 1), filters=[10, 20], units=64, labels=3, B=8, prec="f32"), dict(shape=(64, 171, 1), filters=[30, 40], units=64, labels=3, B=4, prec="f32"),
        dict(shape=(64, 43, 1), filters=[10, 20, 30], units=128, labels=7, B=3, prec="f32"), dict(shape=(64, 171, 1), filters=[30, 40], units=64, labels=3, B=4, prec="f16"),
        dict(shape=(32, 12, 1), filters=[10, 20], units=64, labels=3, B=8, prec="f32", arch="ResNet1DConv")]
bad = 0
for cfg in CFGS:
    H, W, _ = cfg["shape"]
    T = H // 2 ** len(cfg["filters"])
    rng = np.random.default_rng(0)
    x = torch.from_numpy(rng.random((cfg["B"], H, W), dtype=np.float32)).cuda().view(-1)
    y = torch.from_numpy((rng.random((cfg["B"], T, cfg["labels"])) > 0.5).astype(np.float32)).cuda()
    res = {}
    for poison in (None, "twin", float("nan"), 3e38, -7.0):  # "twin": a second clean run = the run-to-run spread of the path itself
        # poisoned from the trainer's construction on: persistent workspaces (packed weight copies, partial-sum buffers, LSTM packs) included.
        # Learning rate 0: the sizing step leaves the weights alone, so the compared step starts from identical weights in every run.
        POISON[0] = None if poison == "twin" else poison
        if cfg.get("arch") == "ResNet1DConv":
            from orcai_amd.architectures import ResNet1DConv
            model = ResNet1DConv(cfg["shape"], cfg["labels"], cfg["filters"], 3, 0.3, seed=1)
        else:
            model = ResNetLSTM(cfg["shape"], cfg["labels"], cfg["filters"], 3, 0.3, cfg["units"], seed=1, precision=cfg["prec"])
        tr = Trainer(model, learning_rate=0.0, seed=5)
        tr.train_step(x, H * W, cfg["B"], y)
        out = tr.train_step(x, H * W, cfg["B"], y)
        POISON[0] = None
        a = out["acc"].cpu().numpy()
        res[poison if (poison is None or isinstance(poison, str) or poison == poison) else "nan"] = (a[0] / a[1], tr.P.g.clone(), tr.P.batch_flat.clone(), out["probs"].clone())
    base = res[None]
    for k, r in res.items():
        if k is None:
            continue
        dl = abs(r[0] - base[0])
        dg = float((r[1] - base[1]).abs().max() / base[1].abs().max())
        ds = float((r[2] - base[2]).abs().max() / base[2].abs().max())
        dp = float((r[3] - base[3]).abs().max())
        ok = np.isfinite(dl) and dl <= 1e-6 and dg <= 2e-5 and ds <= 1e-6 and dp <= 1e-6
        bad += not ok
        names = []
        if not ok:
            for n, (o, kk, _) in tr.P.offsets.items():
                d = (r[1][o:o + kk] - base[1][o:o + kk]).abs().max()
                if not bool(d <= 2e-5 * base[1].abs().max()):
                    names.append(n)
        print(cfg["prec"], cfg.get("arch", "ResNetLSTM"), cfg["shape"], cfg["filters"], "poison", k, "dloss", dl, "dgrad", dg, "dstat", ds, "dprobs", dp, "OK" if ok else f"DIFFERS in {names[:12]}", flush=True)
print("poison test:", "CLEAN" if not bad else f"{bad} poisoned runs differ")
