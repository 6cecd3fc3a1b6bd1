"""Does a hipGraph replay of the training step compute what the eager step computes?  (VERDICT r2 item 2 / ADVICE r2.)

Round 2 saw a captured step with the fused-statistics kernels (orcai_sepconv_planes_stats) "drift" from an eager trainer over 24 steps and
switched the fused kernels off under capture.  A drift between two TRAJECTORIES does not separate a wrong replay from the amplification
of float-atomic reordering by a dropout network under Adam, so this tool measures both:

  A. noise floor:   two EAGER trainers (same seed, same batches, fused statistics) against each other over 24 steps;
  B. trajectories:  eager vs graph, fused statistics on in both, and eager vs graph with the separate statistics pass;
  C. step by step:  before every step the eager trainer's complete state (weights, Adam moments, BatchNorm moving statistics, step
                    counter) is copied into the graph trainer; both then take ONE step on the same batch: loss, every gradient tensor,
                    every batch statistic and the updated weights are compared.  A replay that computes something else than the eager step
                    shows up here at the replay where it happens, by tensor name -- without any amplification.

usage: python tools/debug_graph_divergence.py [steps]   (small network; ~1 min)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from orcai_amd.architectures import ResNetLSTM
from orcai_amd.training import Trainer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 24
CFGS = [dict(shape=(32, 12, 1), filters=[10, 20], units=64, labels=3, B=8, lr=3e-3, drop=0.3),
        dict(shape=(64, 171, 1), filters=[30, 40], units=64, labels=3, B=4, lr=3e-3, drop=0.3)]  # the second one reaches the strip-tile kernel (two strips of 62 columns)


def make(cfg, fused, fused_capture=True):
    tr = Trainer(ResNetLSTM(cfg["shape"], cfg["labels"], cfg["filters"], 3, cfg["drop"], cfg["units"], seed=1), learning_rate=cfg["lr"], seed=5)
    tr.trunk.stats_in_epilogue = fused
    tr.trunk.fused_stats_under_capture = fused_capture
    return tr


def batches(cfg):
    rng = np.random.default_rng(0)
    H, W, _ = cfg["shape"]
    T = H // 2 ** len(cfg["filters"])
    x = rng.random((4, cfg["B"], H, W), dtype=np.float32)
    y = (rng.random((4, cfg["B"], T, cfg["labels"])) > 0.5).astype(np.float32)
    return [torch.from_numpy(x[b]).cuda().view(-1) for b in range(4)], [torch.from_numpy(y[b]).cuda() for b in range(4)], H * W


def trajectory(cfg, mode, fused):
    tr = make(cfg, fused)
    xs, ys, stride = batches(cfg)
    losses = []
    for s in range(steps):
        fn = tr.train_step if mode == "eager" else tr.train_step_graphed
        a = fn(xs[s % 4], stride, cfg["B"], ys[s % 4])["acc"].cpu().numpy()
        losses.append(a[0] / a[1])
    return np.array(losses), tr.P.w.cpu().numpy().copy()


def copy_state(src, dst):
    dst.P.w.copy_(src.P.w); dst.P.m.copy_(src.P.m); dst.P.v.copy_(src.P.v); dst.P.stats_flat.copy_(src.P.stats_flat)
    dst.P.batch_flat.copy_(src.P.batch_flat)
    dst.counter.copy_(src.counter); dst.step_count = src.step_count
    dst.lr = src.lr


def named_max_rel(P, a, b):
    out = {}
    for n, (o, k, _) in P.offsets.items():
        d = float((a[o:o + k] - b[o:o + k]).abs().max())
        sc = float(a[o:o + k].abs().max())
        out[n] = d / max(sc, 1e-12)
    return out


for cfg in CFGS:
    print("=== config", {k: cfg[k] for k in ("shape", "filters", "B")}, flush=True)
    e1, w1 = trajectory(cfg, "eager", True)
    e2, w2 = trajectory(cfg, "eager", True)
    g1, wg = trajectory(cfg, "graph", True)
    es, ws = trajectory(cfg, "eager", False)
    gs, wgs = trajectory(cfg, "graph", False)
    print(f"A noise floor   eager(fused) vs eager(fused):   max|dloss| {np.abs(e1 - e2).max():.3e}  max|dw| {np.abs(w1 - w2).max():.3e}")
    print(f"B trajectories  eager(fused) vs graph(fused):   max|dloss| {np.abs(e1 - g1).max():.3e}  max|dw| {np.abs(w1 - wg).max():.3e}")
    print(f"B trajectories  eager(sep)   vs graph(sep):     max|dloss| {np.abs(es - gs).max():.3e}  max|dw| {np.abs(ws - wgs).max():.3e}")
    print(f"  (fused vs separate, both eager:               max|dloss| {np.abs(e1 - es).max():.3e}  max|dw| {np.abs(w1 - ws).max():.3e})")
    # divergence is an event (a ReLU / arg-max / dropout-scaled unit flips), so compare DISTRIBUTIONS: first step with |dloss| > 1e-4 over several pairs
    def first_over(a, b, thr=1e-4):
        d = np.abs(a - b)
        return int(np.argmax(d > thr)) if (d > thr).any() else -1
    ee = [first_over(trajectory(cfg, "eager", True)[0], trajectory(cfg, "eager", True)[0]) for _ in range(4)]
    eg = [first_over(trajectory(cfg, "eager", True)[0], trajectory(cfg, "graph", True)[0]) for _ in range(4)]
    print(f"  first step with |dloss| > 1e-4 (-1 = never) over 4 fresh pairs: eager-eager {ee}   eager-graph {eg}")
    print("  per-step |dloss| eager-eager:", np.array2string(np.abs(e1 - e2), precision=1, max_line_width=250))
    print("  per-step |dloss| eager-graph:", np.array2string(np.abs(e1 - g1), precision=1, max_line_width=250), flush=True)

    # C. one step from a common state, every step
    xs, ys, stride = batches(cfg)
    E, G, E2 = make(cfg, True), make(cfg, True), make(cfg, True)
    G.train_step_graphed(xs[0], stride, cfg["B"], ys[0])  # warm-up + capture + first replay
    worst = {"loss": 0.0, "grad": 0.0, "stat": 0.0, "w": 0.0, "grad_ee": 0.0, "stat_ee": 0.0}
    for s in range(steps):
        copy_state(E, G); copy_state(E, E2)
        oe = E.train_step(xs[s % 4], stride, cfg["B"], ys[s % 4])
        ge, be = E.P.g.clone(), E.P.batch_flat.clone()
        og = G.train_step_graphed(xs[s % 4], stride, cfg["B"], ys[s % 4])
        o2 = E2.train_step(xs[s % 4], stride, cfg["B"], ys[s % 4])
        ae, ag = oe["acc"].cpu().numpy(), og["acc"].cpu().numpy()
        dl = abs(ae[0] / ae[1] - ag[0] / ag[1])
        rel = named_max_rel(E.P, ge, G.P.g)
        rel2 = named_max_rel(E.P, ge, E2.P.g)
        # biases in front of a BatchNorm have an identically zero gradient (never written): skip 0/0
        big = {n: v for n, v in rel.items() if v > 1e-3 and float(ge[E.P.offsets[n][0]:E.P.offsets[n][0] + E.P.offsets[n][1]].abs().max()) > 1e-9}
        ds = float((be - G.P.batch_flat).abs().max() / be.abs().max())
        ds2 = float((be - E2.P.batch_flat).abs().max() / be.abs().max())
        dw = float((E.P.w - G.P.w).abs().max())
        gmax = max(v for n, v in rel.items() if float(ge[E.P.offsets[n][0]:E.P.offsets[n][0] + E.P.offsets[n][1]].abs().max()) > 1e-9)
        gmax2 = max(v for n, v in rel2.items() if float(ge[E.P.offsets[n][0]:E.P.offsets[n][0] + E.P.offsets[n][1]].abs().max()) > 1e-9)
        worst = {"loss": max(worst["loss"], dl), "grad": max(worst["grad"], gmax), "stat": max(worst["stat"], ds), "w": max(worst["w"], dw),
                 "grad_ee": max(worst["grad_ee"], gmax2), "stat_ee": max(worst["stat_ee"], ds2)}
        if big or s < 3 or s == steps - 1:
            print(f"C step {s:2d}: |dloss| {dl:.2e}  grad rel (graph) {gmax:.2e} (eager twin {gmax2:.2e})  batch-stat rel {ds:.2e} (twin {ds2:.2e})  max|dw| {dw:.2e}  counters {int(E.counter)} {int(G.counter)}"
                  + (f"  OUTLIERS {big}" if big else ""), flush=True)
    print("C worst over", steps, "state-synchronised steps:", {k: f"{v:.2e}" for k, v in worst.items()}, flush=True)

# D. free-running trajectories, all named state compared after every step: graph (state restored after its warm-up) and an eager twin
# against an eager trainer.  Shows WHERE a trajectory difference enters (weights / Adam moments / which step), and what two eager
# trainers show at the same step.
for cfg in CFGS[:1]:
    xs, ys, stride = batches(cfg)
    E, E2, G = make(cfg, True), make(cfg, True), make(cfg, True)
    for s in range(8):
        E.train_step(xs[s % 4], stride, cfg["B"], ys[s % 4]); E2.train_step(xs[s % 4], stride, cfg["B"], ys[s % 4]); G.train_step_graphed(xs[s % 4], stride, cfg["B"], ys[s % 4])
        row = {}
        for tag, O in (("graph", G), ("twin", E2)):
            row[tag] = {k: float((getattr(E.P, k) - getattr(O.P, k)).abs().max()) for k in ("w", "m", "v", "g", "stats_flat", "batch_flat")}
            row[tag]["counter"] = int(O.counter) - int(E.counter)
        print(f"D step {s}:", {t: {k: (f"{v:.1e}" if isinstance(v, float) else v) for k, v in r.items()} for t, r in row.items()}, flush=True)
        if s == 0:
            d = (E.P.w - G.P.w).abs()
            i = int(d.argmax())
            name = [n for n, (o, k, _) in E.P.offsets.items() if o <= i < o + k][0]
            print("   largest weight difference after step 0 in", name, "w", float(E.P.w[i]), float(G.P.w[i]), "g", float(E.P.g[i]), float(G.P.g[i]), "m", float(E.P.m[i]), float(G.P.m[i]),
                  "v", float(E.P.v[i]), float(G.P.v[i]))
