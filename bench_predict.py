"""bench.py workload "predict": BASELINE configs[2] -- orcai-V1 inference over one synthetic 1 h recording
(172 800 000 samples @ 48 kHz -> 675 001 frames -> 1 833 snippets), inputs resident in HBM.

One step = front end (STFT/dB/percentile clip/normalise) + ResNetLSTM forward over all 50 %-overlap snippets
+ overlap average + label extraction = everything ``predict_wav`` (predict.py:367-471) does after file decode.
Weights: orcai-V1 architecture with seeded synthetic weights (the trained orcai-v1.keras is absent from the
reference mount, .MISSING_LARGE_BLOBS).
"""

from __future__ import annotations

import os
import time

import numpy as np
import torch

SPEC_PARAM = {"sampling_rate": 48000, "nfft": 512, "n_overlap": 256, "freq_range": [0, 16000], "quantiles": [0.01, 0.999], "duration": 4}
CALLS = ["BR", "BUZZ", "HERDING", "PHS", "SS", "TAILSLAP", "WHISTLE"]
FILTERS = [30, 40, 50, 60]
HBM_PEAK_GBS = 8000.0
MFMA_F32_PEAK_TFLOPS = 157.3
FWD_FLOP_PER_SNIPPET = 0.972e9  # 485.8 M MAC


def _sep_cost(cin, cout, h, w):
    """(algorithmic bytes, flops) per snippet of one fused separable-conv launch: read in, write out; dw + pw MACs."""
    return 4.0 * h * w * (cin + cout), 2.0 * h * w * (9 * cin + cin * cout)


def kernel_costs():
    """label -> (algorithmic HBM bytes, FLOPs) per snippet, fp32 planar activations, each tensor read/written once."""
    shapes = [(736, 171, 16), (368, 86, 30), (184, 43, 40), (92, 22, 50), (46, 11, 60)]
    costs = {"conv0": (4.0 * 736 * 171 * (1 + 16), 2.0 * 736 * 171 * 9 * 16)}
    for b in range(1, 5):
        h, w, cin = shapes[b - 1]
        ho, wo, f = shapes[b]
        costs[f"b{b}/sep_a"] = _sep_cost(cin, f, h, w)
        costs[f"b{b}/sep_b"] = _sep_cost(f, f, h, w)
        # pool + residual: read s (all), read prev at stride 2 (counted as the sampled quarter), write out
        costs[f"b{b}/pool_res"] = (4.0 * (h * w * f + ho * wo * cin + ho * wo * f), 2.0 * ho * wo * cin * f)
    costs["sep_f"] = _sep_cost(60, 36, 46, 11)
    return costs


def measured_traffic(symbol: str, workload: str = "predict"):
    """HBM bytes per launch of `symbol` from the committed rocprofv3 --pmc passes of this same command (FETCH_SIZE and WRITE_SIZE
    in separate runs; gfx950: FETCH_SIZE counts wide reads at half, MI355X_MICROARCH.md HBM section), or None if not collected."""
    import json
    from pathlib import Path

    f = Path(__file__).resolve().parent / "profiles" / "r01_pmc_traffic.json"
    if not f.exists():
        return None
    rec = json.loads(f.read_text()).get(workload, {}).get("kernels", {}).get(symbol)
    return None if rec is None else rec["hbm_bytes_per_launch"]


class PredictWorkload:
    name = "orcai-V1 predict, 1 h synthetic recording @48 kHz, 1833 snippets"
    metric = "audio_seconds_per_s"
    unit = "audio-s/s"
    dtype = "f32"

    def __init__(self, device, rank):
        from bench import synth_pcm_device
        from orcai_amd.architectures import ResNetLSTM
        from orcai_amd.frontend import FrontEnd

        self.device = device
        self.seconds = float(os.environ.get("ORCAI_BENCH_SECONDS", "3600"))
        self.n_samples = int(self.seconds * 48000)
        self.pcm = synth_pcm_device(self.n_samples, 3 + rank, device)
        self.fe = FrontEnd(device)
        self.model = ResNetLSTM((736, 171, 1), 7, FILTERS, 3, 0.0, 128, seed=1)
        self.model.prepare()
        self.chunk = int(os.environ.get("ORCAI_BENCH_CHUNK", "128"))
        self.T = 1 + self.n_samples // 256
        self.n_snippets = (self.T - 736) // 368 + 1
        self.units_per_step = self.seconds
        self.last = None

    def step(self, timed: bool):
        from orcai_amd.predict import aggregate_predictions_device, compute_binary_predictions, compute_labels

        self.model.kernel_events = self.events if timed else None
        spec = self.fe.make_spectrogram(self.pcm, SPEC_PARAM)
        pred = self.model.predict_spectrogram(spec, chunk=self.chunk)
        agg, cnt = aggregate_predictions_device(pred, self.T, 736, 4)
        s, e, n = compute_binary_predictions(agg, cnt, CALLS, 0.5)
        self.last = compute_labels(s, e, n, 16, "*")
        self.model.kernel_events = None

    events: dict = {}

    @staticmethod
    def kernel_symbol(label: str) -> str:
        """HIP kernel symbol a timed label runs as (the name rocprofv3 --kernel-trace --stats reports): the separable-conv and
        pool kernels are templated on the tap size and on ceil(Cout/16) output tiles, so several layers share one symbol."""
        couts = {"b1": 30, "b2": 40, "b3": 50, "b4": 60}
        if label == "conv0":
            return "conv0_kernel<3>"
        if label == "sep_f":
            return "sepconv_kernel<3, 3>"
        blk, _, op = label.partition("/")
        if blk in couts and op in ("sep_a", "sep_b"):
            return f"sepconv_kernel<3, {(couts[blk] + 15) // 16}>"
        if blk in couts and op == "pool_res":
            return f"pool_res_add_kernel<{(couts[blk] + 15) // 16}>"
        return {"gemm": "gemm_kernel", "rec": "lstm_kernel<128>"}.get(op, "dense_sigmoid_kernel" if label == "dense2" else "gemm_kernel")

    def roofline(self):
        """Dominant kernel SYMBOL (as rocprofv3 names it): average launch duration from HIP events recorded on the launch
        stream around every launch of the timed steps; achieved = algorithmic bytes per launch / that duration."""
        per_label_ms = {k: sum(a.elapsed_time(b) for a, b in v) for k, v in self.events.items()}  # ms over all timed steps
        n_steps = max(1, len(self.events["dense2"]))  # the head runs once per step
        costs = kernel_costs()
        sym = {}
        for label, ms in per_label_ms.items():
            d = sym.setdefault(self.kernel_symbol(label), {"ms": 0.0, "launches": 0, "bytes": 0.0, "flops": 0.0, "labels": []})
            launches = len(self.events[label])
            d["ms"] += ms
            d["launches"] += launches
            d["labels"].append(label)
            if label in costs:  # whole-run algorithmic cost of this label: per-snippet cost x snippets x timed steps
                d["bytes"] += costs[label][0] * self.n_snippets * n_steps
                d["flops"] += costs[label][1] * self.n_snippets * n_steps
        dominant = max(sym, key=lambda k: sym[k]["ms"])
        d = sym[dominant]
        avg_ms = d["ms"] / d["launches"]
        out = {"kernel": dominant, "layers": sorted(d["labels"]), "kernel_ms": round(avg_ms, 4), "launches_per_step": d["launches"] // n_steps,
               "snippets_per_launch": round(self.n_snippets * n_steps * len(d["labels"]) / d["launches"], 2)}
        if d["bytes"] > 0:
            achieved = d["bytes"] / d["launches"] / (avg_ms * 1e-3) / 1e9
            out.update({"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                        "algorithmic_bytes_per_launch": round(d["bytes"] / d["launches"]), "traffic": measured_traffic(dominant),
                        "kernel_tflops": round(d["flops"] / d["launches"] / (avg_ms * 1e-3) / 1e12, 2)})
        else:
            out.update({"bound": "mfma", "achieved": None, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": None, "traffic": None})
        model_ms = sum(per_label_ms.values()) / n_steps
        out["model_ms_per_step"] = round(model_ms, 3)
        out["model_tflops"] = round(FWD_FLOP_PER_SNIPPET * self.n_snippets / (model_ms * 1e-3) / 1e12, 2)
        out["per_layer_ms_per_step"] = {k: round(v / n_steps, 3) for k, v in sorted(per_label_ms.items(), key=lambda kv: -kv[1])}
        return out

    def cpu_baseline(self):
        """Oracle (numpy front end + torch-CPU fp32 model) on a bounded sample: 120 s of audio for the front end,
        32 snippets for the model; audio-s/s = sample seconds / (front-end time + model time scaled to the same audio)."""
        from oracle import frontend_ref as F
        from oracle import model_ref as M

        cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("ORCAI_BENCH_CPU_THREADS", "16")))
        torch.set_num_threads(cores)
        rng = np.random.default_rng(7)
        seconds = 120.0
        y = (np.round(np.clip(0.125 * rng.standard_normal(int(seconds * 48000)), -1, 1) * 32767) / 32768).astype(np.float32)
        t0 = time.perf_counter()
        spec, _, _ = F.make_spectrogram_ref(y, {"spectrogram": SPEC_PARAM})
        t_fe = time.perf_counter() - t0
        p = M.random_params(seed=1)
        n_snip = 128  # ~10-15 s of 16-thread CPU work
        snippets = np.stack([spec[(i % 50) * 368 : (i % 50) * 368 + 736] for i in range(n_snip)])[..., None]
        M.forward_ref(p, snippets[:4])  # warm-up
        t0 = time.perf_counter()
        for s in range(0, n_snip, 16):
            M.forward_ref(p, snippets[s : s + 16])
        t_model = time.perf_counter() - t0
        audio_per_snippet = 368 * 256 / 48000.0  # stride between snippets in seconds
        t_total_per_audio_s = t_fe / seconds + (t_model / n_snip) / audio_per_snippet
        return {"value": round(1.0 / t_total_per_audio_s, 1), "unit": self.unit, "cores": cores, "kind": "port",
                "sample": f"oracle (numpy/scipy front end on {seconds:.0f} s: {t_fe:.1f} s; torch-CPU fp32 model on {n_snip} snippets, {cores} threads: {t_model:.1f} s)"}


class TrainWorkload:
    """BASELINE configs[3]: orcai train, orcai-V1 architecture, synthetic snippets resident in HBM, batch 64 per GPU,
    data parallel over RCCL (one flat 3.98 MB gradient all-reduce per step).  One step = forward (training mode) +
    masked BCE + backward + all-reduce + Adam; metric snippets/s."""

    name = "orcai-V1 train step, batch 64 per GPU, synthetic snippets"
    metric = "snippets_per_s"
    unit = "snippets/s"
    dtype = "f32"

    def __init__(self, device, rank):
        from orcai_amd.architectures import ResNetLSTM
        from orcai_amd.training import Trainer

        self.B = int(os.environ.get("ORCAI_BENCH_BATCH", "64"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.model = ResNetLSTM((736, 171, 1), 7, FILTERS, 3, 0.5, 128, seed=1)
        self.trainer = Trainer(self.model, 1e-4, seed=rank)
        g = torch.Generator(device=device)
        g.manual_seed(4 + rank)
        self.x = torch.rand((self.B, 736, 171), device=device, generator=g).view(-1)
        self.y = (torch.rand((self.B, 46, 7), device=device, generator=g) > 0.7).float()
        self.units_per_step = float(self.B)
        self.ev = []

    def step(self, timed: bool):
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        self.trainer.train_step(self.x, 736 * 171, self.B, self.y, world_size=self.world)
        if timed:
            e1.record()
            self.ev.append((e0, e1))

    def roofline(self):
        ms = float(np.mean([a.elapsed_time(b) for a, b in self.ev]))
        flops = 3.0 * FWD_FLOP_PER_SNIPPET * self.B  # fwd + bwd ~ 3x forward
        ach = flops / (ms * 1e-3) / 1e12
        return {"bound": "mfma", "kernel": "train_step (all kernels)", "achieved": round(ach, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / MFMA_F32_PEAK_TFLOPS, 4), "traffic": None, "kernel_ms": round(ms, 3)}

    def cpu_baseline(self):
        from oracle import model_ref as M
        from oracle import train_ref as T

        cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("ORCAI_BENCH_CPU_THREADS", "16")))
        torch.set_num_threads(cores)
        p = M.random_params(seed=1)
        rng = np.random.default_rng(0)
        n = 8
        x = rng.random((n, 736, 171, 1), dtype=np.float32)
        y = (rng.random((n, 46, 7)) > 0.7).astype(np.float32)
        T.loss_and_grads(p, x[:2], y[:2], None, 0.0, dtype=torch.float32)  # warm-up
        done, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 12.0:
            T.loss_and_grads(p, x, y, None, 0.0, dtype=torch.float32)
            done += n
        dt = time.perf_counter() - t0
        return {"value": round(done / dt, 2), "unit": self.unit, "cores": cores, "kind": "port",
                "sample": f"oracle.train_ref.loss_and_grads (torch-CPU autograd forward+backward, fp32, {cores} threads) on {done} snippets in batches of {n}: {dt:.1f} s"}
