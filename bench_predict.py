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


def _sep_cost(cin, cout, h, w, w_out=None):
    """(algorithmic bytes, flops) per snippet of one fused separable-conv launch: the input planes read once, the output AS THE
    KERNEL WRITES IT written once (w_out = ceil(w/2) columns for the x-pooled layout of a block's second conv); dw + pw MACs."""
    return 4.0 * h * (w * cin + (w if w_out is None else w_out) * cout), 2.0 * h * w * (9 * cin + cin * cout)


def kernel_costs():
    """label -> (algorithmic HBM bytes, FLOPs) per snippet: every tensor a kernel reads or writes, once, at its true channel count
    and in the shape the kernel actually moves (x-pooled outputs of */sep_b, x-pooled inputs of */pool_res, the compact (2i, 2j)
    subsample the fused entry path hands to block 1's residual conv).  fp32."""
    shapes = [(736, 171, 16), (368, 86, 30), (184, 43, 40), (92, 22, 50), (46, 11, 60)]
    costs = {"conv0": (4.0 * 736 * 171 * (1 + 16), 2.0 * 736 * 171 * 9 * 16)}
    for b in range(1, 5):
        h, w, cin = shapes[b - 1]
        ho, wo, f = shapes[b]
        wx = (w + 1) // 2
        costs[f"b{b}/sep_a"] = _sep_cost(cin, f, h, w)
        costs[f"b{b}/sep_b"] = _sep_cost(f, f, h, w, w_out=wx)
        # pool + residual: read the x-pooled s, read prev at the sampled (2i, 2j) pixels, write out
        costs[f"b{b}/pool_res"] = (4.0 * (h * wx * f + ho * wo * cin + ho * wo * f), 2.0 * ho * wo * cin * f)
        # the second conv with the block's tail in its epilogue (orcai_sepconv_pool_res): a read once, prev at the sampled pixels, the block output written once --
        # the x-pooled tensor of the two-launch path (h * wx * f, written and read back) does not exist
        costs[f"b{b}/sep_b+pool_res"] = (4.0 * (h * w * f + ho * wo * cin + ho * wo * f), costs[f"b{b}/sep_b"][1] + costs[f"b{b}/pool_res"][1])
    costs["sep_f"] = _sep_cost(60, 36, 46, 11)
    # fused entry (orcai_conv0_sepconv): read the 1-channel snippet, write a1 and the (2i, 2j) subsample of the entry activation
    costs["conv0+b1/sep_a"] = (4.0 * (736 * 171 * (1 + 30) + 368 * 86 * 16), costs["conv0"][1] + costs["b1/sep_a"][1])
    return costs


def traffic_file():
    """The newest round's PMC traffic table (profiles/rNN_pmc_traffic.json, written fresh by tools/make_pmc_traffic.py from the --pmc passes of
    tools/evidence_rNN.sh), or None."""
    from pathlib import Path

    found = sorted((Path(__file__).resolve().parent / "profiles").glob("r*_pmc_traffic.json"))
    return found[-1] if found else None


def measured_issue(symbol: str):
    """SIMD occupancy / busy shares of `symbol` from the newest committed counter pass at the benchmark's launch size (tools/pmc_occupancy.sh ->
    profiles/rNN_pmc_occupancy_fused_tail.json): resident waves per SIMD, share of SIMD cycles the vector ALU / the matrix pipe was occupied (SQ_* count
    quad-cycles, GRBM_GUI_ACTIVE sums the 8 XCDs).  None when there is no such file or it does not know the symbol."""
    import json
    from pathlib import Path

    found = sorted((Path(__file__).resolve().parent / "profiles").glob("r*_pmc_occupancy_fused_tail.json"))
    if not found:
        return None
    d = json.loads(found[-1].read_text())
    try:
        g = lambda c: d[c][symbol]["mean_per_launch"]  # noqa: E731
        dur = g("GRBM_GUI_ACTIVE") / 8
        return {"waves_per_simd": round(g("SQ_WAVE_CYCLES") * 4 / 1024 / dur, 2), "valu_busy": round(g("SQ_ACTIVE_INST_VALU") * 4 / 1024 / dur, 3),
                "mfma_busy": round(g("SQ_VALU_MFMA_BUSY_CYCLES") / 1024 / dur, 3), "wait_share_of_wave_life": round(g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"), 2),
                "source": "profiles/" + found[-1].name}
    except KeyError:
        return None


def measured_traffic(symbol: str, workload: str = "predict"):
    """HBM bytes per launch of `symbol` from the committed rocprofv3 --pmc passes of this same command (FETCH_SIZE and WRITE_SIZE
    in separate runs; gfx950: FETCH_SIZE counts wide reads at half, MI355X_MICROARCH.md HBM section).  None only when no table has been
    collected at all; a table that does not know the symbol is stale evidence and says so on stderr (tests/test_capi_symbols.py holds
    kernel_symbol() and the newest table together, so the CPU suite fails first)."""
    import json
    import sys

    f = traffic_file()
    if f is None:
        return None
    rec = json.loads(f.read_text()).get(workload, {}).get("kernels", {}).get(symbol)
    if rec is None:
        print(f"bench: {f.name} has no PMC traffic for kernel symbol {symbol!r} of workload {workload!r}: stale profile, re-run the newest tools/evidence_rNN_*.sh", file=sys.stderr)
        return None
    return rec["hbm_bytes_per_launch"]


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
        self.model.kernel_event_labels = set(self.DOMINANT)

    def step(self, timed: bool):
        """One recording, as predict()'s table mode runs it (orcai_amd/predict.py: _launch_recording / _finish_recording): the GPU half
        (front end, model, overlap average, copy to pinned host memory) is queued, then the host half of the PREVIOUS recording
        (threshold, runs, label table: ~1 ms of numpy / pandas) runs beside it.  drain() finishes the last one; bench.py calls it
        inside the timed region, so K steps = K complete recordings."""
        from orcai_amd.predict import aggregate_predictions_device

        self.model.kernel_events = self.events if timed else None
        spec = self.fe.make_spectrogram(self.pcm, SPEC_PARAM)
        pred = self.model.predict_spectrogram(spec, chunk=self.chunk)
        pending = aggregate_predictions_device(pred, self.T, 736, 4, wait=False)
        self.model.kernel_events = None
        self.drain()
        self.in_flight = pending

    in_flight = None

    def drain(self):
        from orcai_amd.predict import compute_binary_predictions, compute_labels

        if self.in_flight is not None:
            agg, cnt = self.in_flight.result()
            self.in_flight = None
            s, e, n = compute_binary_predictions(agg, cnt, CALLS, 0.5)
            self.last = compute_labels(s, e, n, 16, "*")

    events: dict = {}

    @staticmethod
    def kernel_symbol(label: str) -> str:
        """HIP kernel symbol a timed label runs as (the name rocprofv3 --kernel-trace --stats reports): the separable-conv and
        pool kernels are templated on the tap size and on ceil(Cout/16) output tiles, so several layers share one symbol."""
        from orcai_amd import _native as N

        couts = {"b1": 30, "b2": 40, "b3": 50, "b4": 60}
        cins = {"b1": 16, "b2": 30, "b3": 40, "b4": 50}
        widths = {"b1": 171, "b2": 86, "b3": 43, "b4": 22}
        if label == "conv0":
            return "conv0_kernel<3>"
        if label == "conv0+b1/sep_a":
            tile = N.lib().orcai_entry_tile(-1)
            return f"conv0_sep_tile_kernel<2, {tile}>" if tile else "conv0_sep_kernel<2>"
        if label == "sep_f":
            return "sepconv_kernel<3, 3>"  # Keras-reshape output layout: not a streaming shape
        blk, _, op = label.partition("/")
        if blk in couts and op == "sep_b+pool_res":  # <CQ = input quads, RELU on load>
            return f"sepconv_pool_march_kernel<{(couts[blk] + 3) // 4}, false>"
        if blk in couts and op in ("sep_a", "sep_b"):
            cin, cout = (cins[blk] if op == "sep_a" else couts[blk]), couts[blk]
            mt, cqr = (cout + 15) // 16, (cin + 3) // 4
            mode = N.lib().orcai_sepconv_tile_mode(-1)
            xp, relu = ("true", "false") if op == "sep_b" else ("false", "true")  # sep_b: x-pooled output; sep_a: ReLU on load
            val = 60 if op == "sep_b" else 62
            nstrip = -(-widths[blk] // val)
            if mode == 1 and mt == 2 and cqr <= 8 and nstrip >= 2 and widths[blk] * 100 >= nstrip * val * 85:  # launch_sepconv_impl's rule
                return f"sepconv_tile_kernel<2, {4 if cqr <= 4 else 8}, {xp}, {relu}, 8, false, 0, false>"  # <MT, CQ, XP, RELU, TR, UOUT, EPI, BNIN>
            if mode >= 1:
                return f"sepconv_ftile_kernel<{mt}, {xp}, {relu}, false, 8, 0, false>"  # <MT, XP, RELU, UOUT, NWV, EPI, BNIN>
            return f"sepconv_kernel<3, {mt}>"
        if blk in couts and op == "pool_res":  # inference: the x-pooled fast path, <MT, VERT>: stacked tiles where the pooled plane is >= 40 columns wide
            vert = N.lib().orcai_pool_vertical(-1) and (widths[blk] + 1) // 2 >= 40
            return f"pool_res_add_x_kernel<{(couts[blk] + 15) // 16}, {'true' if vert else 'false'}>"
        return {"gemm": "gemm_kernel", "rec": "lstm_split_kernel<128>"}.get(op, "dense_sigmoid_kernel" if label == "dense2" else "gemm_kernel")

    # the layers bracketed with HIP events inside the timed steps: the two heaviest launches of block 1; the second one
    # (sepconv_tile_kernel<2, 8, true, false, 8, false, 0, false>) is the top symbol of rocprofv3 --stats for this workload
    DOMINANT = ("conv0+b1/sep_a", "b1/sep_a", "b1/sep_b", "b1/sep_b+pool_res")

    def roofline(self):
        """Dominant kernel SYMBOL (as rocprofv3 names it): average launch duration from HIP events recorded on the launch stream
        around its launches inside the timed steps; achieved = algorithmic bytes per launch / that duration.  Only those launches
        are bracketed in the timed region (an event pair costs ~15 us of queue time); the per-layer table comes from one more
        step, after the timed region, with every launch bracketed."""
        timed_events = self.events
        n_steps = max(1, max(len(timed_events.get(k, [])) for k in ("b1/sep_b", "b1/sep_b+pool_res")) // max(1, -(-self.n_snippets // self.chunk)))
        self.events, self.model.kernel_event_labels = {}, None
        self.step(True)  # all layers bracketed, outside the timed region
        self.drain()
        torch.cuda.synchronize()
        table = {k: sum(a.elapsed_time(b) for a, b in v) for k, v in self.events.items()}
        self.events = timed_events
        self.model.kernel_event_labels = set(self.DOMINANT)
        per_label_ms = {k: sum(a.elapsed_time(b) for a, b in v) for k, v in self.events.items()}  # ms over all timed steps
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
            if out["traffic"]:  # PMC bytes (2*FETCH_SIZE + WRITE_SIZE of the committed passes) over the algorithmic bytes: > 1 = padding / re-reads
                out["traffic_ratio"] = round(out["traffic"] / out["algorithmic_bytes_per_launch"], 3)
            issue = measured_issue(dominant)
            if issue:  # what the kernel is bound by when it is not its bytes: vector ALU and f32 MFMA share one datapath
                out["simd_issue"] = issue
        else:
            out.update({"bound": "mfma", "achieved": None, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": None, "traffic": None})
        by_symbol = {}
        for label, ms in table.items():
            by_symbol[self.kernel_symbol(label)] = by_symbol.get(self.kernel_symbol(label), 0.0) + ms
        out["top_symbol_of_full_table"] = max(by_symbol, key=by_symbol.get)  # must equal "kernel": the bracketed symbol is the dominant one
        model_ms = sum(table.values())
        out["model_ms_per_step"] = round(model_ms, 3)
        out["model_tflops"] = round(FWD_FLOP_PER_SNIPPET * self.n_snippets / (model_ms * 1e-3) / 1e12, 2)
        out["per_layer_ms_per_step"] = {k: round(v, 3) for k, v in sorted(table.items(), key=lambda kv: -kv[1])}
        return out

    def cpu_baseline(self):
        """BASELINE.md section 3, bounded to ~25 s of CPU work.  The CPU restatement (oracle/: numpy/scipy front end, torch-CPU fp32 model,
        numpy/pandas post-processing -- NOT librosa/Keras, which cannot be installed here), timed on this box's host cores:
          CPU-1  configs[0], one 60 s recording at 48 kHz (29 snippets), all cores: wall split STFT+dB / normalise / model / post
                 (file decode is not part of the GPU step either); the model also on ONE thread (8 snippets, scaled to 29);
          CPU-3  configs[2] extrapolated as the protocol allows: front end per audio second from the 60 s recording + model time per
                 snippet from 100 snippets (all cores) -> `value`, audio-s/s of a 1 h recording."""
        from oracle import frontend_ref as F
        from oracle import model_ref as M
        from oracle import postprocess_ref as P

        from orcai_amd.synthetic import pcm16_to_float, synth_recording

        cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("ORCAI_BENCH_CPU_THREADS", "16")))
        torch.set_num_threads(cores)
        seconds = 60.0
        y = pcm16_to_float(synth_recording(seconds, 48000, seed=20250620))
        p = M.random_params(seed=1)
        t = {}
        t0 = time.perf_counter()
        db, freqs, times = F.calculate_spectrogram_ref(y, SPEC_PARAM)
        t["stft_db"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        spec = F.preprocess_spectrogram_ref(db, freqs, SPEC_PARAM)
        t["normalise"] = time.perf_counter() - t0
        snippets = P.slice_snippets(spec, 736)
        M.forward_ref(p, snippets[:2])  # warm-up
        t0 = time.perf_counter()
        pred = np.concatenate([M.forward_ref(p, snippets[s : s + 16]) for s in range(0, len(snippets), 16)])
        t["model"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        agg, cnt = P.aggregate_predictions_ref(pred, spec.shape[0], 736, 4, 7)
        st, en, nm = P.compute_binary_predictions_ref(agg, cnt, CALLS)
        P.labels_to_tsv_ref(P.compute_labels_ref(st, en, nm, 16, "*"), times[1] - times[0])
        t["post"] = time.perf_counter() - t0
        cpu1 = {k: round(v, 3) for k, v in t.items()}
        cpu1.update({"snippets": int(len(snippets)), "audio_s": seconds, "threads": cores, "audio_s_per_s": round(seconds / sum(t.values()), 2)})
        torch.set_num_threads(1)
        t0 = time.perf_counter()
        M.forward_ref(p, snippets[:8])
        t1 = (time.perf_counter() - t0) / 8 * len(snippets)
        torch.set_num_threads(cores)
        cpu1["model_one_thread_scaled"] = round(t1, 2)
        cpu1["audio_s_per_s_one_thread"] = round(seconds / (t["stft_db"] + t["normalise"] + t1 + t["post"]), 2)
        # CPU-3: 100 snippets of the model on all cores (the 29 above + 71 more), front end and post-processing scaled by audio time
        n_more = 71
        more = np.stack([spec[(i % 20) * 368 : (i % 20) * 368 + 736] for i in range(n_more)])[..., None]
        t0 = time.perf_counter()
        for s0 in range(0, n_more, 16):
            M.forward_ref(p, more[s0 : s0 + 16])
        per_snippet = (t["model"] + time.perf_counter() - t0) / (len(snippets) + n_more)
        fe_per_audio_s = (t["stft_db"] + t["normalise"] + t["post"]) / seconds
        hour = 3600.0 * fe_per_audio_s + 1833 * per_snippet
        return {"value": round(3600.0 / hour, 1), "unit": self.unit, "cores": cores, "kind": "port",
                "sample": (f"CPU restatement (numpy/scipy + torch-CPU fp32; not librosa/Keras): 1 h extrapolated from the 60 s recording's front end / post-processing "
                           f"({fe_per_audio_s * 1e3:.2f} ms per audio-s) and {len(snippets) + n_more} snippets of the model on {cores} threads ({per_snippet * 1e3:.0f} ms per snippet)"),
                "config1_60s_wall_split_s": cpu1}


def _per_rank_batch(world: int) -> int:
    """64 snippets per rank ("replicate": weak scaling), or the reference's GLOBAL batch of 64 cut into `world` slices (ORCAI_BENCH_DP_BATCH=split, bench.py
    --dp-batch split: MirroredStrategy's contract, hpsearch.py:170-205 -- the same number of optimiser steps per epoch as one GPU, strong scaling)."""
    B = int(os.environ.get("ORCAI_BENCH_BATCH", "64"))
    if os.environ.get("ORCAI_BENCH_DP_BATCH", "replicate") == "split":
        if B % world:
            raise SystemExit(f"bench: --dp-batch split needs the global batch {B} divisible by the {world} ranks")
        B //= world
    return B


KERNEL_ONLY_BRACKETS = ("orcai_bn_bwd_pointwise_wgrad", "orcai_h_bn_bwd_pointwise_wgrad")


class _RawEvent:
    """A HIP event owned through the C ABI (orcai_event_*), with torch.cuda.Event's elapsed_time(): what orcai_profile_bracket takes."""

    def __init__(self, lib):
        import ctypes

        self._lib, h = lib, ctypes.c_void_p()
        rc = lib.orcai_event_create(ctypes.byref(h))
        if rc != 0:
            raise RuntimeError(f"orcai_event_create: HIP error {rc}")
        self.handle = h

    def elapsed_time(self, other) -> float:
        import ctypes

        ms = ctypes.c_float()
        rc = self._lib.orcai_event_elapsed_ms(self.handle, other.handle, ctypes.byref(ms))
        if rc != 0:
            raise RuntimeError(f"orcai_event_elapsed_ms: HIP error {rc}")
        return float(ms.value)

    def __del__(self):
        try:
            self._lib.orcai_event_destroy(self.handle)
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass


def bracket_overhead_ms() -> float:
    """What an EMPTY pair of torch events reads on the current stream (median of 30 pairs): a bracket around a launcher call contains the second event's own
    dispatch, ~8-10 us -- nothing beside a 0.8 ms kernel, but ten of them make a ten-launch symbol outrank a one-launch symbol of the same kernel time.
    The fully bracketed tables subtract it per call (not from the kernel-only brackets, whose events the launcher records back to back with its kernel)."""
    pairs = []
    torch.cuda.synchronize()
    for _ in range(30):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        b.record()
        pairs.append((a, b))
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in pairs]))


def _net(name, e0, e1, overhead):
    t = e0.elapsed_time(e1)
    return t if name in KERNEL_ONLY_BRACKETS else max(t - overhead, 0.0)


class _TimedLib:
    """Wraps the ctypes library handle of a trainer (bench instrumentation only; the product path calls the handle directly).
    mode "dominant": only the launches of the step's dominant kernel symbol -- outer_reduce_kernel, the pointwise / residual weight
    gradients (13 launches per step), the top row of rocprofv3 --stats since the block-1 separable convolutions moved to the LDS-tile
    kernels -- are bracketed by HIP events on the launch stream: an event pair costs ~15 us of queue time, and a training step has
    ~250 short launches, so bracketing all of them would slow the step by 20 %.  mode "all": every orcai_* launcher (used for two
    extra steps AFTER the timed region to report where the time goes)."""

    def __init__(self, lib, is_dominant=None):
        self._lib, self.events, self.mode, self.order = lib, None, "dominant", []  # order: (launcher, index into events[launcher]) in call order
        if is_dominant is not None:
            self.is_dominant = is_dominant

    @staticmethod
    def is_dominant(name, args):
        return name == TRAIN_DOMINANT_LAUNCHER and _train_call_symbol(name, args) == TRAIN_DOMINANT_SYMBOL

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if not name.startswith("orcai_"):
            return fn

        def call(*args):
            if self.events is None or (self.mode == "dominant" and not self.is_dominant(name, args)):
                return fn(*args)
            if name in KERNEL_ONLY_BRACKETS:
                # the launcher itself records the pair around its MAIN kernel (orcai_profile_bracket): the call enqueues two small kernels after it, and
                # a bracket around the whole call reads ~35 us above the kernel duration rocprofv3 lists
                e0, e1 = _RawEvent(self._lib), _RawEvent(self._lib)
                self._lib.orcai_profile_bracket(e0.handle, e1.handle)
                rc = fn(*args)
                if rc == -2:
                    self._lib.orcai_profile_bracket(None, None)
                else:
                    self.events.setdefault(name, []).append((e0, e1, args))
                    self.order.append((name, len(self.events[name]) - 1))
                return rc
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*args)
            e1.record()
            if rc != -2:  # ORCAI_E_UNSUPPORTED: the launcher refused before touching anything and the caller runs its fallback -- not a launch
                self.events.setdefault(name, []).append((e0, e1, args))
                self.order.append((name, len(self.events[name]) - 1))
            return rc

        return call


# The launcher bracketed inside the timed training steps: the top row of the last fully bracketed table / rocprofv3 summary
# (profiles/r02_train_rocprofv3_kernel_stats.csv: bn_bwd_pw_kernel<*> 15.6 % of the step summed over its instantiations).  roofline()
# re-derives the dominant launcher from its own table every run and reports whether the two agree.
TRAIN_DOMINANT_LAUNCHER = "orcai_bn_bwd_pointwise_wgrad"
TRAIN_DOMINANT_SYMBOL = "bn_bwd_pw_wgrad_kernel<2, 2, 4>"  # the top row of profiles/r03_train_rocprofv3_kernel_stats.csv: block 1's second separable conv
LAUNCHER_KERNELS = {"orcai_bn_bwd_pointwise": "bn_bwd_pw_kernel<MT>", "orcai_outer_reduce": "outer_reduce_kernel<NP>", "orcai_dw_wgrad": "dw_wgrad_kernel<3>",
                    "orcai_sepconv_planes_stats": "sepconv_tile_kernel / sepconv_ftile_kernel<..., STATS = true>", "orcai_sepconv_planes_u": "sepconv_*_kernel",
                    "orcai_bn_planes_stats": "planes_sums_kernel", "orcai_bn_planes_apply": "bn_planes_apply_kernel",
                    "orcai_bn_bwd_pointwise_wgrad": "bn_bwd_pw_wgrad_kernel<MT, NT, 4>", "orcai_sepconv_planes_epi": "sepconv_tile_kernel / sepconv_ftile_kernel<..., EPI = 2 | 3>",
                    "orcai_sepconv_planes_stats_bn": "sepconv_tile_kernel / sepconv_ftile_kernel<..., EPI = 1, BNIN>", "orcai_dw_wgrad_bn": "dw_wgrad_kernel<3, true>",
                    "orcai_dw_bwd_fused": "dw_bwd_march_kernel<SW, EPI, BNIN>"}


def _train_call_symbol(name, a):
    """The kernel symbol (template instantiation, as rocprofv3 --kernel-trace --stats prints it) a launcher call runs, for the launchers whose
    launches top the training step; the family name of LAUNCHER_KERNELS otherwise.  rocprofv3 ranks SYMBOLS, so the roofline table does too."""
    if name == "orcai_bn_bwd_pointwise_wgrad":  # dy,v,u,B,C,H,W,ksize,...,wt,Cin,...: MT = conv-input tiles, NT = conv-output tiles, four-wave workgroups up to 4 tiles
        mt, nt = (a[19] + 15) // 16, (a[4] + 15) // 16
        return f"bn_bwd_pw_wgrad_kernel<{mt}, {nt}, {4 if mt + nt <= 4 else 2}>"
    if name == "orcai_dw_bwd_fused":  # x,du,B,C,H,W,relu_in,dw_rev,dr,dW,epi,bn_mean,...: strip width as the launcher picks it
        W, epi, bn = a[5], a[10], a[11] is not None
        best, lanes = 64, ((W + 61) // 62) * 64
        if ((W + 29) // 30) * 32 < lanes:
            best, lanes = 32, ((W + 29) // 30) * 32
        if ((W + 13) // 14) * 16 < lanes:
            best = 16
        return f"dw_bwd_march_kernel<{best}, {epi}, {'true' if bn else 'false'}, false>"
    if name == "orcai_bn_bwd_pointwise":  # ...,wt,Cin,dv,du,stream
        return f"bn_bwd_pw_kernel<{(a[18] + 15) // 16}>"
    if name == "orcai_dw_bwd_fused_conv0":
        return "dw_bwd_march_kernel<64, 2, true, true>"
    if name in ("orcai_sepconv_planes_stats", "orcai_sepconv_planes_stats_bn"):  # the LDS-tile kernels with the depthwise-output store and the statistics epilogue
        bn = name.endswith("_bn")
        Cin, W, Cout = a[2], a[4], a[14 if bn else 10]
        relu = "false" if bn else ("true" if a[5] else "false")
        mt, cq, nstrip = (Cout + 15) // 16, (Cin + 3) // 4, (W + 61) // 62
        if mt == 2 and cq <= 8 and nstrip >= 2 and W * 100 >= nstrip * 62 * 85:  # launch_sepconv_impl's rule for the strip tiles
            return f"sepconv_tile_kernel<2, {4 if cq <= 4 else 8}, false, {relu}, 8, true, 1, {'true' if bn else 'false'}>"
        return f"sepconv_ftile_kernel<{mt}, false, {relu}, true, 8, 1, {'true' if bn else 'false'}>"
    if name == "orcai_outer_reduce":  # A,Ca,Bq,Cb,...: 256 pixels per pass up to 32 channels per operand, 128 beyond
        return f"outer_reduce_kernel<{256 if max(a[1], a[3]) <= 32 else 128}>"
    if name in ("orcai_pool_bwd_bn_bias", "orcai_pool_bwd_bn", "orcai_pool_bwd"):
        return "pool_bwd_kernel"
    if name == "orcai_pool_res_add_bn":  # s,prev,B,C,...
        return f"pool_res_add_kernel<{(a[3] + 15) // 16}>"
    return LAUNCHER_KERNELS.get(name, name)


def measured_traffic_symbol(symbol: str, workload: str):
    """HBM bytes per launch of one kernel symbol from the newest PMC table (None when it is not there)."""
    import json

    f = traffic_file()
    if f is None:
        return None
    rec = json.loads(f.read_text()).get(workload, {}).get("kernels", {}).get(symbol)
    return None if rec is None else rec["hbm_bytes_per_launch"]


def measured_traffic_prefix(prefix: str, workload: str):
    """Mean HBM bytes per launch over every symbol of the table that starts with `prefix` (a launcher's templated kernel family)."""
    import json

    f = traffic_file()
    if f is None:
        return None
    ks = json.loads(f.read_text()).get(workload, {}).get("kernels", {})
    hit = [(v["hbm_bytes_per_launch"], v["launches"]) for k, v in ks.items() if k.startswith(prefix)]
    if not hit:
        return None
    return round(sum(b * n for b, n in hit) / sum(n for _, n in hit))


def traffic_has(workload: str) -> bool:
    import json

    f = traffic_file()
    return f is not None and workload in json.loads(f.read_text())


def _train_call_bytes(name, a):
    """Algorithmic HBM bytes of one launcher call (each fp32 tensor read / written once at its true channel count), from its
    arguments; None for launchers that are not plane-streaming kernels."""
    if name == "orcai_sepconv_planes_u":  # in,B,Cin,H,W,ksize,ktap,relu_in,dw,pw,scale,shift,Cout,relu_out,layout,H2,W2,out,u_out,stream
        B, Cin, H, W, Cout, u_out = a[1], a[2], a[3], a[4], a[12], a[18]
        return 4.0 * B * H * W * (Cin + Cout + (Cin if u_out else 0))
    if name == "orcai_sepconv_planes_stats":  # in,B,Cin,H,W,relu_in,dw,pw,scale,shift,Cout,out,u_out,shards,stream (statistics in the epilogue)
        B, Cin, H, W, Cout = a[1], a[2], a[3], a[4], a[10]
        return 4.0 * B * H * W * (2 * Cin + Cout)
    if name == "orcai_sepconv_planes_stats_bn":  # v_in,B,Cin,H,W,mean,var,gamma,beta,eps,dw,pw,scale,shift,Cout,out,u_out,shards,stream: v in, v out, depthwise output
        B, Cin, H, W, Cout = a[1], a[2], a[3], a[4], a[14]
        return 4.0 * B * H * W * (2 * Cin + Cout)
    if name == "orcai_sepconv_planes_epi":  # in,B,Cin,H,W,dw,pw,scale,shift,Cout,out,epi,ref,...: gradient in, gradient out, the reference tensor
        B, Cin, H, W, Cout = a[1], a[2], a[3], a[4], a[9]
        return 4.0 * B * H * W * (Cin + 2 * Cout)
    if name == "orcai_bn_bwd_pointwise_wgrad":  # dy,v,u,B,C,H,W,ksize,...,wt,Cin,du,...: dy, v (C channels), u and du (Cin channels); dv is never moved
        B, C, H, W, Cin = a[3], a[4], a[5], a[6], a[19]
        return 4.0 * B * H * W * (2 * C + 2 * Cin)
    if name == "orcai_dw_bwd_fused":  # x,du,B,C,H,W,...: du and x read once, dr written once
        return 4.0 * a[2] * a[4] * a[5] * 3 * a[3]
    if name == "orcai_dw_bwd_fused_conv0":  # in,stride,du,B,H,W,...: du read, dr written (16 channels), the snippet and the quarter-size residual gradient
        return 4.0 * a[3] * a[4] * a[5] * (2 * 16 + 1 + 4)
    if name == "orcai_dw_wgrad_bn":  # v,du,B,C,H,W,...
        return 4.0 * a[2] * a[4] * a[5] * 2 * a[3]
    if name == "orcai_bn_bwd_pointwise":  # dy,v,B,C,H,W,ksize,mean,var,gamma,beta,eps,relu,scratch,sums_ready,dbeta,dgamma,wt,Cin,dv,du,stream
        B, C, H, W, Cin = a[2], a[3], a[4], a[5], a[18]
        return 4.0 * B * H * W * (3 * C + Cin)
    if name == "orcai_outer_reduce":  # A,Ca,Bq,Cb,B,H,W,...
        return 4.0 * a[4] * a[5] * a[6] * (a[1] + a[3])
    if name == "orcai_dw_wgrad":  # x,du,B,C,H,W,...
        return 4.0 * a[2] * a[4] * a[5] * 2 * a[3]
    if name in ("orcai_bn_planes_stats",):  # v,B,C,H,W,...
        return 4.0 * a[1] * a[3] * a[4] * a[2]
    if name == "orcai_bn_planes_apply":
        return 4.0 * a[1] * a[3] * a[4] * 2 * a[2]
    return None


class TrainWorkload:
    """BASELINE configs[3]: orcai train, orcai-V1 architecture, synthetic snippets resident in HBM, batch 64 per GPU,
    data parallel over RCCL (one flat 3.98 MB gradient all-reduce per step).  One step = forward (training mode) +
    masked BCE + backward + all-reduce + Adam; metric snippets/s."""

    name = "orcai-V1 train step, synthetic snippets (64 per GPU, or the global batch of 64 split over the ranks with --dp-batch split)"
    metric = "snippets_per_s"
    unit = "snippets/s"
    dtype = "f32"

    def __init__(self, device, rank):
        from orcai_amd.architectures import ResNetLSTM
        from orcai_amd.training import Trainer

        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.B = _per_rank_batch(self.world)
        self.model = ResNetLSTM((736, 171, 1), 7, FILTERS, 3, 0.5, 128, seed=1)
        self.trainer = Trainer(self.model, 1e-4, seed=rank)
        self.timed = _TimedLib(self.trainer.trunk.lib)
        self.trainer.trunk.lib = self.timed
        self.trainer.head.lib = self.timed
        g = torch.Generator(device=device)
        g.manual_seed(4 + rank)
        self.x = torch.rand((self.B, 736, 171), device=device, generator=g).view(-1)
        self.y = (torch.rand((self.B, 46, 7), device=device, generator=g) > 0.7).float()
        self.units_per_step = float(self.B)
        self.ev = []

    def step(self, timed: bool):
        if timed:
            if self.timed.events is None:
                self.timed.events = {}
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        self.trainer.train_step(self.x, 736 * 171, self.B, self.y, world_size=self.world)
        if timed:
            e1.record()
            self.ev.append((e0, e1))

    def roofline(self):
        """The step against the f32-MFMA peak, and the dominant kernel SYMBOL of the step against HBM.  Which symbol dominates is measured, not
        assumed: two extra steps after the timed region bracket every orcai_* launcher with HIP events (the brackets slow the step, so they
        stay outside the timed region); the calls are grouped by the template instantiation they run (_train_call_symbol: what rocprofv3
        --stats ranks) and the symbol with the largest summed time among those whose algorithmic bytes are known (_train_call_bytes: every
        tensor read / written once at its true channel count) is reported -- per launch, achieved = bytes / time.  The calls bracketed INSIDE
        the timed steps (TimedLib.is_dominant) are last run's dominant symbol; `kernel_in_timed_steps` says whether the timed-region figure
        and the table agree on it."""
        ms = float(np.mean([a.elapsed_time(b) for a, b in self.ev]))
        n_steps = len(self.ev)
        timed_calls = dict(self.timed.events or {})
        flops = 3.0 * FWD_FLOP_PER_SNIPPET * self.B  # fwd + bwd ~ 3x forward
        out = {"step_ms": round(ms, 3), "step_tflops": round(flops / (ms * 1e-3) / 1e12, 2)}
        # where the time goes: two more steps with every launcher bracketed (rank-local: NO collective here -- only rank 0 calls roofline())
        self.timed.mode, self.timed.events = "all", {}
        for _ in range(2):
            self.trainer.train_step(self.x, 736 * 171, self.B, self.y, world_size=1)
        torch.cuda.synchronize()
        table, by_symbol = {}, {}
        overhead = bracket_overhead_ms()
        out["bracket_overhead_us_subtracted_per_call"] = round(overhead * 1e3, 1)
        for name, calls in self.timed.events.items():
            t = sum(_net(name, a, b, overhead) for a, b, _ in calls) / 2
            by = [_train_call_bytes(name, args) for _, _, args in calls]
            table[name] = (t, None if any(b is None for b in by) else sum(by) / 2, len(calls) // 2)
            for (e0, e1, args), b in zip(calls, by):  # the same calls by kernel SYMBOL: what rocprofv3 --stats ranks
                d = by_symbol.setdefault(_train_call_symbol(name, args), {"ms": 0.0, "bytes": 0.0, "n": 0, "launcher": name, "priced": True})
                d["ms"] += _net(name, e0, e1, overhead) / 2
                d["n"] += 0.5
                d["priced"] = d["priced"] and b is not None
                d["bytes"] += (b or 0.0) / 2
        self.timed.mode, self.timed.events = "dominant", None
        out["launcher_ms_per_step_instrumented"] = {k: round(v[0], 3) for k, v in sorted(table.items(), key=lambda kv: -kv[1][0])[:12]}
        out["symbol_ms_per_step_instrumented"] = {k: round(v["ms"], 3) for k, v in sorted(by_symbol.items(), key=lambda kv: -kv[1]["ms"])[:8]}
        priced = {k: v for k, v in by_symbol.items() if v["priced"] and v["bytes"] > 0}
        if priced:
            top = max(priced, key=lambda k: priced[k]["ms"])
            d = priced[top]
            t, by, n = d["ms"], d["bytes"], max(1, int(round(d["n"])))
            ach = by / (t * 1e-3) / 1e9
            out.update({"bound": "hbm", "kernel": top, "launcher": d["launcher"], "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": measured_traffic_symbol(top, "train") if traffic_has("train") else None,
                        "kernel_ms": round(t / n, 4), "launches_per_step": n, "algorithmic_bytes_per_launch": round(by / n),
                        "measured": "fully bracketed steps after the timed region"})
            inside = [(e0, e1, args) for e0, e1, args in timed_calls.get(d["launcher"], []) if _train_call_symbol(d["launcher"], args) == top]
            out["kernel_in_timed_steps"] = bool(inside)
            if inside:  # the same symbol bracketed inside the timed steps (only it: an event pair costs ~15 us of queue time)
                ti = sum(a.elapsed_time(b) for a, b, _ in inside)
                bi = sum(_train_call_bytes(d["launcher"], args) for _, _, args in inside)
                out.update({"achieved": round(bi / (ti * 1e-3) / 1e9, 1), "frac": round(bi / (ti * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "kernel_ms": round(ti / len(inside), 4),
                            "launches_per_step": len(inside) // max(1, n_steps), "algorithmic_bytes_per_launch": round(bi / len(inside)), "measured": "inside the timed steps"})
        return out

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


HPS_FILTER_SETS = {"set1": [10, 20, 30, 40], "set2": [20, 30, 40, 50], "set3": [30, 40, 50, 60]}  # reference defaults/default_hps_parameter.json:2-25


def _train_flops_per_snippet(filters, k=3, units=128, H=736, W=171):
    """Forward MACs of a ResNetLSTM width variant (the SURVEY 8a row B3 count, generalised), x 2 FLOP x 3 (fwd + bwd)."""
    mac = H * W * k * k * 16
    c, h, w = 16, H, W
    for f in filters:
        mac += h * w * (k * k * c + c * f + k * k * f + f * f)
        h2, w2 = -(-h // 2), -(-w // 2)
        mac += h2 * w2 * c * f
        c, h, w = f, h2, w2
    mac += h * w * (k * k * c + c * 36)
    feat = w * 36
    mac += h * 2 * (feat * 4 * units + units * 4 * units) + h * 2 * (2 * units * 4 * units + units * 4 * units) + h * (2 * units * 128 + 128 * 7)
    return 6.0 * mac


def _h_call_symbol(name, a):
    """Kernel symbol of an f16-path launcher call (csrc/half_fwd.hip, half_bwd.hip), as tools/summarize_pmc.py names it in the PMC tables."""
    mt = lambda c: (c + 15) // 16  # noqa: E731
    if name == "orcai_h_sepconv":  # in,B,Cin,H,W,ksize_planes,ktap,relu_in,dw,pwf,scale,shift,Cout,relu_out,out_layout,H2,W2,out,u_out,stream
        ktap, cout, layout, u_out = a[6], a[12], a[14], a[18]
        if ktap == 3 and layout in (0, 2) and not (layout == 2 and u_out):
            return f"sepconv_h_ftile_kernel<{mt(cout)}, {'true' if layout == 2 else 'false'}, {'true' if u_out else 'false'}, false, false>"
        return f"sepconv_h_kernel<{ktap}, {mt(cout)}>"
    if name == "orcai_h_sepconv_stats":
        return f"sepconv_h_ftile_kernel<{mt(a[10])}, false, true, true, false>"
    if name == "orcai_h_sepconv_stats_bn":
        return f"sepconv_h_ftile_kernel<{mt(a[14])}, false, true, true, true>"
    if name == "orcai_h_bn_bwd_pointwise":  # dy,v,B,C,H,W,ksize,mean,var,gamma,beta,eps,relu,scratch,sums_ready,dbeta,dgamma,wt,Cin,dv,du,stream
        return f"bn_bwd_pw_h_kernel<{mt(a[18])}>"
    if name == "orcai_h_bn_bwd_pointwise_wgrad":  # dy,v,u,B,C,H,W,ksize,...,wt,Cin,du,...: <conv-input tiles, conv-output tiles>
        return f"bn_bwd_pw_wgrad_h_kernel<{mt(a[19])}, {mt(a[4])}>"
    if name in ("orcai_h_dw_bwd_fused", "orcai_h_dw_bwd_fused_res"):  # x,du,B,C,H,W,relu_in,dw_rev,dr,dW,epi,bn_mean,...
        W = a[5]
        best, lanes = 64, ((W + 61) // 62) * 64
        if ((W + 29) // 30) * 32 < lanes:
            best, lanes = 32, ((W + 29) // 30) * 32
        if ((W + 13) // 14) * 16 < lanes:
            best = 16
        if name.endswith("_res"):
            return f"dw_bwd_march_h_kernel<{best}, 2, true, true>"
        return f"dw_bwd_march_h_kernel<{best}, {a[10]}, {'true' if a[11] is not None else 'false'}, false>"
    if name == "orcai_h_pool_res_add":
        return f"pool_res_add_h_kernel<{mt(a[3])}>"
    return {"orcai_h_outer_reduce": "outer_reduce_h_kernel", "orcai_h_pool_bwd_bn": "pool_bwd_h_kernel", "orcai_h_pool_bwd_bn_bias": "pool_bwd_h_kernel", "orcai_h_bn_planes_apply": "bn_planes_apply_h_kernel",
            "orcai_h_planes_sum": "planes_sums_h_kernel", "orcai_h_bn_planes_stats": "planes_sums_h_kernel", "orcai_h_conv0_affine": "conv0_h_kernel<3>",
            "orcai_h_conv0_bn_bwd": "conv0_bn_wgrad_h_kernel<3, 8>", "orcai_h_conv0_bn_bwd_ready": "conv0_bn_wgrad_h_kernel<3, 8>"}.get(name, name)


def _h_call_bytes(name, a):
    """Algorithmic HBM bytes of one f16-path launcher call (2 bytes per element, each tensor once at its true channel count); None where no plane streams."""
    if name == "orcai_h_sepconv":
        B, Cin, H, W, Cout, layout, u_out = a[1], a[2], a[3], a[4], a[12], a[14], a[18]
        if layout == 1:
            return 2.0 * B * H * W * Cin + 4.0 * B * H * W * Cout + (2.0 * B * H * W * Cin if u_out else 0)
        return 2.0 * B * H * W * (Cin + Cout + (Cin if u_out else 0))
    if name == "orcai_h_sepconv_stats":
        return 2.0 * a[1] * a[3] * a[4] * (2 * a[2] + a[10])
    if name == "orcai_h_sepconv_stats_bn":
        return 2.0 * a[1] * a[3] * a[4] * (2 * a[2] + a[14])
    if name == "orcai_h_bn_bwd_pointwise":
        B, C, H, W, Cin = a[2], a[3], a[4], a[5], a[18]
        return 2.0 * B * H * W * (3 * C + Cin)
    if name == "orcai_h_bn_bwd_pointwise_wgrad":  # dy, v (C channels) read, u read and du written (Cin channels); dv is never moved
        return 2.0 * a[3] * a[5] * a[6] * (2 * a[4] + 2 * a[19])
    if name in ("orcai_h_dw_bwd_fused", "orcai_h_dw_bwd_fused_res"):
        return 2.0 * a[2] * a[4] * a[5] * 3 * a[3]
    if name == "orcai_h_outer_reduce":  # A,Ca,Bq,Cb,B,H,W,...
        return 2.0 * a[4] * a[5] * a[6] * (a[1] + a[3])
    if name in ("orcai_h_pool_bwd_bn", "orcai_h_pool_bwd_bn_bias"):  # dout,ybn,B,C,H,W,...: dout at the pooled resolution, v read, dy written
        return 2.0 * a[2] * a[4] * a[5] * a[3] * 2.25
    if name == "orcai_h_pool_res_add":  # s,prev,B,C,Cp,H,W,...: v_b read, prev at the sampled pixels, the block output
        return 2.0 * a[2] * a[5] * a[6] * (a[3] + 0.25 * a[4] + 0.25 * a[3])
    if name == "orcai_h_bn_planes_apply":  # v,B,C,H,W,...
        return 2.0 * a[1] * a[3] * a[4] * 2 * a[2]
    if name in ("orcai_h_planes_sum", "orcai_h_bn_planes_stats"):
        return 2.0 * a[1] * a[3] * a[4] * a[2]
    return None


class HpsearchWorkload:
    """BASELINE configs[4]: the three CNN width variants of the reference's hyper-parameter file (lstm_units 128, kernel 3, dropout
    0.5), batch 64 per GPU, data parallel (one RCCL all-reduce of each variant's flat gradient bucket per step), on the f16 path:
    f16 octet-plane activations and gradients, v_mfma_f32_16x16x32_f16 contractions, f32 master weights + Adam.  One step = one
    training step of EACH variant (3 x 64 snippets per GPU); metric snippets/s over the sweep."""

    name = "hpsearch sweep: width variants set1/set2/set3 (default_hps_parameter.json), lstm 128, k 3, dropout 0.5, batch 64 per GPU, f16 MFMA path"
    metric = "snippets_per_s"
    unit = "snippets/s"
    dtype = "f16"

    def __init__(self, device, rank, precision="f16", variants=("set1", "set2", "set3")):
        from orcai_amd.architectures import ResNetLSTM
        from orcai_amd.training import Trainer

        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.B = _per_rank_batch(self.world)
        self.device, self.rank, self.precision = device, rank, precision
        g = torch.Generator(device=device)
        g.manual_seed(5 + rank)
        self.x = torch.rand((self.B, 736, 171), device=device, generator=g).view(-1)
        self.y = (torch.rand((self.B, 46, 7), device=device, generator=g) > 0.7).float()
        self.variants = [v for v in os.environ.get("ORCAI_HPS_VARIANTS", ",".join(variants)).split(",") if v]  # profiling: one variant at a time
        self.trainers, self.timed = {}, {}

        def dominant(name, args):  # the f16 separable convolutions of block 1 (k = 3 taps, two output tiles or the widest plane)
            # (the training forward launches it through orcai_h_sepconv_stats: BatchNorm statistics in the epilogue, a 3 us zero fill in the bracket)
            return (name == "orcai_h_sepconv" and args[6] == 3 and args[3] >= 736) or (name in ("orcai_h_sepconv_stats", "orcai_h_sepconv_stats_bn") and args[3] >= 736)

        for v in self.variants:
            model = ResNetLSTM((736, 171, 1), 7, HPS_FILTER_SETS[v], 3, 0.5, 128, seed=1, precision=precision)
            tr = Trainer(model, 1e-4, seed=rank)
            tl = _TimedLib(tr.trunk.lib, lambda name, args: False)  # nothing is bracketed inside the timed steps: roofline() ranks its own extra step
            tr.trunk.lib = tl
            tr.head.lib = tl
            self.trainers[v], self.timed[v] = tr, tl
        self.units_per_step = float(self.B * len(self.variants))
        self.ev = {v: [] for v in self.variants}

    def step(self, timed: bool):
        for v in self.variants:
            tr = self.trainers[v]
            if timed:
                if self.timed[v].events is None:
                    self.timed[v].events = {}
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            tr.train_step(self.x, 736 * 171, self.B, self.y, world_size=self.world)
            if timed:
                e1.record()
                self.ev[v].append((e0, e1))

    def per_variant(self):
        out = {}
        for v in self.variants:
            ms = float(np.mean([a.elapsed_time(b) for a, b in self.ev[v]])) if self.ev[v] else None
            out[v] = {"filters": HPS_FILTER_SETS[v], "ms_per_step": None if ms is None else round(ms, 3),
                      "snippets_per_s_per_gpu": None if ms is None else round(self.B / (ms * 1e-3), 1),
                      "step_tflops": None if ms is None else round(_train_flops_per_snippet(HPS_FILTER_SETS[v]) * self.B / (ms * 1e-3) / 1e12, 2)}
        return out

    def roofline(self):
        """The dominant kernel SYMBOL of the sweep against HBM, ranked from this run's own table like TrainWorkload.roofline: one more sweep step after the
        timed region with every orcai_* launcher bracketed by HIP events, the calls grouped by the template instantiation they run (_h_call_symbol: the
        names rocprofv3 --stats prints, _Float16 symbols demangled by tools/summarize_pmc.py's rule) and priced with their algorithmic bytes (_h_call_bytes:
        every f16 tensor read / written once at its true channel count).  `traffic`: HBM bytes per launch of that symbol from the newest PMC table
        (section hpsearch_f16_set3: the counters were collected on the widest variant)."""
        for v in self.variants:
            self.timed[v].mode, self.timed[v].events = "all", {}
        world, self.world = self.world, 1  # rank-local: only rank 0 calls roofline(), so NO collective may run here
        try:
            self.step(False)
        finally:
            self.world = world
        torch.cuda.synchronize()
        by_symbol = {}
        overhead = bracket_overhead_ms()
        for v in self.variants:
            for name, calls in self.timed[v].events.items():
                for e0, e1, args in calls:
                    sym, by = _h_call_symbol(name, args), _h_call_bytes(name, args)
                    d = by_symbol.setdefault(sym, {"ms": 0.0, "bytes": 0.0, "n": 0, "priced": True, "launcher": name})
                    d["ms"] += _net(name, e0, e1, overhead)
                    d["n"] += 1
                    d["priced"] = d["priced"] and by is not None
                    d["bytes"] += by or 0.0
            self.timed[v].mode, self.timed[v].events = "dominant", None
        out = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "measured": "one fully bracketed sweep step after the timed region",
               "symbol_ms_per_sweep_step_instrumented": {k: round(d["ms"], 3) for k, d in sorted(by_symbol.items(), key=lambda kv: -kv[1]["ms"])[:10]}}
        priced = {k: d for k, d in by_symbol.items() if d["priced"] and d["bytes"] > 0}
        if priced:
            top = max(priced, key=lambda k: priced[k]["ms"])
            d = priced[top]
            ach = d["bytes"] / (d["ms"] * 1e-3) / 1e9
            out.update({"kernel": top, "launcher": d["launcher"], "achieved": round(ach, 1), "frac": round(ach / HBM_PEAK_GBS, 4), "kernel_ms": round(d["ms"] / d["n"], 4),
                        "launches_per_sweep_step": d["n"], "algorithmic_bytes_per_launch": round(d["bytes"] / d["n"]),
                        "traffic": measured_traffic_symbol(top, "hpsearch_f16_set3"), "traffic_from": "PMC passes on the set3 variant alone (mean per launch of the symbol)",
                        "top_symbol_of_full_table": max(by_symbol, key=lambda k: by_symbol[k]["ms"])})
        out["variants"] = self.per_variant()
        return out

    def loss_curves(self, steps=200):
        """fp16-vs-fp32 loss deviation: every variant trained `steps` steps on the same 4 synthetic batches with
        the same initial weights and dropout masks in both precisions (outside the timed region)."""
        from orcai_amd.architectures import ResNetLSTM
        from orcai_amd.training import Trainer

        g = torch.Generator(device=self.device)
        g.manual_seed(99)
        xs = torch.rand((4, self.B, 736, 171), device=self.device, generator=g)
        band = xs.view(4, self.B, 46, 16, 171)[..., :56].mean(dim=(3, 4))  # a learnable target: mean energy of a band per output step
        ys = (band[..., None] > band.median()).float().repeat(1, 1, 1, 7)
        out = {}
        for v in self.variants:
            curves = {}
            for prec in ("f32", "f16"):
                model = ResNetLSTM((736, 171, 1), 7, HPS_FILTER_SETS[v], 3, 0.5, 128, seed=1, precision=prec)
                tr = Trainer(model, 1e-3, seed=3)
                acc = []
                for s in range(steps):
                    o = tr.train_step(xs[s % 4].reshape(-1), 736 * 171, self.B, ys[s % 4], world_size=1)
                    acc.append(o["acc"][:2].clone())
                a = torch.stack(acc).cpu().numpy()
                curves[prec] = a[:, 0] / a[:, 1]
                skipped = int(tr.skipped.item()) if prec == "f16" else 0
                del tr, model
                torch.cuda.empty_cache()
            d = np.abs(curves["f16"] - curves["f32"])
            out[v] = {"steps": steps, "loss_first": round(float(curves["f32"][0]), 4), "loss_last20_f32": round(float(curves["f32"][-20:].mean()), 4),
                      "loss_last20_f16": round(float(curves["f16"][-20:].mean()), 4), "max_abs_dev": round(float(d.max()), 4), "mean_abs_dev": round(float(d.mean()), 4),
                      "overflow_steps": skipped}
        return out

    def cpu_baseline(self):
        from oracle import model_ref as M
        from oracle import train_ref as T

        cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("ORCAI_BENCH_CPU_THREADS", "16")))
        torch.set_num_threads(cores)
        rng = np.random.default_rng(0)
        n, done, t0 = 4, 0, time.perf_counter()
        x = rng.random((n, 736, 171, 1), dtype=np.float32)
        y = (rng.random((n, 46, 7)) > 0.7).astype(np.float32)
        for v in self.variants:
            p = M.random_params(seed=1, filters=tuple(HPS_FILTER_SETS[v]))
            T.loss_and_grads(p, x, y, None, 0.0, dtype=torch.float32)
            done += n
        dt = time.perf_counter() - t0
        return {"value": round(done / dt, 2), "unit": self.unit, "cores": cores, "kind": "port",
                "sample": f"oracle.train_ref.loss_and_grads (torch-CPU autograd, fp32, {cores} threads): one batch of {n} snippets per width variant, {dt:.1f} s"}
