#!/usr/bin/env python3
"""Benchmark of the orcAI hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload frontend|predict]

One "step" = one pass of the hot path over one synthetic recording already resident in HBM.
For N > 1 (launched by torch.distributed.run, one rank per GPU) every rank processes its own
recording (independent objects: weak scaling, no data-path collective); the timed region is
bracketed by barrier + synchronize and the MAX over ranks is reported.  Rank 0 prints ONE JSON line.

The CPU baseline leg times the CPU oracle (``oracle/``: numpy/scipy/torch-CPU restatement of the
reference path -- "port", NOT Keras/TF/librosa, which cannot be installed here) on a bounded sample.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

SPEC_PARAM = {"sampling_rate": 48000, "nfft": 512, "n_overlap": 256, "freq_range": [0, 16000], "quantiles": [0.01, 0.999], "duration": 4}
SNIPPET_FRAMES = 736
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3


def synth_pcm_device(n_samples: int, seed: int, device) -> torch.Tensor:
    """White noise + chirps generated on the device (same recipe as orcai_amd.synthetic)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    x = 0.2 * torch.randn(n_samples, generator=g, device=device, dtype=torch.float32)
    sr = 48000
    t = torch.arange(int(0.5 * sr), device=device, dtype=torch.float32) / sr
    chirp = 0.5 * torch.sin(2 * np.pi * (1000.0 * t + 0.5 * 16000.0 * t * t))
    for start in range(5 * sr, n_samples - len(chirp), 17 * sr):
        x[start : start + len(chirp)] += chirp
    x = torch.clamp(x / 1.6, -1.0, 1.0)
    return (torch.round(x * 32767.0) / 32768.0).contiguous()  # PCM16-representable


class FrontendWorkload:
    """configs[1]: STFT + dB + exact percentile clip + normalise of 1024 snippets of audio."""

    name = "frontend_1024_snippets_48kHz"
    kernel_symbol = "stft_db_kernel<true, true, 171>"  # as rocprofv3 names the launch of orcai_stft_db with the level-1 histogram and the 171-bin crop
    metric = "audio_seconds_per_s"
    unit = "audio-s/s"
    dtype = "f32"

    def __init__(self, device, rank):
        from orcai_amd import _native as N
        from orcai_amd.frontend import TOP_DB, FrontEnd, nearest_rank_index

        self.N, self.TOP_DB = N, TOP_DB
        self.fe = FrontEnd(device)
        self.lib = self.fe.lib
        self.n_snippets = 1024
        self.n_samples = self.n_snippets * SNIPPET_FRAMES * 256
        self.T = 1 + self.n_samples // 256
        self.K = 171
        self.pcm = synth_pcm_device(self.n_samples, 2 + rank, device)
        self.out = torch.empty((self.T, self.K), dtype=torch.float32, device=device)
        total = self.T * self.K
        self.r_lo = nearest_rank_index(total, 0.01)
        self.r_hi = nearest_rank_index(total, 0.999)
        self.units_per_step = self.n_samples / 48000.0  # audio seconds
        # PCM read once (753 664 B/snippet) + [T,171] f32 written once (503 424 B/snippet)
        self.kernel_alg_bytes = 1257088.0 * self.n_snippets
        self.kernel_name = "stft_db_kernel"
        self.ev = []

    def step(self, timed: bool):
        N, lib = self.N, self.lib
        s = N.stream_ptr()
        ws = N.ptr(self.fe.workspace)
        n = self.T * self.K
        N.check(lib.orcai_frontend_reset(ws, s), "reset")
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        N.check(lib.orcai_stft_db(N.ptr(self.pcm), self.n_samples, 512, 256, self.T, self.K, N.ptr(self.out), ws, s), "stft_db")
        if timed:
            e1.record()
            self.ev.append((e0, e1))
        N.check(lib.orcai_quantile_select(N.ptr(self.out), n, self.r_lo, self.r_hi, ws, s), "select")
        N.check(lib.orcai_frontend_finalize(1, self.TOP_DB, ws, s), "finalize")
        N.check(lib.orcai_clip_normalize(N.ptr(self.out), n, ws, s), "normalize")

    def roofline(self):
        ms = [a.elapsed_time(b) for a, b in self.ev]
        avg_s = float(np.mean(ms)) * 1e-3
        achieved = self.kernel_alg_bytes / avg_s / 1e9
        return {"bound": "hbm", "kernel": self.kernel_name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": round(self.kernel_alg_bytes),
                "traffic": measured_traffic(self.kernel_symbol), "kernel_ms": round(avg_s * 1e3, 4)}

    def cpu_baseline(self):
        from oracle import frontend_ref as F

        seconds = 1800.0
        rng = np.random.default_rng(7)
        y = (np.round(np.clip(0.125 * rng.standard_normal(int(seconds * 48000)), -1, 1) * 32767) / 32768).astype(np.float32)
        F.make_spectrogram_ref(y[: 48000 * 5], {"spectrogram": SPEC_PARAM})  # warm-up
        reps, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 12.0:
            F.make_spectrogram_ref(y, {"spectrogram": SPEC_PARAM})
            reps += 1
        dt = time.perf_counter() - t0
        return {"value": round(reps * seconds / dt, 1), "unit": self.unit, "cores": 1, "kind": "port",
                "sample": f"oracle.frontend_ref.make_spectrogram_ref (numpy/scipy, one thread) on {reps} x {seconds:.0f} s of 48 kHz noise, {dt:.1f} s wall"}


def measured_traffic(symbol: str):
    from bench_predict import measured_traffic as m

    return m(symbol, "frontend")


WORKLOADS = {"frontend": FrontendWorkload}
try:
    from bench_predict import HpsearchWorkload, PredictWorkload, TrainWorkload

    WORKLOADS["predict"] = PredictWorkload
    WORKLOADS["train"] = TrainWorkload
    WORKLOADS["hpsearch"] = HpsearchWorkload
except ImportError:
    pass


def _dp_label(dist, world, what="one flat fp32 gradient bucket") -> str:
    return f"dp{world} ({_backend_name(dist)} all-reduce of {what})" if dist else "dp1 (single rank: no collective)"


def _backend_name(dist) -> str:
    """What carries the collectives: RCCL (torch's "nccl" backend on ROCm) or, in the one-device rehearsal, gloo; "no" for a single rank."""
    if not dist:
        return "no"
    return "RCCL" if dist.get_backend() == "nccl" else dist.get_backend()


def _reduce(dist, t, op):
    """all_reduce of a small device tensor: in place over RCCL; staged through the host for the gloo rehearsal backend."""
    if dist.get_backend() == "nccl":
        dist.all_reduce(t, op=op)
        return t
    h = t.cpu()
    dist.all_reduce(h, op=op)
    t.copy_(h)
    return t


def allreduce_probe(dist, wl) -> dict:
    """The one collective of a training step, alone: all-reduce of the flat gradient bucket (3.98 MB for orcai-V1) over RCCL / xGMI.  Collective: every rank calls it."""
    tr = wl.trainer if hasattr(wl, "trainer") else next(iter(wl.trainers.values()))
    g = tr.P.g
    for _ in range(3):
        _reduce(dist, g, dist.ReduceOp.SUM)
    torch.cuda.synchronize()
    dist.barrier()
    t1 = time.perf_counter()
    for _ in range(20):
        _reduce(dist, g, dist.ReduceOp.SUM)
    torch.cuda.synchronize()
    return {"allreduce_us": round((time.perf_counter() - t1) / 20 * 1e6, 1), "allreduce_bytes": int(g.numel() * 4)}


def measure_secondary(device, rank, world, dist, steps=20, warmup=3):
    """BASELINE.json's metric has a second half -- training snippets/s at 1/2/4/8 GPUs -- that a single JSON line cannot carry as
    `value`.  After the headline measurement every rank also times the training step (configs[3]: batch 64 per GPU, data parallel,
    one RCCL all-reduce of the flat gradient bucket per step) the same way (warm-up, barrier + synchronize on both sides, MAX over
    ranks) and rank 0 attaches it as the `secondary` object.  A rank-local probe step runs first and its success is agreed on by all
    ranks, so a failure degrades to an `error` field instead of a hang."""
    ok, err, tw = 1, None, None
    try:
        tw = WORKLOADS["train"](device, rank)
        tw.trainer.train_step(tw.x, 736 * 171, tw.B, tw.y, world_size=1)  # probe: no collective
        torch.cuda.synchronize()
    except Exception as e:  # noqa: BLE001 - reported, never raised: the headline line must still be printed
        ok, err = 0, repr(e)
    if dist:
        flag = torch.tensor([ok], dtype=torch.int32, device=device)
        _reduce(dist, flag, dist.ReduceOp.MIN)
        ok = int(flag.item())
    if not ok:
        return {"metric": "snippets_per_s", "error": err or "the probe step failed on another rank"}
    for _ in range(warmup):
        tw.step(False)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tw.step(True)  # brackets only the dominant kernel's launches (HIP events on the launch stream)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        _reduce(dist, t, dist.ReduceOp.MAX)
        elapsed = float(t.item())
    out = {"metric": tw.metric, "value": round(tw.units_per_step * steps * world / elapsed, 1), "unit": tw.unit, "n_gpus": world, "steps": steps,
           "warmup": warmup, "ms_per_step": round(elapsed / steps * 1e3, 4), "scaling": dp_scaling(), "dtype": tw.dtype, "data": "synthetic",
           "config": {"workload": tw.name, "units_per_step_per_gpu": tw.units_per_step, "dp_batch": dp_batch_mode(), "per_rank_batch": tw.B, "global_batch": tw.B * world,
                      "parallelism": _dp_label(dist, world)}}
    if dist:
        out.update(allreduce_probe(dist, tw))
    if rank == 0:  # both halves of BASELINE's metric carry a roofline: the step against the f32 MFMA peak, its dominant kernel against HBM
        r = tw.roofline()
        r["step_frac_of_f32_mfma_peak"] = round(r["step_tflops"] / MFMA_F32_PEAK_TFLOPS, 4)
        out["roofline"] = r
    return out


def measure_sweep(device, rank, world, dist, steps=5, warmup=2):
    """BASELINE configs[4] attached to the default line as `secondary2`: the three width variants of the hyper-parameter sweep trained
    data parallel on the f16 path (HpsearchWorkload), timed like every other number here (warm-up, barrier + synchronize on both
    sides, MAX over ranks).  The 200-step f16-vs-f32 loss comparison belongs to `--workload hpsearch` (it takes a minute)."""
    ok, err, hw = 1, None, None
    try:
        hw = WORKLOADS["hpsearch"](device, rank)
        hw.world = 1
        hw.step(False)  # probe without collectives
        hw.world = world
        torch.cuda.synchronize()
    except Exception as e:  # noqa: BLE001
        ok, err = 0, repr(e)
    if dist:
        flag = torch.tensor([ok], dtype=torch.int32, device=device)
        _reduce(dist, flag, dist.ReduceOp.MIN)
        ok = int(flag.item())
    if not ok:
        return {"metric": "snippets_per_s", "error": err or "the probe step failed on another rank"}
    for _ in range(warmup):
        hw.step(False)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        hw.step(True)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        _reduce(dist, t, dist.ReduceOp.MAX)
        elapsed = float(t.item())
    out = {"metric": hw.metric, "value": round(hw.units_per_step * steps * world / elapsed, 1), "unit": hw.unit, "n_gpus": world, "steps": steps, "warmup": warmup,
           "ms_per_step": round(elapsed / steps * 1e3, 4), "scaling": dp_scaling(), "dtype": hw.dtype, "data": "synthetic",
           "config": {"workload": hw.name, "units_per_step_per_gpu": hw.units_per_step, "dp_batch": dp_batch_mode(), "per_rank_batch": hw.B, "global_batch": hw.B * world, "parallelism": _dp_label(dist, world, "one flat fp32 gradient bucket per variant")}}
    if rank == 0:
        out["roofline"] = hw.roofline()
    return out


def dp_batch_mode() -> str:
    """Batch semantics of the data-parallel training workloads (orcai_amd/datasets.py dp_batch, reference hpsearch.py:170-205): "replicate" keeps 64 snippets per
    rank (weak scaling: the throughput mode), "split" cuts the reference's GLOBAL batch of 64 into world slices -- MirroredStrategy's contract, strong scaling."""
    return os.environ.get("ORCAI_BENCH_DP_BATCH", "replicate")


def dp_scaling() -> str:
    return "strong" if dp_batch_mode() == "split" else "weak"


def launch_ranks(n: int, argv: list[str]) -> int:
    """`--gpus N` (N > 1) outside a launcher: start N fresh rank processes -- one per GPU, `python -m torch.distributed.run`, rendezvous on
    127.0.0.1 -- BEFORE anything in this process touches the GPU, pass the children's output through (rank 0 prints the JSON line) and
    return their exit code.  The driver's own `torch.distributed.run ... bench.py --gpus N` sets WORLD_SIZE and never comes here."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL needs it on this pool
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           str(Path(__file__).resolve())] + list(argv)
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks = GPUs of this node (default: WORLD_SIZE of the launcher, else 1)")
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 20; 200 for the 0.7 ms front-end step)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed warm-up steps (default 3; 50 for the front end: its step is too short to bring the clocks up)")
    ap.add_argument("--workload", default="predict" if "predict" in WORKLOADS else "frontend", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the training-throughput measurement attached to the predict line")
    ap.add_argument("--no-loss-curves", action="store_true", help="hpsearch workload: skip the 200-step f16-vs-f32 loss comparison")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for N > 1; gloo + --one-device rehearses the multi-rank control flow on a single GPU")
    ap.add_argument("--one-device", action="store_true", help="rehearsal only: every rank uses cuda:0 (never a measurement)")
    ap.add_argument("--curve-steps", type=int, default=200)
    ap.add_argument("--dp-batch", default=None, choices=["replicate", "split"],
                    help="training workloads at N > 1: replicate = 64 snippets per rank (weak scaling, default); split = the reference's global batch of 64 cut into N slices "
                         "(MirroredStrategy, hpsearch.py:170-205; strong scaling)")
    args = ap.parse_args()
    if args.dp_batch is not None:
        os.environ["ORCAI_BENCH_DP_BATCH"] = args.dp_batch  # read by the training workloads (and inherited by self-launched ranks)

    if args.gpus is None:
        args.gpus = int(os.environ.get("WORLD_SIZE", "1"))
    if args.steps is None:
        args.steps = 200 if args.workload == "frontend" else 20
    if args.warmup is None:
        args.warmup = 50 if args.workload == "frontend" else 3
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without an outer launcher: this process touches no GPU, starts N ranks and relays rank 0's line
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (no CPU fallback exists for the product path)")
    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)  # RCCL over xGMI
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    wl = WORKLOADS[args.workload](device, rank)
    if hasattr(wl, "events"):
        wl.events = {}
    drain = getattr(wl, "drain", lambda: None)  # predict: the host half of the last recording (see PredictWorkload.step)
    for _ in range(args.warmup):
        wl.step(False)
    drain()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wl.step(True)
    drain()  # inside the timed region: every recording's label table is complete when the clock stops
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        _reduce(dist, t, dist.ReduceOp.MAX)
        elapsed = float(t.item())

    secondary = None
    secondary2 = None
    if args.workload == "predict" and "train" in WORKLOADS and not args.no_secondary:
        torch.cuda.empty_cache()  # the headline workload's cached blocks back to the driver: the training buffers get fresh, contiguous ranges as in a stand-alone run
        secondary = measure_secondary(device, rank, world, dist)
        torch.cuda.empty_cache()
        secondary2 = measure_sweep(device, rank, world, dist)

    ar_probe = allreduce_probe(dist, wl) if dist and args.workload in ("train", "hpsearch") else {}  # a collective: every rank
    if rank == 0:
        value = wl.units_per_step * args.steps * world / elapsed
        line = {
            "metric": wl.metric, "value": round(value, 1), "unit": wl.unit, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": wl.dtype, "data": "synthetic",
            "config": {"workload": wl.name, "units_per_step_per_gpu": round(wl.units_per_step, 3), "parallelism": f"independent recordings x{world}"},
            "roofline": wl.roofline(),
        }
        if dist:  # the world size the process group itself reports (RCCL when backend is nccl)
            line["rccl_ranks" if args.backend == "nccl" else "gloo_ranks"] = dist.get_world_size()
        if secondary is not None:
            line["secondary"] = secondary
        if secondary2 is not None:
            line["secondary2"] = secondary2
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = wl.cpu_baseline()
        if args.workload in ("train", "hpsearch"):
            line["scaling"] = dp_scaling()
            line["config"].update({"dp_batch": dp_batch_mode(), "per_rank_batch": wl.B, "global_batch": wl.B * world})
            line["config"]["parallelism"] = _dp_label(dist, world)
            line.update(ar_probe)
        if args.workload == "hpsearch" and world == 1 and not args.no_loss_curves:
            line["loss_curves_f16_vs_f32"] = wl.loss_curves(args.curve_steps)
        print(json.dumps(line), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
