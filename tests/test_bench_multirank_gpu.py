"""bench.py with N > 1 ranks, rehearsed on one GPU over gloo (`--backend gloo --one-device`): the control flow the driver's multi-GPU run takes -- self-launched
ranks, barriers, MAX-over-ranks timing, rank 0's roofline tables (which must not contain a collective: only rank 0 computes them) -- has to finish and print one
JSON line.  RCCL itself needs two GPUs and is not exercised here."""

import json
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _bench(*args, timeout=600):
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--backend", "gloo", "--one-device", "--no-cpu-baseline", *args], capture_output=True, text=True,
                       timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # rank 0 prints ONE line
    return json.loads(lines[0])


def test_default_line_with_two_ranks_and_split_batches():
    d = _bench("--steps", "1", "--warmup", "1", "--dp-batch", "split")
    assert d["n_gpus"] == 2 and d["gloo_ranks"] == 2 and d["scaling"] == "weak" and d["metric"] == "audio_seconds_per_s"  # predict: independent recordings
    s, s2 = d["secondary"], d["secondary2"]
    assert "error" not in s and s["n_gpus"] == 2 and s["scaling"] == "strong" and s["config"]["per_rank_batch"] == 32 and s["config"]["global_batch"] == 64
    assert s["allreduce_us"] > 0 and s["allreduce_bytes"] == 4 * 994959
    assert "error" not in s2 and s2["scaling"] == "strong" and s2["roofline"]["kernel"]  # rank 0's own fully bracketed sweep step ran without a collective
    assert d["roofline"]["kernel"].split("<")[0] in ("sepconv_pool_march_kernel", "conv0_sep_tile_kernel")  # (two ranks time-share the one card here: either of the two largest may top)


def test_training_workloads_with_two_ranks():
    d = _bench("--workload", "hpsearch", "--steps", "1", "--warmup", "1", "--no-loss-curves")
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["per_rank_batch"] == 64 and d["allreduce_us"] > 0 and d["roofline"]["variants"]["set3"]["ms_per_step"] > 0
    d = _bench("--workload", "train", "--steps", "2", "--warmup", "1", "--dp-batch", "split")
    assert d["scaling"] == "strong" and d["config"]["per_rank_batch"] == 32 and d["value"] > 0 and d["roofline"]["kernel"]
