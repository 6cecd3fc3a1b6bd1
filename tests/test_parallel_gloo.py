"""N > 1 path on CPU: world_size-2 gloo processes agree on the recording shards (the predict path has no
data-path collective) and on the max-over-ranks timing bench.py reports."""

import os
import socket
import sys
from pathlib import Path

import pytest
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, size, port, q):
    sys.path.insert(0, str(ROOT))
    os.environ.update({"RANK": str(rank), "WORLD_SIZE": str(size), "LOCAL_RANK": str(rank), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    from orcai_amd import parallel

    r, s, _ = parallel.init(backend="gloo")
    durations = [60, 3600, 10, 10, 1800, 600, 5]
    mine_rr = parallel.shard_indices(len(durations), r, s)
    mine_lb = parallel.shard_indices(len(durations), r, s, costs=durations)
    everyone = parallel.gather_objects({"rank": r, "rr": mine_rr, "lb": mine_lb})
    slowest = parallel.max_over_ranks(1.0 + r)
    q.put((r, everyone, slowest))
    import torch.distributed as dist

    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, everyone, slowest in results:
        assert [e["rank"] for e in everyone] == [0, 1]
        rr = sorted(everyone[0]["rr"] + everyone[1]["rr"])
        lb = sorted(everyone[0]["lb"] + everyone[1]["lb"])
        assert rr == list(range(7)) and lb == list(range(7))  # a partition: every recording exactly once
        assert not set(everyone[0]["lb"]) & set(everyone[1]["lb"])
        assert slowest == 2.0
    durations = [60, 3600, 10, 10, 1800, 600, 5]
    loads = [sum(durations[i] for i in results[0][1][k]["lb"]) for k in range(2)]
    assert max(loads) == 3600  # longest-first greedy: the 1 h recording alone on one rank


def test_single_process_defaults():
    from orcai_amd import parallel

    assert parallel.shard_indices(5, 0, 1) == [0, 1, 2, 3, 4]
    assert parallel.gather_objects("x") == ["x"] and parallel.max_over_ranks(3.5) == 3.5
