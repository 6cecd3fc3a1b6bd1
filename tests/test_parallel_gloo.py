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
    # one recording sharded by contiguous snippet ranges: rank blocks of a fake prediction tensor gather back to the whole
    import torch

    n = 29  # snippets of a 60 s recording
    whole = torch.arange(n * 46 * 7, dtype=torch.float32).reshape(n, 46, 7)
    a, b = parallel.contiguous_range(n, r, s)
    gathered = parallel.all_gather_rows(whole[a:b].clone(), n)
    q.put((r, everyone, slowest, (a, b), bool(torch.equal(gathered, whole))))
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
    assert sorted(res[3] for res in results) == [(0, 15), (15, 29)]
    assert all(res[4] for res in results)
    for rank, everyone, slowest, _, _ in results:
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


def test_contiguous_range_is_a_partition():
    from orcai_amd.parallel import contiguous_range

    for n in (0, 1, 7, 29, 1833):
        for size in (1, 2, 3, 8):
            blocks = [contiguous_range(n, r, size) for r in range(size)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[k][1] == blocks[k + 1][0] for k in range(size - 1))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
