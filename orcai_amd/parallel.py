"""One process per GPU.  The predict path shards by independent recordings: no data-path collective;
ranks only agree on the shard assignment and exchange small result summaries.  Training (when built) all-reduces
one flat fp32 gradient bucket over RCCL (backend "nccl" on ROCm)."""

from __future__ import annotations

import os

import torch
import torch.distributed as dist


def world() -> tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torchrun environment (1-process defaults)."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend: str | None = None) -> tuple[int, int, int]:
    """Initialise the default process group when launched with WORLD_SIZE > 1 (RCCL on GPUs, gloo on CPU)."""
    rank, size, local = world()
    if size > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=size, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=size)
    return rank, size, local


def shard_indices(n_items: int, rank: int, size: int, costs=None) -> list[int]:
    """Indices of the items this rank processes.  Without costs: round-robin.  With costs (e.g. recording
    durations): longest-first greedy assignment to the least-loaded rank (deterministic, identical on all ranks)."""
    if costs is None:
        return list(range(rank, n_items, size))
    order = sorted(range(n_items), key=lambda i: (-float(costs[i]), i))
    load = [0.0] * size
    mine = []
    for i in order:
        r = min(range(size), key=lambda k: (load[k], k))
        load[r] += float(costs[i])
        if r == rank:
            mine.append(i)
    return sorted(mine)


def gather_objects(obj):
    """All ranks' objects, in rank order (small summaries only: never the data path)."""
    if not dist.is_initialized():
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


def max_over_ranks(value: float) -> float:
    if not dist.is_initialized():
        return float(value)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
