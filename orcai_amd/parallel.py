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


def contiguous_range(n_items: int, rank: int, size: int) -> tuple[int, int]:
    """[start, stop) of the contiguous block of n_items this rank owns (sizes differ by at most one, earlier ranks larger).
    Used to shard the snippets of ONE recording: rank r needs spectrogram rows [shift*start, shift*(stop-1) + length)."""
    base, extra = divmod(max(n_items, 0), size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def all_gather_rows(local: torch.Tensor, n_total: int) -> torch.Tensor:
    """Concatenate the ranks' row blocks (made with contiguous_range) along dim 0: the one exchange step of a predict
    sharded by snippet ranges ([n_g, 46, 7] f32, 1.3 KB per snippet).  Blocks are padded to the largest block
    because all_gather needs equal shapes.  RCCL when the group is nccl; staged through the host for gloo."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    size = dist.get_world_size()
    counts = [contiguous_range(n_total, r, size) for r in range(size)]
    biggest = max(b - a for a, b in counts)
    nccl = dist.get_backend() == "nccl"
    buf = local if nccl else local.cpu()
    padded = torch.zeros((biggest,) + tuple(buf.shape[1:]), dtype=buf.dtype, device=buf.device)
    padded[: buf.shape[0]] = buf
    parts = [torch.empty_like(padded) for _ in range(size)]
    dist.all_gather(parts, padded)
    out = torch.cat([parts[r][: b - a] for r, (a, b) in enumerate(counts)], dim=0)
    return out if nccl else out.to(local.device)


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
