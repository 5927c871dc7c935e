"""Stream-parallel sharding: independent camera streams are dealt to ranks (one
process per GPU); the per-frame path has no communication.  The only exchanges are
the barrier/MAX around a timed region and a final gather of per-stream results."""
from typing import List, Sequence

import torch
import torch.distributed as dist


def shard_streams(n_streams: int, rank: int, world: int) -> List[int]:
    """Global stream ids owned by `rank`: stream s runs on rank s mod world."""
    return list(range(rank, n_streams, world))


def stream_seed(base_seed: int, stream_id: int) -> int:
    """Seed of a stream depends on its global id only, never on the rank layout."""
    return base_seed + stream_id


def gather_per_stream(local: torch.Tensor, owned: Sequence[int], n_streams: int, world: int) -> torch.Tensor:
    """all_gather per-stream rows (first dim = owned streams, padded to equal length) and
    reorder them by global stream id.  Works on any backend (gloo on CPU, RCCL on GPU)."""
    if world == 1:
        return local.clone()
    per = (n_streams + world - 1) // world
    pad = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:len(owned)] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    out = torch.zeros((n_streams,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    for r in range(world):
        ids = shard_streams(n_streams, r, world)
        out[ids] = parts[r][:len(ids)]
    return out


def max_over_ranks(value: float, device, world: int) -> float:
    if world == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_scalars(value: float, device, world: int) -> List[float]:
    """Every rank's value, in rank order, on every rank (an imbalanced node shows in the list)."""
    if world == 1:
        return [float(value)]
    t = torch.tensor([value], dtype=torch.float64, device=device)
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t)
    return [float(p.item()) for p in parts]
