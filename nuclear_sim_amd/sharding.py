"""Multi-GPU host logic: plants are independent, so the path shards as contiguous blocks of
plants per rank with NO collective while stepping.  The only exchange is the end-of-episode
concat of the observation block (all-gather, RCCL over xGMI on GPUs / gloo in CPU tests) and a
sum of a few int64 event counters."""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist


def shard_range(n_global: int, rank: int, world: int) -> Tuple[int, int]:
    """[lo, hi) of the contiguous shard of rank `rank`; shards differ by at most one plant and are
    indexed by GLOBAL plant id so per-plant seeds / ICs do not depend on the world size."""
    base, rem = divmod(n_global, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_observations(obs_local: torch.Tensor, n_global: int, group=None) -> torch.Tensor:
    """All-gather the [n_local, 22] observation blocks into [n_global, 22] on every rank."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = [shard_range(n_global, r, world) for r in range(world)]
    maxn = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((maxn, obs_local.shape[1]), dtype=obs_local.dtype, device=obs_local.device)
    pad[: obs_local.shape[0]] = obs_local
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad, group=group)
    return torch.cat([o[: hi - lo] for o, (lo, hi) in zip(out, sizes)], dim=0)


def reduce_counters(counters: torch.Tensor, group=None) -> torch.Tensor:
    """Sum int64 event counters (scram count, pump trips, turbine trips, maintenance events)."""
    c = counters.clone()
    dist.all_reduce(c, op=dist.ReduceOp.SUM, group=group)
    return c


def event_histogram(events: torch.Tensor, bins: int = 16, group=None) -> torch.Tensor:
    """Histogram of a per-plant event count over the WHOLE job (BASELINE config 4's report: how many plants topped off 0, 1, 2 ...
    times): the local bincount, counts beyond the last bin clamped into it, summed over the ranks when there are several."""
    hist = torch.bincount(events.to(torch.int64).clamp(min=0, max=bins - 1), minlength=bins)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(hist, op=dist.ReduceOp.SUM, group=group)
    return hist
