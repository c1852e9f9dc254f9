"""Sharding helpers for the multi-GPU path (one process per GPU, torch.distributed over RCCL).

The path shards by codeword block: decoder, convs and interleavers are per-block
independent.  The single cross-block coupling is power_constraint's batch-global mean / unbiased std
(encoders.py:107-108), reproduced exactly by all-reducing three doubles (sum, sum of squares,
count) between the encoder and the normalisation; error counts need one more tiny all-reduce per
SNR point.  Both are latency-bound 16-24 byte collectives - no ring/bucket design is needed.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch


def shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) share of n items for `rank`; sizes differ by at most one."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_reduce_sum_(t: torch.Tensor, group=None) -> torch.Tensor:
    """In-place SUM all-reduce if a process group is initialised, else a no-op."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def stats_from_tensor(x_tx: torch.Tensor) -> torch.Tensor:
    """(sum, sum of squares, count) in fp64 - the host twin of what tae_encode_prenorm returns."""
    x = x_tx.double()
    return torch.stack([x.sum(), (x * x).sum(), torch.tensor(float(x.numel()), dtype=torch.float64, device=x.device)])


def mean_std_from_stats(stats: torch.Tensor) -> Tuple[float, float]:
    """mean and UNBIASED std exactly as normalize_kernel derives them (fp64, then rounded to fp32)."""
    s, ss, n = (float(v) for v in stats.detach().cpu().tolist())
    mean = s / n
    var = max((ss - s * mean) / (n - 1.0), 0.0)
    return float(torch.tensor(mean, dtype=torch.float32)), float(torch.tensor(var ** 0.5, dtype=torch.float32))
