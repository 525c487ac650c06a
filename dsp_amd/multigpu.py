"""Multi-GPU plumbing shared by bench.py and the tests: one process per GPU, channels sharded, no
data-path collective.  torch.distributed is used for the rendezvous, the barrier
and the max-over-ranks of the timing only."""
from __future__ import annotations


def shard_range(total: int, rank: int, world: int):
    """Contiguous channel range [lo, hi) of `rank`: GPU g gets channels [g*C/G, (g+1)*C/G)."""
    if not (0 <= rank < world):
        raise ValueError("rank outside the world")
    return (total * rank) // world, (total * (rank + 1)) // world


def max_over_ranks(value: float, device=None, group=None) -> float:
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def sum_over_ranks(value: float, device=None, group=None) -> float:
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return float(t.item())
