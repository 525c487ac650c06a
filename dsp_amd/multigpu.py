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


def iir_time_sharded(x_local, a, b, c0: float = 1.0, d0: float = 0.0, *, rank: int, world: int, gather_states=None,
                     state=None, mode: str = "auto", iir=None, settle=None):
    """ONE long channel (or a few) spread over GPUs in TIME: rank g holds samples [g*n/G, (g+1)*n/G) of every channel
    (BASELINE config C4b across GPUs).  The recursion of Coefficients<P>::operator()
    (coefficients.h:88-130) is sequential in time, but a stable filter forgets: after `settle` samples
    (dsp_gpu_iir_settle: ||A^k|| <= 2^-44) the state no longer depends on where it started.  So
      1. every rank filters its segment from a zero state and keeps the end state Z_g = {last P inputs, last P outputs};
      2. the end states are all-gathered (2P doubles per channel and rank: the only exchange);
      3. rank g > 0 re-runs its first 2 * `settle` outputs from Z_{g-1}, which is the true state there to double rounding.
    Returns (y_local, final_state): final_state is the state after the LAST rank's last sample on every rank (what a
    following block would start from).  `state` (rank 0 only matters): the state before the first sample.
    x_local: [channels, n_local] or [n_local] fp32 on this rank's GPU, n_local >= settle.
    gather_states(t) -> list of `world` tensors; default torch.distributed.all_gather.  `iir` / `settle`: the local
    filter and its memory, default dsp_amd.ops.iir / ops.iir_settle (tests pass CPU stand-ins)."""
    import torch
    if iir is None or settle is None:
        from . import ops
        iir = iir or ops.iir
        settle = settle if settle is not None else ops.iir_settle(a, b)
    poles = len(a) - 1
    x2 = x_local if x_local.ndim > 1 else x_local[None]
    channels, n_local = x2.shape
    if world > 1 and n_local < settle:
        raise ValueError(f"segments of {n_local} samples are shorter than the filter's memory ({settle} samples)")
    if gather_states is None:
        import torch.distributed as dist

        def gather_states(t):
            parts = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(parts, t)
            return parts
    st = torch.zeros(channels, 2 * poles, dtype=torch.float64, device=x2.device)
    if rank == 0 and state is not None:
        st.copy_(state.reshape(channels, 2 * poles))
    y = iir(x2.contiguous(), a, b, c0, d0, state=st, mode=mode)
    states = gather_states(st)
    if rank > 0:
        head_state = states[rank - 1].clone()
        w = min(2 * int(settle), n_local)  # twice the settle length: 2^-44 of the STATE can be 1e-5 of the OUTPUT of an ill-conditioned
        # high-order direct form (iir_batch_ext.hip, where the same sharding runs inside one GPU)
        y[:, :w] = iir(x2[:, :w].contiguous(), a, b, c0, d0, state=head_state, mode=mode)
    final = states[world - 1]
    return (y if x_local.ndim > 1 else y[0]), final
