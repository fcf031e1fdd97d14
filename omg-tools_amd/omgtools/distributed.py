"""Sharding of independent agents across ranks (one process per GPU).

The reference has no multi-process execution; independent
point-to-point problems shard by agents with no collective on the solve path.
The only cross-rank traffic is reporting: sum of solved agents and max of the
elapsed time.  Works with any torch.distributed backend (`nccl` = RCCL on the
GPUs, `gloo` in the CPU tests).
"""
import numpy as np


def shard_range(n_total, rank, world):
    """Contiguous block of agents owned by `rank` (sizes differ by at most 1)."""
    base, rem = divmod(int(n_total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_workload(P, rank, world):
    """Strong scaling of one batch: the per-agent arrays of a workload (`p`, `x0`, ... -- every array whose first axis is the
    agent axis) cut to the contiguous block of `rank`; everything else shared.  The agents are independent problems, so a
    rank's block solved alone gives the bits it gives inside the whole batch (tests/test_distributed_cpu.py)."""
    n_total = P['p'].shape[0]
    lo, hi = shard_range(n_total, rank, world)
    out = dict(P)
    for k, v in P.items():
        if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == n_total:
            out[k] = v[lo:hi]
    return out, (lo, hi)


def reduce_report(elapsed_s, n_solved, device=None, dist=None):
    """(max elapsed over ranks, total solved over ranks)."""
    if dist is None or not dist.is_initialized():
        return float(elapsed_s), int(n_solved)
    import torch
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    c = torch.tensor([float(n_solved)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return float(t.item()), int(c.item())


def gather_solutions(x_local, n_total, dist=None):
    """All ranks' coefficient blocks in agent order (used by examples/tests)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return np.asarray(x_local)
    import torch
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, np.asarray(x_local))
    out = np.concatenate(parts, axis=0)
    assert out.shape[0] == n_total
    return out
