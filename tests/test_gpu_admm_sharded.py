"""The sharded formation iteration on device tensors (option 1): two ranks of one fleet run in one process on
one GPU -- a thread each, with its own `BatchSolver` handle, `HipAdmmOps` and `BatchADMM(rank, world=2)` -- and exchange
through a stand-in for `torch.distributed` whose `all_gather_into_tensor` copies between the two ranks' device buffers
behind a thread barrier.  Everything else is the product path: the `omgx_admm_*_ex` kernels that read and write the
exchange buffers in place, the published rows and the residual sums riding in the send buffers, the remapped neighbour
indices.  (An `nccl` run needs two GPUs; the gloo tests of tests/test_admm_cpu.py cover the collectives themselves.)

Checks: the same iterates as the one-rank fleet, three launches (x-update with the centre step in its epilogue, z / l
update, read-back) and two collectives per iteration, and the accelerated
(Nesterov) iteration -- which takes the general exchange path on the same view-backed buffers."""
import os
import sys
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


class ThreadDist(object):
    """all_gather_into_tensor / all_reduce between the threads of one process (same device, same stream)."""

    class ReduceOp(object):
        SUM = 'sum'

    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world

    def for_rank(self, rank):
        outer = self

        class _D(object):
            ReduceOp = ThreadDist.ReduceOp

            def all_gather_into_tensor(self, out, inp):
                outer.slots[rank] = inp
                outer.barrier.wait()                      # every rank has enqueued what produces its rows
                rows = inp.shape[0]
                for r in range(outer.world):
                    out[r * rows:(r + 1) * rows].copy_(outer.slots[r])
                outer.barrier.wait()                      # nobody overwrites its send buffer before all have read it

            def all_reduce(self, t, op=None):
                outer.slots[rank] = t.clone()
                outer.barrier.wait()
                total = sum(outer.slots[r] for r in range(outer.world))
                outer.barrier.wait()
                t.copy_(total)
        return _D()


def _run_sharded(tpl, lay, P, n, world, iters, kw, t_rels):
    import torch
    from omgtools.admm import BatchADMM, HipAdmmOps
    from omgtools.backend import BatchSolver
    from omgtools.distributed import shard_range
    dev = torch.device('cuda', 0)
    td = ThreadDist(world)
    out, errs = [None] * world, []

    def work(rank):
        try:
            torch.cuda.set_device(0)
            lo, hi = shard_range(n, rank, world)
            solver = BatchSolver(tpl, hi - lo, options=dict(tol=1e-6, max_iter=200))
            ops = HipAdmmOps(solver, tpl, lay, P['p'][lo:hi], P['x0'][lo:hi], dev)
            admm = BatchADMM(lay, P['nbr'], ops, rank=rank, world=world, dist=td.for_rank(rank), rho=1.0, **kw)
            admm.initialize()
            counts = []
            for it in range(iters):
                l0, c0 = ops.launches, ops.collectives
                admm.iterate(t_rels[it], sync=False)
                counts.append((ops.launches - l0, ops.collectives - c0))
            torch.cuda.synchronize()
            out[rank] = dict(x=ops.x.cpu().numpy(), p=ops.p.cpu().numpy(), z=ops.z_ij.cpu().numpy(), l=ops.l_ij.cpu().numpy(),
                             res=np.array(admm.residuals), counts=counts, fused=ops.fused)
            solver.close()
        except Exception as e:                            # pragma: no cover
            errs.append((rank, repr(e)))
            td.barrier.abort()
    threads = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errs, errs
    return out


@pytest.mark.parametrize('kw', [dict(), dict(nesterov_acceleration=True)])
def test_two_ranks_on_one_gpu_match_the_single_rank_fleet(kw):
    import torch
    from test_admm_cpu import _scenario
    from omgtools.admm import BatchADMM, HipAdmmOps
    from omgtools.backend import BatchSolver
    n, iters = 8, 4
    t_rels = [0.0, 0.0, 0.1, 0.2]
    tpl, lay, P = _scenario(n)
    solver = BatchSolver(tpl, n, options=dict(tol=1e-6, max_iter=200))
    ops = HipAdmmOps(solver, tpl, lay, P['p'], P['x0'], torch.device('cuda', 0))
    ref = BatchADMM(lay, P['nbr'], ops, rho=1.0, **kw)
    ref.initialize()
    for it in range(iters):
        ref.iterate(t_rels[it], sync=False)
    x_ref, p_ref, z_ref, l_ref = (a.cpu().numpy() for a in (ops.x, ops.p, ops.z_ij, ops.l_ij))
    res_ref = np.array(ref.residuals)
    solver.close()
    got = _run_sharded(tpl, lay, P, n, 2, iters, kw, t_rels)
    assert all(g['fused'] for g in got)
    x = np.concatenate([g['x'] for g in got]); p = np.concatenate([g['p'] for g in got])
    z = np.concatenate([g['z'] for g in got]); l = np.concatenate([g['l'] for g in got])
    # every agent sees the same inputs as in the one-rank fleet: the same bits come out (only the fleet sums of the
    # residuals are added up in another order)
    assert np.array_equal(x, x_ref) and np.array_equal(z, z_ref) and np.array_equal(l, l_ref)
    cons = np.r_[lay.p_zi:lay.p_zi + lay.ns, lay.p_li:lay.p_li + lay.ns,
                 lay.p_zji:lay.p_zji + lay.n_nghb * lay.ns, lay.p_lji:lay.p_lji + lay.n_nghb * lay.ns]
    assert np.array_equal(p[:, cons], p_ref[:, cons])
    for g in got:
        assert np.allclose(g['res'], res_ref, rtol=1e-12, atol=1e-15)
        if not kw:
            assert all(c == (3, 2) for c in g['counts']), g['counts']
