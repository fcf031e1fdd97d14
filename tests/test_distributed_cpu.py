"""world_size-2 gloo test of the N>1 path: agents shard contiguously, each rank
solves its block (oracle CPU port standing in for the GPU in this CPU-only test),
the report reduces to max-time / sum-solved and the gathered solutions equal the
single-process result."""
import os
import sys
import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, 'omg-tools_amd'))
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import omgtools.backend as be
    be.create_nlp = lambda tpl, opt, name='': (None, 0.)
    from omgtools.scenarios import holonomic_p2p
    from omgtools.distributed import shard_range, reduce_report, gather_solutions
    from oracle import port_binding
    problem, P = holonomic_p2p(6)
    tpl = problem.father.template
    lo, hi = shard_range(6, rank, world)
    res = port_binding.solve(tpl, P['p'][lo:hi], P['x0'][lo:hi], tol=1e-3, max_iter=100)
    elapsed, solved = reduce_report(1.0 + rank, int((res['status'] == 0).sum()), dist=dist)
    x_all = gather_solutions(res['x'], 6, dist=dist)
    if rank == 0:
        ref = port_binding.solve(tpl, P['p'], P['x0'], tol=1e-3, max_iter=100)
        q.put((elapsed, solved, int((ref['status'] == 0).sum()), float(np.abs(x_all - ref['x']).max())))
    dist.destroy_process_group()


def test_two_rank_sharding():
    from omgtools.distributed import shard_range
    assert [shard_range(7, r, 3) for r in range(3)] == [(0, 3), (3, 5), (5, 7)]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 1000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    elapsed, solved, solved_ref, err = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert elapsed == 2.0                   # max over ranks
    assert solved == solved_ref             # sum over ranks == single-process count
    assert err == 0.0                       # identical solutions, agent order preserved


def _strong_worker(rank, world, port, q):
    """`bench.py --scaling strong`: ONE batch sharded over the ranks, the receding-horizon loop per rank (host build of the
    solver standing in for the GPU), report reduced, plans gathered in agent order."""
    sys.path.insert(0, os.path.join(ROOT, 'omg-tools_amd'))
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from omgtools import workloads
    from omgtools.batch import BatchP2P
    from omgtools.distributed import shard_workload, reduce_report, gather_solutions
    from oracle import port_binding
    n = 7                                            # (uneven split: 4 + 3)
    problem, P_all = workloads.holonomic_p2p(n)
    P, (lo, hi) = shard_workload(P_all, rank, world)
    opts = dict(tol=1e-3, max_iter=300)

    def run(Pw):
        mpc = BatchP2P(problem, Pw, ops=port_binding, options=opts)
        mpc.solve_cold()
        ok = 0
        for _ in range(3):
            mpc.step()
            ok += int((mpc.status == 0).sum())
        return mpc.x, ok
    x, ok = run(P)
    elapsed, solved = reduce_report(0.5 + rank, ok, dist=dist)
    x_all = gather_solutions(x, n, dist=dist)
    if rank == 0:
        x_ref, ok_ref = run(P_all)
        q.put(((lo, hi), elapsed, solved, ok_ref, bool(np.array_equal(x_all, x_ref))))
    dist.destroy_process_group()


def test_strong_scaling_shards_one_batch():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29600 + os.getpid() % 1000
    procs = [ctx.Process(target=_strong_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    block, elapsed, solved, solved_ref, same = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert block == (0, 4)
    assert elapsed == 1.5                   # max over ranks
    assert solved == solved_ref == 21       # 7 agents x 3 steps, all converged, counted once each
    assert same                             # the sharded run reproduces the single-process plans bit for bit
