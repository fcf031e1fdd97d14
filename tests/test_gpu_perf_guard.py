"""Regression guard of the iteration counts behind the bench lines (GPU tier): the point-to-point headline protocol, the
formation protocol and the rendez-vous protocol of bench.py, each compared with the committed values of
tests/golden/perf_guard.json (10 %).  Time is not asserted (boxes differ by 20 %); iteration counts are deterministic, and
they are what a solver change moves first."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'perf_guard.json')


def _expect():
    return json.load(open(HERE))


def _check(key, value):
    if os.environ.get('OMGX_GUARD_WRITE'):
        d = _expect()
        d[key] = round(float(value), 4)
        json.dump(d, open(HERE, 'w'), indent=1)
        return
    want = _expect()[key]
    assert abs(value - want) <= 0.10 * want, (key, value, want)


def test_point_to_point_protocol_iteration_counts():
    import torch
    from omgtools import workloads
    from omgtools.batch import BatchP2P
    problem, P = workloads.holonomic_p2p(1024)
    mpc = BatchP2P(problem, P, ops='hip', device=torch.device('cuda', 0), options=dict(tol=1e-3, max_iter=300))
    mpc.solve_cold()
    assert (mpc.host('status') == 0).all()
    _check('p2p_cold_mean_iters', mpc.host('iters').mean())
    for _ in range(3):
        mpc.step()
    its, worst = [], 0
    for _ in range(20):
        mpc.step()
        it = mpc.host('iters')
        assert (mpc.host('status') == 0).all()
        its.append(it.mean()); worst = max(worst, int(it.max()))
    _check('p2p_step_mean_iters', np.mean(its))
    assert worst <= 24                                   # (round 3: 18 on a knot-crossing step)


@pytest.mark.parametrize('kind', ['formation', 'rendezvous'])
def test_consensus_protocol_iteration_counts(kind):
    import torch
    from omgtools import workloads
    from omgtools.backend import BatchSolver
    from omgtools.admm import BatchADMM, HipAdmmOps, FormationMPC
    N = 512
    rendezvous = kind == 'rendezvous'
    problem, updater, father, lay, P = (workloads.rendezvous_holonomic if rendezvous else workloads.formation_holonomic)(N)
    tpl = father.template
    dev = torch.device('cuda', 0)
    solver = BatchSolver(tpl, N, options=dict(tol=1e-3, max_iter=300))
    ops = HipAdmmOps(solver, tpl, lay, P['p'], P['x0'], dev)
    admm = BatchADMM(lay, P['nbr'], ops, rho=2.0 if rendezvous else 1.0)
    moving = []
    for obs in problem.environment.obstacles:
        ox, ov, oa = (tpl.entry_range(obs.label, nm, 'par') for nm in ('x', 'v', 'a'))
        if np.any(P['p'][:, ov[0]:ov[1]] != 0.) or np.any(P['p'][:, oa[0]:oa[1]] != 0.):
            moving.append((ox[0], ov[0], oa[0], ox[1] - ox[0]))
    mpc = FormationMPC(admm, father, tpl, lay, problem.vehicles[0], obstacles=moving, update_time=0.1, init_iter=5,
                       knot_time=problem.knot_time, consensus_is_spline=not rendezvous)
    mpc.initialize()
    for _ in range(5):
        mpc.step()
    steps = 50
    stats = torch.zeros((steps, 4), dtype=torch.int64, device=dev)
    solver.set_stats(stats)
    for _ in range(steps):
        status, _ = mpc.step()
    torch.cuda.synchronize()
    solver.set_stats(None)
    stats = stats.cpu().numpy()
    assert int((status == 0).sum().item()) == N
    _check(kind + '_x_update_mean_iters', float(stats[:, 1].sum()) / max(1, int(stats[:, 3].sum())))
    solver.close()


def test_whole_manoeuvre_with_the_stop_rule():
    """bench.py's `sustained` leg as a guard: cold solve, then 120 updates of the 1024-agent batch on the three-stream product path with
    the stop rule of the solve kernel on (`omgx_batch_set_stop`: a vehicle's loop ends where the reference's `Simulator.run` ends it).
    Every solve converges, every vehicle arrives, the solves the kernels count and the mean iterations are the committed ones, and the
    update at t = T = 10.0 -- where the reference's knot clock contradicts itself, `omgtools.splines.since_knot` -- is an ordinary
    crossing (slowest agent 62 iterations with the verbatim clock)."""
    import torch
    from omgtools import workloads
    from omgtools.batch import receding_horizon_batch, StreamedP2P
    dev = torch.device('cuda', 0)
    problem, P = workloads.holonomic_p2p(1024)
    rh = receding_horizon_batch(problem, P, device=dev, n_streams='auto', options=dict(tol=1e-3, max_iter=300))
    assert isinstance(rh, StreamedP2P) and len(rh.parts) == 3
    try:
        rh.solve_cold(bends=())
        rh.stop_at_arrival(1e-3)
        n = 120
        stats = [torch.zeros((n, 4), dtype=torch.int64, device=dev) for _ in rh.parts]
        for m, sd in zip(rh.parts, stats):
            m.solver.set_stats(sd)
        for _ in range(n):
            rh.step()
        rh.synchronize()
        st = sum(sd.cpu().numpy() for sd in stats).astype(float)
        worst = np.max([sd.cpu().numpy()[:, 2] for sd in stats], axis=0)
        for m in rh.parts:
            m.solver.set_stats(None)
        assert (st[:, 0] == st[:, 3]).all()                                   # every solve converged
        assert st[0, 3] == 1024 and st[-1, 3] == 0 and (np.diff(st[:, 3]) <= 0).all()      # loops end, none starts again
        assert int(rh.under_way.sum().item()) == 0
        _check('p2p_manoeuvre_solves', st[:, 3].sum())
        _check('p2p_manoeuvre_mean_iters', st[:, 1].sum() / st[:, 3].sum())
        assert worst.max() <= 60 and worst[99] <= 4                          # (update 99: t = 10.0)
    finally:
        rh.close()


def test_sub_batch_streams_are_shared_and_have_a_queue_level_of_their_own():
    """`omgtools.batch.sub_batch_streams`: the sub-batches of every `StreamedP2P` of a device run on the same high-priority streams (the HIP
    runtime maps streams onto four hardware queues per priority level at first use; on the default level two sub-batches could share
    a queue depending on what the process had created before: 1.2 M instead of 2.2 M solves/s, profiles/r06_stream_placement.txt)."""
    import torch
    from omgtools import workloads
    from omgtools.batch import StreamedP2P, sub_batch_streams
    dev = torch.device('cuda', 0)
    s3 = sub_batch_streams(dev, 3)
    assert len(s3) == 3 and len({s.cuda_stream for s in s3}) == 3 and all(s.priority < 0 for s in s3)
    assert [s.cuda_stream for s in sub_batch_streams(dev, 2)] == [s.cuda_stream for s in s3[:2]]
    problem, P = workloads.holonomic_p2p(12)
    a = StreamedP2P(problem, P, n_streams=3, device=dev, options=dict(tol=1e-3, max_iter=300))
    b = StreamedP2P(problem, P, n_streams=2, device=dev, options=dict(tol=1e-3, max_iter=300))
    try:
        assert [s.cuda_stream for s in a.streams] == [s.cuda_stream for s in s3]
        assert [s.cuda_stream for s in b.streams] == [s.cuda_stream for s in s3[:2]]
        for m in (a, b):                                  # (two instances on the same streams: each a correct loop of its own)
            m.solve_cold(bends=())
        for _ in range(3):
            a.step(); b.step()
        assert np.array_equal(a.x.cpu().numpy(), b.x.cpu().numpy()) and (a.status.cpu().numpy() == 0).all()
    finally:
        a.close(); b.close()
