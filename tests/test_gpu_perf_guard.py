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
