"""Configs whose per-agent arrays exceed one CU's LDS (cfg 3 Quadrotor K=13 O=5, cfg 5 Holonomic3D
K=15 O=10): the library spills the KKT store / Jacobian / row arrays to HBM
slabs (`omgx_batch_workspace` modes 1-3).  Same iteration, so the iterates must match the host
port (oracle) like the all-LDS path does."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _build(fn, B):
    import omgtools.backend as be
    saved = be.create_nlp
    be.create_nlp = lambda tpl, opt, name='': (None, 0.)      # template only; the batch solver is made below
    try:
        return fn(B)
    finally:
        be.create_nlp = saved


@pytest.mark.parametrize('name,B', [('quadrotor_p2p', 6), ('holonomic3d_p2p', 5)])
def test_spilled_fixed_iterations_match_port(name, B):
    from omgtools import scenarios
    from omgtools.backend import BatchSolver
    from oracle import port_binding
    problem, P = _build(getattr(scenarios, name), B)
    tpl = problem.father.template
    opts = dict(tol=1e-300, max_iter=10)
    solver = BatchSolver(tpl, B, options=opts)
    ws = solver.workspace()
    assert ws['mode'] >= 1 and ws['lds_bytes'] <= 160 * 1024 and ws['hbm_bytes_per_slab'] > 0
    res = solver.solve(P['p'], P['x0'])
    ref = port_binding.solve(tpl, P['p'], P['x0'], **opts)
    assert np.array_equal(res['iters'], ref['iters'])
    scale = 1 + np.abs(ref['x']).max()
    assert np.abs(res['x'] - ref['x']).max() < 1e-7 * scale
    solver.close()


def test_cfg5_converges_like_port():
    from omgtools import scenarios
    from omgtools.backend import BatchSolver
    from oracle import port_binding
    from oracle.nlp_numpy import NumpyNLP
    B = 6
    problem, P = _build(scenarios.holonomic3d_p2p, B)
    tpl = problem.father.template
    opts = dict(tol=1e-5, max_iter=300)
    solver = BatchSolver(tpl, B, options=opts)
    res = solver.solve(P['p'], P['x0'])
    ref = port_binding.solve(tpl, P['p'], P['x0'], **opts)
    assert (res['status'] == ref['status']).sum() >= B - 1
    good = (res['status'] == 0) & (ref['status'] == 0)
    assert good.sum() >= B - 2
    nlp = NumpyNLP(tpl)
    for b in np.nonzero(good)[0]:
        c = nlp.term_coefs(P['p'][b])
        f_gpu, g = nlp.fg(res['x'][b], c)
        f_ref, _ = nlp.fg(ref['x'][b], c)
        assert abs(f_gpu - f_ref) < 1e-4 * (1 + abs(f_ref))
        assert np.all(g <= tpl.ub + 1e-4) and np.all(g >= tpl.lb - 1e-4)     # tol x row scaling
    solver.close()


def test_more_agents_than_slabs():
    """Persistent workgroups walk over several agents each: results do not depend on the slab."""
    from omgtools import scenarios
    from omgtools.backend import BatchSolver
    B = 600                                         # > 2 * 256 slabs
    problem, P = _build(scenarios.holonomic3d_p2p, 4)
    tpl = problem.father.template
    p = np.tile(P['p'], (B // 4, 1)); x0 = np.tile(P['x0'], (B // 4, 1))
    solver = BatchSolver(tpl, B, options=dict(tol=1e-3, max_iter=60))
    assert solver.workspace()['n_slabs'] < B
    res = solver.solve(p, x0)
    for k in range(4):
        # every sum of the solve has one owner and a fixed order, the launch slots are handed out by an atomic
        # counter: which workgroup (and which slab) solves an agent must not change a single bit
        assert len(set(res['status'][k::4].tolist())) == 1
        assert np.ptp(res['iters'][k::4]) == 0
        assert np.array_equal(res['x'][k::4], np.broadcast_to(res['x'][k], res['x'][k::4].shape))
        assert np.array_equal(res['lam_g'][k::4], np.broadcast_to(res['lam_g'][k], res['lam_g'][k::4].shape))
    again = solver.solve(p, x0)                     # ... nor a second launch with another assignment
    assert np.array_equal(again['x'], res['x']) and np.array_equal(again['iters'], res['iters'])
    solver.close()


def test_cfg2_safety_distance_variant():
    """SURVEY 8d's s = 1 variant of config 2 (safety_distance > 0: n_var 206, n_con 612): does
    not fit one CU's LDS any more, so it runs with the KKT store in HBM; converged solutions agree
    with the port."""
    from omgtools import scenarios
    from omgtools.backend import BatchSolver
    from oracle import port_binding
    B = 16
    problem, P = _build(lambda n: scenarios.holonomic_p2p(n, safety_distance=0.1), B)
    tpl = problem.father.template
    assert (tpl.n_var, tpl.n_con) == (206, 612)
    opts = dict(tol=1e-3, max_iter=300)
    solver = BatchSolver(tpl, B, options=opts)
    assert solver.workspace()['mode'] >= 1
    res = solver.solve(P['p'], P['x0'])
    ref = port_binding.solve(tpl, P['p'], P['x0'], **opts)
    assert (res['status'] == ref['status']).sum() >= B - 1
    good = (res['status'] == 0) & (ref['status'] == 0)
    assert good.sum() >= B - 3
    lo, hi = tpl.entry_range(problem.vehicles[0].label, 'splines_seg0', 'var')
    assert np.abs(res['x'][good][:, lo:hi] - ref['x'][good][:, lo:hi]).max() < 1e-2
    solver.close()
