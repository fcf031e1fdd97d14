"""The formation-ADMM path pinned to the reference: tests/golden/admm_formation.npz holds values produced by
EXECUTING the reference's own `problems/formation.py`, `distributedproblem.py`, `dualmethod.py` and
`admm.py` (generator tests/golden/generate_golden_admm.py, casadi stand-in of omgx_shim evaluated on numbers)
for `examples/formation_holonomic.py` (4 vehicles, two rectangles, the moving circle).  Compared here, at
1e-9: the x-update NLP of this repository's front end (`formation.build_updx_template`: layout, bounds, f
and g at seeded random points), the z-update (coupling matrix A as a projector, b = 0, outputs of the
closed form), the multiplier update and the residuals (product formulas in `formation.py`, the numpy
statement the GPU test compares the HIP kernels with: tests/admm_numpy_ops.py / oracle/admm_numpy.py)."""
import os

import numpy as np
import pytest

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'admm_formation.npz')


@pytest.fixture(scope='module')
def fix():
    return np.load(FIX)


@pytest.fixture(scope='module')
def updx():
    import omgtools.backend as be
    from omgtools import Holonomic, Environment, Obstacle, Rectangle, Circle, Square
    from omgtools.formation import build_updx_template
    saved = be.create_nlp
    be.create_nlp = lambda tpl, opt, name='': (None, 0.)
    try:
        vehicle = Holonomic()
        vehicle.set_initial_conditions([-1.5, -1.7])
        vehicle.set_terminal_conditions([2., 1.8])
        environment = Environment(room={'shape': Square(5.)})
        rectangle = Rectangle(width=3., height=0.2)
        environment.add_obstacle(Obstacle({'position': [-2.1, -0.5]}, shape=rectangle))
        environment.add_obstacle(Obstacle({'position': [1.7, -0.5]}, shape=rectangle))
        environment.add_obstacle(Obstacle({'position': [1.5, 0.5]}, shape=Circle(0.4)))
        return build_updx_template(vehicle, environment, 2, {'horizon_time': 10.})
    finally:
        be.create_nlp = saved


def _ordinal_labels(names):
    """Object labels carry per-process counters (vehicle8 in a process that built vehicles before): renumber
    them in order of first appearance."""
    import re
    seen = {}
    return [re.sub(r'(vehicle|obstacle|environment)(\d+)',
                   lambda m: m.group(1) + str(seen.setdefault(m.group(0), len(seen))), n) for n in names]


def test_xupdate_nlp_equals_the_reference(fix, updx):
    """`ADMM.construct_upd_x` (`problems/admm.py:63-115`): same variables, parameters, rows, bounds and the
    same f, g as the reference's graphs."""
    from oracle.nlp_numpy import NumpyNLP
    problem, updater, father = updx
    tpl = father.template
    assert (tpl.n_var, tpl.n_con, tpl.n_par) == (fix['updx_x0'].size, fix['updx_lb'].size, fix['updx_p0'].size) == (151, 671, 212)
    for which, tag in (('var', 'var'), ('par', 'par'), ('con', 'con')):
        mine = [(name, off, r, c) for (_, name, off, r, c) in tpl.block_table(which)]
        ref = [(str(n).split('/')[-1], int(o), int(r), int(c)) for n, (o, r, c) in zip(fix['updx_%s_names' % tag], fix['updx_%s_layout' % tag])]
        assert [(o, r, c) for _, o, r, c in mine] == [(o, r, c) for _, o, r, c in ref], which
        if which != 'con':                                       # (constraint names carry per-process counters)
            assert _ordinal_labels([n for n, _, _, _ in mine]) == _ordinal_labels([n for n, _, _, _ in ref]), which
    assert np.array_equal(tpl.lb, fix['updx_lb']) and np.array_equal(tpl.ub, fix['updx_ub'])
    nlp = NumpyNLP(tpl)
    for xv, pv, fr, gr in zip(fix['updx_xs'], fix['updx_ps'], fix['updx_fs'], fix['updx_gs']):
        f, g = nlp.fg(xv, nlp.term_coefs(pv))
        assert abs(f - fr) < 1e-9 * (1 + abs(fr))
        assert np.abs(g - gr).max() < 1e-9 * (1 + np.abs(gr).max())


def test_zupdate_coupling_equals_the_reference(fix, updx):
    """`_check_for_lineq` (`admm.py:313-354`): A spans the same constraints (same projector on its null
    space; scaling and order of the rows are immaterial), b = 0; neighbour order "next, previous"
    (`distributedproblem.py:181-182`); the closed-form z-update (`admm.py:117-168`) gives the reference's
    outputs, also for t > 0 (forward / backward knot transform)."""
    from omgtools.formation import coupling_matrix, zupdate_matrices
    problem, updater, father = updx
    basis = problem.vehicles[0].basis
    L, d = len(basis), basis.degree
    assert int(fix['n_shared']) == 2 * L == 26
    assert list(fix['nghb_index']) == [1, 3]
    assert np.abs(fix['updz_b']).max() == 0.0
    P_term = [basis.derivative(o)[1][-1, :] for o in range(1, d + 1)]
    A = coupling_matrix(L, 2, d, 2, P_term)
    Ar = fix['updz_A']
    assert A.shape == Ar.shape == (58, 78)

    def projector(M):
        return np.eye(M.shape[1]) - M.T @ np.linalg.solve(M @ M.T, M)
    assert np.abs(projector(A) - projector(Ar)).max() < 1e-9
    ns, nij = 26, 52
    for vin, vout in zip(fix['updz_in'], fix['updz_out']):
        x_i, l_i, l_ij, x_j = vin[:ns], vin[ns:2 * ns], vin[2 * ns:2 * ns + nij], vin[2 * ns + nij:2 * ns + 2 * nij]
        t, T, rho = vin[-3:]
        M, F = zupdate_matrices(basis, 2, 2, t / T)
        z_all = M @ (np.r_[x_i, x_j] + np.r_[l_i, l_ij] / rho)
        assert np.abs(z_all - vout).max() < 1e-9 * (1 + np.abs(vout).max())


def test_lambda_update_and_residuals_equal_the_reference(fix, updx):
    """`construct_upd_l` (`admm.py:248-268`): l <- l + rho (x - z) on untransformed coefficients;
    `construct_upd_res` (`admm.py:270-307`): pr, dr, cr on forward-shifted coefficients."""
    from omgtools.formation import zupdate_matrices
    problem, updater, father = updx
    basis = problem.vehicles[0].basis
    ns, nij = 26, 52
    for vin, vout in zip(fix['updl_in'], fix['updl_out']):
        x_i, z_i = vin[:ns], vin[ns:2 * ns]
        z_ij = vin[2 * ns:2 * ns + nij]
        l_i = vin[2 * ns + nij:3 * ns + nij]
        l_ij = vin[3 * ns + nij:3 * ns + 2 * nij]
        x_j = vin[3 * ns + 2 * nij:3 * ns + 3 * nij]
        rho = vin[-1]
        out = np.r_[l_i + rho * (x_i - z_i), l_ij + rho * (x_j - z_ij)]
        assert np.abs(out - vout).max() < 1e-12 * (1 + np.abs(vout).max())
    for vin, vout in zip(fix['res_in'], fix['res_out']):
        o = 0
        parts = []
        for n in (ns, ns, ns, nij, nij, nij):
            parts.append(vin[o:o + n]); o += n
        x_i, z_i, z_i_p, z_ij, z_ij_p, x_j = parts
        t, T, rho = vin[-3:]
        _, F = zupdate_matrices(basis, 2, 2, t / T)
        assert np.array_equal(F, np.eye(78))         # as executed, the reference measures them untransformed
        fw = lambda a, b: F @ np.r_[a, b]
        pr = np.sum((fw(x_i, x_j) - fw(z_i, z_ij)) ** 2)
        dr = rho * np.sum((fw(z_i, z_ij) - fw(z_i_p, z_ij_p)) ** 2)
        assert np.abs(np.array([pr, dr, rho * pr + dr]) - vout).max() < 1e-9 * (1 + np.abs(vout).max())


# ---- round 4: a sequence of x-updates against an independent solver ------------------------------------------------------
def _check_xupdates(solve_step, f_tol=1e-5, x_tol=1e-3, tight_tol=3e-4, tight_share=0.9):
    """tests/golden/sol_admm_xupdate.npz (generator tests/golden/generate_multistart.py `run_formation`): twelve updates of a
    six-vehicle formation in the receding-horizon protocol of bench.py --workload formation (one knot crossing, the moving
    circle), the inputs of every x-update dumped -- parameters with the consensus state z, l and rho, the warm-start plan,
    the multipliers -- and every one of these NLPs (`problems/admm.py:390`: 151 variables, 671 rows) solved by scipy SLSQP.
    A warm-started product solve from the dumped inputs must return SLSQP's solution."""
    import os
    import omgtools.backend as be
    from omgtools import scenarios
    from oracle.nlp_numpy import NumpyNLP
    from oracle.kkt_check import assert_kkt
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'sol_admm_xupdate.npz'))
    steps, n = d['x'].shape[:2]
    saved = be.create_nlp
    be.create_nlp = lambda tpl, opt, name='': (None, 0.)
    try:
        problem, updater, father, lay, P = scenarios.formation_holonomic(n)
    finally:
        be.create_nlp = saved
    tpl = father.template
    assert (int(d['n_var']), int(d['n_con'])) == (tpl.n_var, tpl.n_con) == (151, 671) and d['ok'].all() and d['crossed'].sum() == 1
    nlp = NumpyNLP(tpl)
    lo, hi = d['spl']
    worst_f = worst_x = 0.0
    tight = total = 0
    for k in range(steps):
        res = solve_step(tpl, d['p'][k], d['x0'][k], d['lam'][k])
        assert (res['status'] == 0).all(), (k, res['status'])
        for b in range(n):
            assert_kkt(nlp, tpl, d['p'][k, b], res['x'][b], res['lam_g'][b], 1e-5, ('x-update', k, b))
            f = nlp.fg(res['x'][b], nlp.term_coefs(d['p'][k, b]))[0]
            worst_f = max(worst_f, abs(f - d['f'][k, b]) / (1 + abs(f)))
            dx = np.abs(res['x'][b, lo:hi] - d['x'][k, b, lo:hi]).max()
            worst_x = max(worst_x, dx)
            tight += dx < tight_tol
            total += 1
    assert worst_f < f_tol, worst_f
    assert worst_x < x_tol, worst_x
    assert tight >= tight_share * total, (tight, total)
    return worst_f, worst_x, tight, total


def test_port_xupdates_match_slsqp():
    from oracle import port_binding

    def solve_step(tpl, p, x0, lam):
        return port_binding.solve(tpl, p, x0, lam_g0=lam, status0=np.zeros(len(p), dtype=np.int32), warm_start=1, n_threads=4,
                                  tol=1e-6, max_iter=500, warm_z_cap=0.0, max_soc=0)
    _check_xupdates(solve_step)


@pytest.mark.gpu
def test_hip_xupdates_match_slsqp():
    from omgtools.backend import BatchSolver
    solver = {}

    def solve_step(tpl, p, x0, lam):
        if 's' not in solver:
            solver['s'] = BatchSolver(tpl, len(p), options=dict(tol=1e-6, max_iter=500, warm_start=1, warm_z_cap=0.0, max_soc=0))
        return solver['s'].solve(p, x0, lam_g0=lam, status0=np.zeros(len(p), dtype=np.int32))
    try:
        _check_xupdates(solve_step)
    finally:
        if 's' in solver:
            solver['s'].close()
