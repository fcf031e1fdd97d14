"""Products of more than four variable factors and quotients by a variable (the Bicycle / AGV / Trailer models of
`vehicles/bicycle.py:53`, `vehicles/agv.py:50`, `vehicles/trailer.py:28`, the free end time of `examples/p2p_dubins.py` as shipped).

Representation (omgtools/symbolic.py `LIFT_CAP`, template.py `_append_lifted`, include/omgx.h `n_lift`): the front end writes such an
expression with auxiliary variables -- aux = the factor of higher degree, q * den = num -- whose defining equality rows sit behind
the caller's rows.  The solver keeps those rows satisfied exactly: every trial point of its line search takes the auxiliaries
from their rows (omgx_core.h `lift_project`), so its iterates are iterates of the caller's own problem.

  CPU  the mechanism on a small NLP against scipy SLSQP on the UNLIFTED problem; the fixtures generated from the reference's own
       construct code (tests/golden/generate_shim_fixtures.py: its modules executed on `omgx_shim`) reproduce the reference's f / g at
       random points; host build of the solver from the reference's guess on the fixed-T Bicycle and AGV problems: optimality
       conditions of the NLP evaluated by the numpy oracle, the objective against SLSQP on the caller's own problem (AGV), the same
       iterates as the dense numpy statement of the solver; a template-file round trip.
  GPU  the HIP path through the C ABI: derivative tables of the lifted template against the oracle, the solve against the host
       build and the optimality conditions."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


# ---- the mechanism on a small problem ------------------------------------------------------------------------------------------
def _toy():
    """min (x0 - 1)^2 + (x1 - 2)^2 + (x2 - 0.5)^2   s.t.  x0^2 x1^2 x2^2 <= 0.5 (six variable factors),  x0 / x2 >= 1.5 (a quotient by
    a variable),  0.2 <= x2."""
    from omgtools.symbolic import SymbolTable, Poly
    from omgtools.template import NLPTemplate
    table = SymbolTable()
    with table:
        syms = table.new_vars(3)
        x = [Poly.symbol(s) for s in syms]
        rows = [(x[0] * x[0] * x[1]) * (x[1] * x[2] * x[2]), x[0] / x[2], x[2]]
        objective = (x[0] - 1.0) * (x[0] - 1.0) + (x[1] - 2.0) * (x[1] - 2.0) + (x[2] - 0.5) * (x[2] - 0.5)
        tpl = NLPTemplate.from_polys(table, syms, [], rows, objective, lb=[-np.inf, 1.5, 0.2], ub=[0.5, np.inf, np.inf])
    return tpl


def test_lifting_builds_the_rows_and_the_solver_returns_the_minimum_of_the_unlifted_problem():
    from scipy.optimize import minimize
    from oracle import port_binding, ipm_numpy
    from oracle.nlp_numpy import NumpyNLP
    tpl = _toy()
    assert tpl.n_lift == 2 and tpl.n_var == 5 and tpl.n_con == 5 and tpl.t_nv.max() <= 4
    assert tpl.var_layout[('lifted', 'aux')] == (3, 2, 1) and tpl.con_layout[('lifted', 'aux')] == (3, 2, 1)
    nlp = NumpyNLP(tpl)
    assert list(nlp.lift_level) == [0, 0]
    p = np.zeros(0)
    c = nlp.term_coefs(p)
    # the caller's vector extended: the auxiliaries from their rows; the lifted rows then hold and the caller's rows have the
    # values of the original expressions
    rng = np.random.default_rng(5)
    for _ in range(3):
        xu = rng.uniform(0.5, 2.0, size=3)
        xf = tpl.lift_extend(xu, p)[0]
        f, g = nlp.fg(xf, c)
        assert np.abs(g[3:]).max() < 1e-14
        assert abs(g[0] - (xu[0] * xu[1] * xu[2]) ** 2) < 1e-12 and abs(g[1] - xu[0] / xu[2]) < 1e-12
        assert np.abs(nlp.project_lifted(np.r_[xu, 7.0, -3.0], c) - xf).max() < 1e-14
    x0 = np.array([1.0, 1.0, 1.0])
    ref = minimize(lambda v: (v[0] - 1) ** 2 + (v[1] - 2) ** 2 + (v[2] - 0.5) ** 2, x0, method='SLSQP', options={'ftol': 1e-14},
                   constraints=[{'type': 'ineq', 'fun': lambda v: 0.5 - (v[0] * v[1] * v[2]) ** 2},
                                {'type': 'ineq', 'fun': lambda v: v[0] / v[2] - 1.5}, {'type': 'ineq', 'fun': lambda v: v[2] - 0.2}])
    assert ref.success
    xf, lb, ub = tpl.lift_extend(x0, p, tpl.lb[:3], tpl.ub[:3])
    res = port_binding.solve(tpl, p[None], xf[None], lb, ub, tol=1e-8, max_iter=200)
    assert res['status'][0] == 0 and res['iters'][0] < 40
    xs, lam = tpl.lift_strip(res['x'][0], res['lam_g'][0])
    assert xs.shape == (3,) and lam.shape == (3,)
    assert np.abs(xs - ref.x).max() < 1e-5, (xs, ref.x)
    assert np.abs(nlp.fg(res['x'][0], c)[1][3:]).max() < 1e-12            # the defining rows hold exactly at the result
    # whatever a caller hands in for the auxiliaries, the solve starts on their rows: the same bits
    junk = xf.copy()
    junk[3:] = [7.0, -3.0]
    again = port_binding.solve(tpl, p[None], junk[None], lb, ub, tol=1e-8, max_iter=200)
    assert np.array_equal(again['x'], res['x']) and again['iters'][0] == res['iters'][0]
    # iterate for iterate the dense numpy statement of the solver (same projection of the auxiliaries)
    for iters in (2, 6):
        a = port_binding.solve(tpl, p[None], xf[None], lb, ub, tol=1e-12, max_iter=iters)
        b = ipm_numpy.solve(nlp, xf, p, lb, ub, opts={'tol': 1e-12, 'max_iter': iters})
        assert a['iters'][0] == b['iters'] == iters
        assert np.abs(a['x'][0] - b['x']).max() < 1e-8 * max(1.0, np.abs(b['x']).max()), iters


def test_the_library_refuses_a_lifted_row_that_reads_a_later_auxiliary():
    import omgtools.backend as be
    tpl = _toy()
    info = be.describe_plan(tpl)
    assert info['n_eq'] == 2
    bad = _toy()
    # swap the two auxiliaries inside the term lists: the first defining row then reads the second auxiliary
    tv = bad.t_var.copy()
    tv[bad.t_var == 3], tv[bad.t_var == 4] = 4, 3
    r0, r1 = int(bad.row_ptr[3]), int(bad.row_ptr[5])
    keep = bad.t_var.copy()
    keep[r0:r1] = tv[r0:r1]
    bad.t_var = keep
    with pytest.raises(Exception):
        be.describe_plan(bad)


# ---- the reference's classes -----------------------------------------------------------------------------------------------
CLASSES = {'bicycle_fixedT': dict(shape=(293, 854, 25), n_lift=208, levels=3),
           'agv_fixedT': dict(shape=(381, 2234, 52), n_lift=278, levels=3),
           'dubins_freeT': dict(shape=(111, 394, 19), n_lift=76, levels=11),
           'trailer_freeT': dict(shape=(168, 1896, 13), n_lift=127, levels=11),
           # `examples/p2p_dubins.py` exactly as shipped: substituted velocity splines (118 two-sided rows) AND a free end time
           'dubins_shipped': dict(shape=(77, 362, 19), n_lift=16, levels=10)}


def _load(name):
    from omgtools.template import NLPTemplate
    path = os.path.join(GOLDEN, name + '.npz')
    tpl = NLPTemplate.from_npz(path)
    d = np.load(path)
    spec = CLASSES[name]
    assert (tpl.n_var, tpl.n_con, tpl.n_par) == spec['shape'] and tpl.n_lift == spec['n_lift'] and tpl.t_nv.max() <= 4
    return tpl, d


@pytest.mark.parametrize('name', sorted(CLASSES))
def test_lifted_templates_reproduce_the_reference_graphs(name):
    """f and g of the reference's own CasADi graphs (evaluated by the shim on the closures the reference built) at three random
    points against the template with the auxiliaries taken from their rows."""
    from oracle.nlp_numpy import NumpyNLP
    tpl, d = _load(name)
    nlp = NumpyNLP(tpl)
    assert int(nlp.lift_level.max()) + 1 == CLASSES[name]['levels']
    nl = tpl.n_lift
    for xv, pv, fs, gs in zip(d['xs'], d['ps'], d['fs'], d['gs']):
        c = nlp.term_coefs(pv)
        xf = nlp.project_lifted(np.r_[xv[:tpl.n_var - nl], np.zeros(nl)], c)
        assert np.abs(xf - xv).max() < 1e-12 * (1 + np.abs(xv).max())       # (template.lift_extend wrote xv)
        f, g = nlp.fg(xf, c)
        assert abs(f - fs) < 1e-9 * (1 + abs(fs))
        assert np.abs(g - gs).max() < 1e-9 * (1 + np.abs(gs).max())          # (the reference's rows; the defining rows: 0)


def _extended(tpl, d):
    nl = tpl.n_lift
    return tpl.lift_extend(d['x0'], d['p0'], tpl.lb[:tpl.n_con - nl], tpl.ub[:tpl.n_con - nl])


def _check_solution(name, tpl, d, res, tol):
    from oracle.nlp_numpy import NumpyNLP
    from oracle.kkt_check import assert_kkt
    nlp = NumpyNLP(tpl)
    assert res['status'][0] == 0, res['status']
    c = nlp.term_coefs(d['p0'])
    f, g = nlp.fg(res['x'][0], c)
    nl = tpl.n_lift
    assert np.abs(g[tpl.n_con - nl:]).max() < 1e-11                          # the auxiliaries sit on their rows
    assert (g - tpl.ub).max() < 10 * tol and (tpl.lb - g).max() < 10 * tol
    assert_kkt(nlp, tpl, d['p0'], res['x'][0], res['lam_g'][0], 10 * tol, name)
    if int(d['slsqp_ok']):
        # SLSQP on the caller's own problem stops where no step improves its merit function any more (exit 8 on these classes):
        # a feasible point of the same NLP; the interior-point path must do at least as well
        assert f < float(d['f_slsqp']) + 1e-6, (f, float(d['f_slsqp']))
    return f


@pytest.mark.parametrize('name,max_iters', [('bicycle_fixedT', 120), ('agv_fixedT', 120)])
def test_host_build_solves_the_lifted_classes_from_the_reference_guess(name, max_iters):
    from oracle import port_binding, ipm_numpy
    from oracle.nlp_numpy import NumpyNLP
    tpl, d = _load(name)
    x, lb, ub = _extended(tpl, d)
    res = port_binding.solve(tpl, d['p0'][None], x[None], lb, ub, tol=1e-3, max_iter=500)
    assert res['iters'][0] <= max_iters, res['iters']
    _check_solution(name, tpl, d, res, 1e-3)
    if name == 'bicycle_fixedT':
        nlp = NumpyNLP(tpl)
        a = port_binding.solve(tpl, d['p0'][None], x[None], lb, ub, tol=1e-12, max_iter=12)
        b = ipm_numpy.solve(nlp, x, d['p0'], lb, ub, opts={'tol': 1e-12, 'max_iter': 12})
        assert np.abs(a['x'][0] - b['x']).max() < 1e-7 * max(1.0, np.abs(b['x']).max())


@pytest.mark.parametrize('name', ['bicycle_fixedT', 'agv_fixedT'])
def test_second_order_condition_at_the_lifted_minima(name):
    """Round 6: no second solver confirms the minima of the lifted classes (SLSQP and trust-constr wander off on these NLPs) -- a
    solver-independent statement does: at the product's solution (host build, tol 1e-6) the first-order conditions hold AND the
    Lagrangian Hessian of the reference's NLP, restricted to the tangent space of the active rows, has no negative eigenvalue
    (`oracle.kkt_check.second_order_report`: numpy restatement of the NLP only).  Not a saddle point, not a maximum; the L1 objective
    leaves flat directions, so the smallest eigenvalue is zero, not positive."""
    from oracle import port_binding
    from oracle.nlp_numpy import NumpyNLP
    from oracle.kkt_check import assert_kkt, second_order_report
    tpl, d = _load(name)
    x, lb, ub = _extended(tpl, d)
    res = port_binding.solve(tpl, d['p0'][None], x[None], lb, ub, tol=1e-6, max_iter=1200)
    assert res['status'][0] == 0, res['status']
    nlp = NumpyNLP(tpl)
    assert_kkt(nlp, tpl, d['p0'], res['x'][0], res['lam_g'][0], 1e-5, name)
    lo, hi, dim, n_act = second_order_report(nlp, tpl, d['p0'], res['x'][0], res['lam_g'][0], lb, ub)
    print('\n%s: %d active rows, tangent space of dimension %d, reduced Lagrangian Hessian eigenvalues in [%.2e, %.2e]' % (name, n_act, dim, lo, hi))
    assert dim > 0 and lo > -1e-8 * max(1.0, hi) - 1e-10, (lo, hi, dim, n_act)


def test_template_file_round_trip_keeps_the_lifted_rows(tmp_path):
    import omgtools.backend as be
    tpl = _toy()
    path = str(tmp_path / 'toy.omgx')
    be.save_template(tpl, path)
    with open(path, 'rb') as fh:
        assert fh.read(8) == b'OMGXTPL5'
    back = be.read_template_counts(path)
    assert back['n_lift'] == 2 and back['lift_row0'] == 3 and back['n_var'] == 5 and back['n_con'] == 5
    # a template without auxiliaries is written as before
    from omgtools import workloads
    problem, P = workloads.holonomic_p2p(2)
    path4 = str(tmp_path / 'cfg2.omgx')
    be.save_template(problem.father.template, path4)
    with open(path4, 'rb') as fh:
        assert fh.read(8) == b'OMGXTPL4'


# ---- `examples/p2p_dubins.py` as shipped ---------------------------------------------------------------------------------------
# The example asks IPOPT for a limited-memory Hessian (`examples/p2p_dubins.py:41-42`); `omgtools.backend.options_from_problem` maps
# that onto `hess_approx` (include/omgx.h: the Hessian without the curvature of the rows, damped by the accepted step length).  With
# the exact Hessian phase I of this problem drowns in an inertia correction of 4e6 and is given up at t = 0.25; with
# the option the first solve of the reference's Simulator -- and its next six updates, HISTORY.md -- end in Solve_Succeeded.
def test_the_reference_option_for_a_limited_memory_hessian_is_mapped():
    import omgtools.backend as be
    kw = be.options_from_problem({'solver': 'ipopt', 'solver_options': {'ipopt': {'ipopt.hessian_approximation': 'limited-memory',
                                                                                  'ipopt.tol': 1e-4}}})
    assert kw['hess_approx'] == 1 and kw['tol'] == 1e-4
    assert 'hess_approx' not in be.options_from_problem({'solver': 'ipopt', 'solver_options': {'ipopt': {'ipopt.tol': 1e-3}}})
    assert be.make_options().hess_approx == 0 and be.make_options(hess_approx=1).hess_approx == 1


def _check_shipped(tpl, d, res, tol):
    from oracle.nlp_numpy import NumpyNLP
    from oracle.kkt_check import assert_kkt
    nlp = NumpyNLP(tpl)
    assert res['status'][0] == 0, res['status']
    f, g = nlp.fg(res['x'][0], nlp.term_coefs(d['p0']))
    nl = tpl.n_lift
    assert np.abs(g[tpl.n_con - nl:]).max() < 1e-11
    assert (g - tpl.ub).max() < 10 * tol and (tpl.lb - g).max() < 10 * tol       # (inside every tube of the substituted model)
    assert_kkt(nlp, tpl, d['p0'], res['x'][0], res['lam_g'][0], 10 * tol, 'dubins_shipped')
    T = res['x'][0][tpl.entry_range('p2p0', 'T', 'var')[0]]
    assert 6.0 < T < 10.5 and abs(f - T) < 1e-9                                 # (the objective is the motion time)
    return f


def test_host_build_solves_the_dubins_example_as_shipped():
    from oracle import port_binding
    tpl, d = _load('dubins_shipped')
    two_sided = np.isfinite(tpl.lb) & np.isfinite(tpl.ub) & (tpl.lb < tpl.ub)
    assert two_sided.sum() == 118
    x, lb, ub = _extended(tpl, d)
    res = port_binding.solve(tpl, d['p0'][None], x[None], lb, ub, tol=1e-3, max_iter=3000, hess_approx=1)
    assert res['iters'][0] < 1200, res['iters']
    _check_shipped(tpl, d, res, 1e-3)
    # the exact Hessian on the same problem: phase I is given up (the state of the art of this solver)
    exact = port_binding.solve(tpl, d['p0'][None], x[None], lb, ub, tol=1e-3, max_iter=3000)
    assert exact['status'][0] == 2


# ---- the AGV in closed loop ---------------------------------------------------------------------------------------------------
# tests/golden/agv_loop.npz (generate_shim_fixtures.py `agv_loop`): the reference's Simulator (`execution/simulator.py:39-52`) on the body of
# `examples/p2p_agv.py` with a fixed horizon, its own classes on the shim -- the solve before the loop and twelve updates, every one
# from the reference's warm start (the shifted previous plan as x0, no multipliers: `problems/problem.py:57-60,113`), all thirteen
# Solve_Succeeded on the host build (35-88 iterations).  Stored: p, x0, bounds and the result of every solve.
def _loop():
    d = np.load(os.path.join(GOLDEN, 'agv_loop.npz'))
    tpl, _ = _load('agv_fixedT')
    assert d['p'].shape == (13, tpl.n_par) and d['x0'].shape == (13, tpl.n_var) and (d['status'] == 0).all()
    return tpl, d


def test_host_build_reproduces_an_update_of_the_agv_loop():
    from oracle import port_binding
    from oracle.nlp_numpy import NumpyNLP
    from oracle.kkt_check import assert_kkt
    tpl, d = _loop()
    k = 7                                                                        # (the shortest solve of the loop)
    res = port_binding.solve(tpl, d['p'][k][None], d['x0'][k][None], d['lbg'], d['ubg'], tol=1e-3, max_iter=500)
    assert res['status'][0] == 0 and res['iters'][0] == d['iters'][k]
    assert np.abs(res['x'][0] - d['x'][k]).max() < 1e-9
    assert_kkt(NumpyNLP(tpl), tpl, d['p'][k], res['x'][0], res['lam_g'][0], 1e-2, ('agv loop', k))


# ---- GPU -------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_the_dubins_example_as_shipped_on_the_device():
    import omgtools.backend as be
    from oracle import port_binding
    tpl, d = _load('dubins_shipped')
    x, lb, ub = _extended(tpl, d)
    solver = be.BatchSolver(tpl, 2, options=dict(tol=1e-3, max_iter=3000, hess_approx=1))
    try:
        res = solver.solve(np.repeat(d['p0'][None], 2, axis=0), np.repeat(x[None], 2, axis=0), lbg=lb, ubg=ub)
    finally:
        solver.close()
    assert np.array_equal(res['x'][0], res['x'][1])
    f = _check_shipped(tpl, d, res, 1e-3)
    port = port_binding.solve(tpl, d['p0'][None], x[None], lb, ub, tol=1e-3, max_iter=3000, hess_approx=1)
    print('\ndubins as shipped: HIP %d iterations, host build %d; T = %.6f / %.6f' % (res['iters'][0], port['iters'][0], f,
                                                                                    port['x'][0][tpl.entry_range('p2p0', 'T', 'var')[0]]))
    assert abs(f - port['x'][0][tpl.entry_range('p2p0', 'T', 'var')[0]]) < 2e-2


@pytest.mark.gpu
def test_the_agv_loop_as_one_batch_on_the_device():
    """The thirteen solves of the closed loop as ONE batch through the C ABI (thirteen agents, each with the parameters and the
    warm start of its update): every one succeeds, lands where the host build landed and satisfies the optimality conditions."""
    import omgtools.backend as be
    from oracle.nlp_numpy import NumpyNLP
    from oracle.kkt_check import assert_kkt
    tpl, d = _loop()
    nlp = NumpyNLP(tpl)
    B = len(d['p'])
    solver = be.BatchSolver(tpl, B, options=dict(tol=1e-3, max_iter=500))
    try:
        res = solver.solve(d['p'], d['x0'], lbg=d['lbg'], ubg=d['ubg'])
        ms = solver.last_kernel_ms() if hasattr(solver, 'last_kernel_ms') else float('nan')
    finally:
        solver.close()
    assert (res['status'] == 0).all(), res['status']
    nv = tpl.n_var - tpl.n_lift
    worst_x, worst_f = 0.0, 0.0
    for k in range(B):
        c = nlp.term_coefs(d['p'][k])
        f, fh = nlp.fg(res['x'][k], c)[0], nlp.fg(d['x'][k], c)[0]
        worst_f = max(worst_f, abs(f - fh) / (1 + abs(fh)))
        worst_x = max(worst_x, np.abs(res['x'][k][:nv] - d['x'][k][:nv]).max())
        assert_kkt(nlp, tpl, d['p'][k], res['x'][k], res['lam_g'][k], 1e-2, ('agv loop', k))
    print('\nAGV loop on the device: iterations %s (host build %s); objective within %.1e, the caller\'s variables within %.1e of the host build'
          % (res['iters'].tolist(), d['iters'].tolist(), worst_f, worst_x))
    assert np.abs(res['iters'] - d['iters']).max() <= 10
    assert worst_f < 1e-3 and worst_x < 1e-2


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['bicycle_fixedT', 'agv_fixedT'])
def test_lifted_classes_on_the_device(name):
    import omgtools.backend as be
    from oracle import port_binding
    from oracle.nlp_numpy import NumpyNLP
    tpl, d = _load(name)
    nlp = NumpyNLP(tpl)
    x, lb, ub = _extended(tpl, d)
    rng = np.random.default_rng(17)
    B = 2
    ps = np.repeat(d['p0'][None], B, axis=0)
    xr = x[None] + rng.normal(scale=0.1, size=(B, tpl.n_var))
    lam = rng.normal(size=(B, tpl.n_con))
    solver = be.BatchSolver(tpl, B, options=dict(tol=1e-3, max_iter=500))
    try:
        got = solver.eval(ps, xr, lam)
        res = solver.solve(ps, np.repeat(x[None], B, axis=0), lbg=lb, ubg=ub)
    finally:
        solver.close()
    c = nlp.term_coefs(d['p0'])
    for b in range(B):
        f, g = nlp.fg(xr[b], c)
        J, H = nlp.jac(xr[b], c), nlp.hess(xr[b], lam[b], c)
        assert np.abs(got['g'][b] - g).max() < 1e-10 * max(1.0, np.abs(g).max())
        assert np.abs(got['jac'][b] - J).max() < 1e-10 * max(1.0, np.abs(J).max())
        assert np.abs(got['hess'][b] - H).max() < 1e-10 * max(1.0, np.abs(H).max())
    assert np.array_equal(res['x'][0], res['x'][1])
    f = _check_solution(name, tpl, d, res, 1e-3)
    port = port_binding.solve(tpl, d['p0'][None], x[None], lb, ub, tol=1e-3, max_iter=500)
    fp = nlp.fg(port['x'][0], c)[0]
    print('\n%s: HIP %d iterations, host build %d; objectives %.6f / %.6f' % (name, res['iters'][0], port['iters'][0], f, fp))
    assert abs(int(port['iters'][0]) - int(res['iters'][0])) <= 10
    assert abs(f - fp) < 1e-3 * (1 + abs(f))


@pytest.mark.gpu
def test_toy_on_the_device_matches_the_host_build():
    import omgtools.backend as be
    from oracle import port_binding
    tpl = _toy()
    p = np.zeros((1, 0))
    xf, lb, ub = tpl.lift_extend(np.array([1.0, 1.0, 1.0]), p[0], tpl.lb[:3], tpl.ub[:3])
    solver = be.BatchSolver(tpl, 1, options=dict(tol=1e-8, max_iter=200))
    try:
        res = solver.solve(p, xf[None], lbg=lb, ubg=ub)
        junk = xf.copy()
        junk[3:] = [7.0, -3.0]
        again = solver.solve(p, junk[None], lbg=lb, ubg=ub)
    finally:
        solver.close()
    port = port_binding.solve(tpl, p, xf[None], lb, ub, tol=1e-8, max_iter=200)
    assert res['status'][0] == 0 and res['iters'][0] == port['iters'][0]
    assert np.abs(res['x'][0] - port['x'][0]).max() < 1e-9
    assert np.array_equal(again['x'], res['x'])          # auxiliaries handed in off their rows: the same bits


# ---- the solver object's second attempt ------------------------------------------------------------------------------------------
def test_the_solver_object_takes_a_given_up_phase_one_again_with_the_convexified_hessian():
    """`omgtools.backend.second_attempt` (the drop-in `nlpsol` object, B = 1): the unsubstituted free-end-time Dubins problem -- whose
    terminal row has a vanishing gradient at the reference's guess -- ends in Infeasible_Problem_Detected with the exact Hessian and
    in Solve_Succeeded at the second attempt; switched off, the first verdict stands.  (Host twin of the object: tests/port_solver.py.)"""
    import port_solver
    tpl, d = _load('dubins_freeT')
    x, lb, ub = _extended(tpl, d)
    options = {'solver': 'ipopt', 'solver_options': {'ipopt': {'ipopt.tol': 1e-3}}}
    solver, _ = port_solver.create_nlp(tpl, options)
    res = solver(x0=x, p=d['p0'], lbg=lb, ubg=ub)
    st = solver.stats()
    assert st['return_status'] == 'Solve_Succeeded' and 1000 < st['iter_count'] < 2500, st
    _check_solution('dubins_freeT', tpl, dict(d, slsqp_ok=0), {'x': res['x'][None], 'lam_g': res['lam_g'][None], 'status': np.zeros(1, int)}, 1e-3)
    off, _ = port_solver.create_nlp(tpl, dict(options, omgx={'hess_fallback': False}))
    off(x0=x, p=d['p0'], lbg=lb, ubg=ub)
    assert off.stats() == {'return_status': 'Infeasible_Problem_Detected', 'iter_count': 40}


@pytest.mark.gpu
def test_the_second_attempt_on_the_device():
    import omgtools.backend as be
    tpl, d = _load('dubins_freeT')
    x, lb, ub = _extended(tpl, d)
    solver = be.NlpSolver(tpl, {'solver': 'ipopt', 'solver_options': {'ipopt': {'ipopt.tol': 1e-3}}})
    try:
        res = solver(x0=x, p=d['p0'], lbg=lb, ubg=ub)
        st = solver.stats()
        assert st['return_status'] == 'Solve_Succeeded' and 1000 < st['iter_count'] < 2500, st
        _check_solution('dubins_freeT', tpl, dict(d, slsqp_ok=0), {'x': np.asarray(res['x'])[None], 'lam_g': np.asarray(res['lam_g'])[None],
                                                                   'status': np.zeros(1, int)}, 1e-3)
        # the handle is back on the exact Hessian afterwards: the same call gives the same answer by the same route
        again = solver(x0=x, p=d['p0'], lbg=lb, ubg=ub)
        assert solver.stats() == st and np.array_equal(np.asarray(again['x']), np.asarray(res['x']))
    finally:
        solver.batch.close()
