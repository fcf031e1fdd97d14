"""Two-sided rows lb < g < ub (`basics/optilayer.py:634-666`) on the checker side (CPU tier: oracle port + oracle/range_rows.py,
the restatement of what the library does around its kernel; GPU tier: tests/test_gpu_range_rows.py)."""
import numpy as np

from test_gpu_range_rows import _merged, TOL


def _case(B=4):
    from omgtools import workloads
    from oracle.nlp_numpy import NumpyNLP
    problem, P = workloads.holonomic_p2p(B)
    tpl = problem.father.template
    nlp = NumpyNLP(tpl)
    t2, pairs = _merged(tpl, nlp, P['p'][0], np.random.default_rng(0).normal(size=tpl.n_var))
    return tpl, t2, pairs, nlp, P


def test_merged_velocity_rows_are_the_same_problem():
    from oracle import port_binding
    tpl, t2, pairs, nlp, P = _case()
    ra = port_binding.solve(tpl, P['p'], P['x0'], tol=TOL, max_iter=500)
    rb = port_binding.solve(t2, P['p'], P['x0'], tol=TOL, max_iter=500)
    assert (ra['status'] == 0).all() and (rb['status'] == 0).all()
    assert np.abs(ra['x'] - rb['x']).max() < 1e-8
    for up, lo in pairs:
        assert np.abs(rb['lam_g'][:, up] - (ra['lam_g'][:, up] - ra['lam_g'][:, lo])).max() < 1e-8
        assert np.abs(rb['lam_g'][:, lo]).max() == 0.0


def test_asymmetric_two_sided_rows_against_slsqp():
    """A problem only two-sided rows state: the x-acceleration limited to [-0.35, 1] m/s^2 (rows -150 T^2 ... : the lower side is
    NOT one of the reference's own rows).  scipy SLSQP takes both sides natively; the interior-point solve -- every such row
    doubled -- must return its solution, with multipliers of either sign on the two-sided rows."""
    from oracle import port_binding
    from oracle.kkt_check import assert_kkt
    from slsqp_reference import solve_slsqp
    tpl, t2, pairs, nlp, P = _case(3)
    import copy
    t3 = copy.copy(tpl)
    t3.lb, t3.ub = tpl.lb.copy(), tpl.ub.copy()
    (up, r, _), = [v for (lab, nm), v in tpl.con_layout.items() if lab.startswith('vehicle') and nm.startswith('c_6_')]    # ddx - T^2 axmax <= 0
    T = 10.0
    t3.lb[up:up + r] = -(1.0 + 0.35) * T ** 2                  # ddx >= -0.35 T^2
    res = port_binding.solve(t3, P['p'], P['x0'], tol=TOL, max_iter=500)
    assert (res['status'] == 0).all()
    lo, hi = tpl.entry_range([lab for (lab, nm) in tpl.var_layout if nm == 'splines_seg0'][0], 'splines_seg0', 'var')
    signs = set()
    for b in range(3):
        # (non-convex: from the straight-line guess the two solvers may pick different sides of an obstacle -- the independent
        # solver is started at the returned point and has to confirm it as a minimum of the two-sided problem)
        xs, fs, ok = solve_slsqp(nlp, t3, res['x'][b], P['p'][b], maxiter=800)
        assert ok
        f = nlp.fg(res['x'][b], nlp.term_coefs(P['p'][b]))[0]
        assert abs(f - fs) < 1e-5 * (1 + abs(f)), (b, f, fs)
        assert np.abs(res['x'][b, lo:hi] - xs[lo:hi]).max() < 1e-3
        g = nlp.fg(res['x'][b], nlp.term_coefs(P['p'][b]))[1]
        assert (g[up:up + r] <= 1e-6).all() and (g[up:up + r] >= t3.lb[up:up + r] - 1e-6).all()
        lam = res['lam_g'][b, up:up + r]
        signs |= set(np.sign(lam[np.abs(lam) > 1e-4]).astype(int).tolist())
        # stationarity with the caller's multipliers: grad f + J' lam = 0
        J = nlp.jac(res['x'][b], nlp.term_coefs(P['p'][b]))
        assert np.abs(J[-1] + J[:-1].T @ res['lam_g'][b]).max() < 1e-4
    assert signs == {-1, 1}, signs                              # both sides are active somewhere (braking and accelerating)
