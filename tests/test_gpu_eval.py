"""K9 of SURVEY.md 8c: the derivatives the solve kernel works with -- Jacobian items, row terms, the Hessian items of its
assembly pass, evaluated on the device by `omgx_batch_eval` from the kernel's own tables -- against the oracle's numpy
restatement of the reference's NLP (oracle/nlp_numpy.py, pinned to the reference's construct code by tests/golden) and
against central finite differences of g, at seeded random points: constraint values, objective, Jacobian, Lagrangian
Hessian.  One class per workspace mode: config 2 (compact store, two agents per CU), the Quadrotor class (cubic rows,
KKT store in an HBM slab), the 3-D class (row arrays in the slab as well)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _build(name, n):
    import omgtools.backend as be
    from omgtools import scenarios
    saved = be.create_nlp
    be.create_nlp = lambda tpl, opt, name='': (None, 0.)
    try:
        return getattr(scenarios, name)(n)
    finally:
        be.create_nlp = saved


@pytest.mark.parametrize('scenario', ['holonomic_p2p', 'quadrotor_p2p', 'holonomic3d_p2p'])
def test_device_derivatives_match_the_oracle(scenario):
    from omgtools.backend import BatchSolver
    from oracle.nlp_numpy import NumpyNLP
    B = 3
    problem, P = _build(scenario, B)
    tpl = problem.father.template
    nlp = NumpyNLP(tpl)
    rng = np.random.default_rng(5)
    x = P['x0'] + rng.normal(scale=0.3, size=P['x0'].shape)
    p = np.array(P['p'])
    o_t = tpl.entry_range(problem.label, 't', 'par')[0]
    p[:, o_t] = [0.0, 0.013, 0.31][:B]                       # time since the last knot: inside the first interval
    lam = rng.normal(size=(B, tpl.n_con))
    solver = BatchSolver(tpl, B)
    try:
        got = solver.eval(p, x, lam)
    finally:
        solver.close()
    for b in range(B):
        c = nlp.term_coefs(p[b])
        f, g = nlp.fg(x[b], c)
        J = nlp.jac(x[b], c)
        H = nlp.hess(x[b], lam[b], c)
        sg, sj, sh = max(1.0, np.abs(g).max()), max(1.0, np.abs(J).max()), max(1.0, np.abs(H).max())
        assert np.abs(got['g'][b] - g).max() < 1e-10 * sg, (scenario, b)
        assert abs(got['f'][b] - f) < 1e-10 * max(1.0, abs(f))
        assert np.abs(got['jac'][b] - J).max() < 1e-10 * sj, (scenario, b, np.abs(got['jac'][b] - J).max())
        assert np.abs(got['hess'][b] - H).max() < 1e-10 * sh, (scenario, b, np.abs(got['hess'][b] - H).max())
        assert np.array_equal(got['hess'][b], got['hess'][b].T)
    # finite differences of the device's own g: the Jacobian is the derivative of what the kernel evaluates
    b, h = 0, 1e-6
    cols = rng.choice(tpl.n_var, size=min(B - 1, 2), replace=False)
    xs = np.repeat(x[b:b + 1], B, axis=0)
    ps, ls = np.repeat(p[b:b + 1], B, axis=0), np.repeat(lam[b:b + 1], B, axis=0)
    solver = BatchSolver(tpl, B)
    try:
        for j in cols:
            xp, xm = xs.copy(), xs.copy()
            xp[:, j] += h; xm[:, j] -= h
            gp, gm = solver.eval(ps, xp, ls), solver.eval(ps, xm, ls)
            fd = (gp['g'][0] - gm['g'][0]) / (2 * h)
            noise_g = 64 * np.finfo(float).eps * np.abs(gp['g'][0]).max() / h
            assert np.abs(fd - got['jac'][b][:-1, j]).max() < 1e-6 * max(1.0, np.abs(fd).max()) + noise_g, (scenario, j)
            # and the Hessian column as the derivative of the Lagrangian gradient
            gl_p = gp['jac'][0][-1] + lam[b] @ gp['jac'][0][:-1]
            gl_m = gm['jac'][0][-1] + lam[b] @ gm['jac'][0][:-1]
            fdh = (gl_p - gl_m) / (2 * h)
            # (rounding of the difference quotient: the Quadrotor's thrust rows have Jacobian entries of 1e7)
            noise = 64 * np.finfo(float).eps * np.abs(gl_p).max() / h
            assert np.abs(fdh - got['hess'][b][:, j]).max() < 1e-5 * max(1.0, np.abs(fdh).max()) + noise, (scenario, j)
    finally:
        solver.close()
