"""Terms of degree 4 in the variables: a free end time with the 2-norm limits of
`vehicles/holonomic.py:54-58, 67-72` -- ddx^2 + ddy^2 <= (T^2 a_max)^2 -- as in the reference's
`examples/p2p_holonomic_balls.py` (free T, moving circles).  The NLP itself is pinned to the reference's construct code
by tests/test_golden_nlp.py (`freeT_balls_norm2`); here:

  CPU  the oracle's Jacobian and Hessian statements against central differences of its own g (the general-degree
       restatement in oracle/nlp_numpy.py), the host build of the kernel source against scipy SLSQP from the reference's
       initial guess, and iterate for iterate against the dense numpy solver;
  GPU  the kernel's derivative tables (`omgx_batch_eval`) against the oracle, the HIP solve against SLSQP and the host
       build, a template-file round trip of the wider term records."""
import numpy as np
import pytest


@pytest.fixture(scope='module')
def balls():
    import omgtools.backend as be
    from test_golden_nlp import build
    saved = be.create_nlp
    be.create_nlp = lambda tpl, opt, name='': (None, 0.)
    try:
        pr = build('freeT_balls_norm2')
        pr.reinitialize()
    finally:
        be.create_nlp = saved
    fa, tpl = pr.father, pr.father.template
    x0 = np.asarray(fa.get_variables()).reshape(-1).copy()
    p = fa.set_parameters(0.).cat.copy()
    sl = slice(*tpl.entry_range(pr.vehicles[0].label, 'splines_seg0', 'var'))
    assert tpl.t_var.shape[1] == 4 and tpl.t_nv.max() == 4 and (tpl.t_nv == 4).sum() > 0
    return pr, tpl, p, x0, sl


def test_oracle_derivatives_of_quartic_terms(balls):
    from oracle.nlp_numpy import NumpyNLP
    pr, tpl, p, x0, sl = balls
    nlp = NumpyNLP(tpl)
    rng = np.random.default_rng(11)
    x = x0 + rng.normal(scale=0.3, size=x0.shape)
    lam = rng.normal(size=tpl.n_con)
    c = nlp.term_coefs(p)
    J, H = nlp.jac(x, c), nlp.hess(x, lam, c)
    assert np.array_equal(H, H.T)
    quartic = np.unique(tpl.t_var[tpl.t_nv == 4])
    h = 1e-6
    for j in list(quartic[:3]) + list(rng.choice(tpl.n_var, size=3, replace=False)):
        e = np.zeros_like(x); e[j] = h
        (fp, gp), (fm, gm) = nlp.fg(x + e, c), nlp.fg(x - e, c)
        fd = np.r_[gp - gm, fp - fm] / (2 * h)
        assert np.abs(fd - J[:, j]).max() < 1e-6 * max(1.0, np.abs(fd).max()), j
        Jp, Jm = nlp.jac(x + e, c), nlp.jac(x - e, c)
        fdh = ((Jp[-1] + lam @ Jp[:-1]) - (Jm[-1] + lam @ Jm[:-1])) / (2 * h)
        assert np.abs(fdh - H[:, j]).max() < 1e-5 * max(1.0, np.abs(fdh).max()), j
    # Gershgorin sums of the inertia correction: H + diag(G) is diagonally dominant term by term
    G = nlp.hess_gershgorin(x, lam, c)
    assert (G >= np.abs(H).sum(axis=1) - np.abs(np.diag(H)) - np.maximum(np.diag(H), 0.0) - 1e-9).all()


def test_port_reaches_the_slsqp_minimum_and_follows_the_numpy_solver(balls):
    from oracle import port_binding, ipm_numpy
    from oracle.nlp_numpy import NumpyNLP
    from oracle.kkt_check import assert_kkt
    from slsqp_reference import solve_slsqp
    pr, tpl, p, x0, sl = balls
    nlp = NumpyNLP(tpl)
    xs, fs, ok = solve_slsqp(nlp, tpl, x0, p)
    assert ok
    res = port_binding.solve(tpl, p[None], x0[None], tol=1e-6, max_iter=500)
    assert res['status'][0] == 0 and res['iters'][0] < 120
    f = nlp.fg(res['x'][0], nlp.term_coefs(p))[0]
    assert abs(f - fs) < 1e-5 * (1 + abs(f))                    # the objective is the motion time T
    assert np.abs(res['x'][0][sl] - xs[sl]).max() < 1e-4
    assert_kkt(nlp, tpl, p, res['x'][0], res['lam_g'][0], 1e-5, 'balls')
    # the same iterates as the dense numpy restatement of the solver
    for iters in (3, 12):
        a = port_binding.solve(tpl, p[None], x0[None], tol=1e-12, max_iter=iters)
        b = ipm_numpy.solve(nlp, x0, p, tpl.lb, tpl.ub, opts={'tol': 1e-12, 'max_iter': iters})
        assert b['iters'] == a['iters'][0] == iters
        assert np.abs(a['x'][0] - b['x']).max() < 1e-8 * max(1.0, np.abs(b['x']).max()), iters


@pytest.mark.gpu
def test_device_tables_of_quartic_terms(balls, tmp_path):
    import omgtools.backend as be
    from oracle.nlp_numpy import NumpyNLP
    pr, tpl, p, x0, sl = balls
    nlp = NumpyNLP(tpl)
    rng = np.random.default_rng(12)
    B = 3
    x = x0[None] + rng.normal(scale=0.3, size=(B, tpl.n_var))
    ps = np.repeat(p[None], B, axis=0)
    lam = rng.normal(size=(B, tpl.n_con))
    solver = be.BatchSolver(tpl, B)
    try:
        got = solver.eval(ps, x, lam)
    finally:
        solver.close()
    c = nlp.term_coefs(p)
    for b in range(B):
        f, g = nlp.fg(x[b], c)
        J, H = nlp.jac(x[b], c), nlp.hess(x[b], lam[b], c)
        assert np.abs(got['g'][b] - g).max() < 1e-10 * max(1.0, np.abs(g).max())
        assert abs(got['f'][b] - f) < 1e-10 * max(1.0, abs(f))
        assert np.abs(got['jac'][b] - J).max() < 1e-10 * max(1.0, np.abs(J).max())
        assert np.abs(got['hess'][b] - H).max() < 1e-10 * max(1.0, np.abs(H).max())
    # the template file carries four variables per term
    path = be.save_template(tpl, str(tmp_path / 'balls.omgx'))
    assert open(path, 'rb').read(8) == b'OMGXTPL4'


@pytest.mark.gpu
def test_hip_solves_the_quartic_problem_like_slsqp_and_the_port(balls):
    import omgtools.backend as be
    from oracle import port_binding
    from oracle.nlp_numpy import NumpyNLP
    from oracle.kkt_check import assert_kkt
    from slsqp_reference import solve_slsqp
    pr, tpl, p, x0, sl = balls
    nlp = NumpyNLP(tpl)
    xs, fs, ok = solve_slsqp(nlp, tpl, x0, p)
    assert ok
    # a batch: the reference's start and goal, and three more goals
    B = 4
    ps = np.repeat(p[None], B, axis=0)
    o_T = tpl.entry_range(pr.vehicles[0].label, 'poseT', 'par')[0]
    ps[1:, o_T + 1] += [0.6, -0.8, 1.5]
    xb = np.repeat(x0[None], B, axis=0)
    solver = be.BatchSolver(tpl, B, options=dict(tol=1e-6, max_iter=500))
    try:
        res = solver.solve(ps, xb)
        again = solver.solve(ps, xb)
    finally:
        solver.close()
    assert (res['status'] == 0).all(), res['status']
    assert np.array_equal(res['x'], again['x'])                             # bit-identical between runs
    f = nlp.fg(res['x'][0], nlp.term_coefs(p))[0]
    assert abs(f - fs) < 1e-5 * (1 + abs(f))
    assert np.abs(res['x'][0][sl] - xs[sl]).max() < 1e-4
    for b in range(B):
        assert_kkt(nlp, tpl, ps[b], res['x'][b], res['lam_g'][b], 1e-5, ('balls', b))
    port = port_binding.solve(tpl, ps, xb, tol=1e-6, max_iter=500)
    assert np.array_equal(port['status'], res['status'])
    assert np.abs(port['x'][:, sl] - res['x'][:, sl]).max() < 1e-4
    # iterate for iterate against the host build of the same source
    for iters in (3, 12):
        s2 = be.BatchSolver(tpl, B, options=dict(tol=1e-12, max_iter=iters))
        try:
            a = s2.solve(ps, xb)
        finally:
            s2.close()
        b_ = port_binding.solve(tpl, ps, xb, tol=1e-12, max_iter=iters)
        assert np.abs(a['x'] - b_['x']).max() < 1e-8 * max(1.0, np.abs(b_['x']).max()), iters


# ---- the reference's own Dubins class (tangent-half-angle model) --------------------------------------------------------
# tests/golden/dubins_fixedT.npz: the template `omgx_shim` derives from the reference's unmodified modules
# (tests/golden/generate_shim_fixtures.py; the generator's full Simulator run reaches the target pose to 7e-3 in 119 updates on
# the host build); 46,912 terms, 13,568 of them with four factors.
@pytest.fixture(scope='module')
def dubins():
    import os
    from omgtools.template import NLPTemplate
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'dubins_fixedT.npz')
    tpl = NLPTemplate.from_npz(path)
    d = np.load(path)
    assert (tpl.n_var, tpl.n_con, tpl.n_par) == (105, 456, 19) and (tpl.t_nv == 4).sum() == 13568
    return tpl, d


def test_dubins_template_reproduces_the_reference_graphs_and_solves(dubins):
    from oracle import port_binding
    from oracle.nlp_numpy import NumpyNLP
    from oracle.kkt_check import assert_kkt
    tpl, d = dubins
    nlp = NumpyNLP(tpl)
    for xv, pv, fs, gs in zip(d['xs'], d['ps'], d['fs'], d['gs']):          # values of the reference's f / g closures
        f, g = nlp.fg(xv, nlp.term_coefs(pv))
        assert abs(f - fs) < 1e-12 * (1 + abs(fs)) and np.abs(g - gs).max() < 1e-12 * (1 + np.abs(gs).max())
    res = port_binding.solve(tpl, d['p0'][None], d['x0'][None], tol=1e-6, max_iter=500)
    assert res['status'][0] == 0
    f = nlp.fg(res['x'][0], nlp.term_coefs(d['p0']))[0]
    assert abs(f - float(d['f_slsqp'])) < 1e-5 * (1 + abs(f))
    off, end = [v for (lab, name), v in ((k, tpl.entry_range(k[0], k[1], 'var')) for k in tpl.var_layout)
                if name.startswith('splines_seg')][0]
    assert np.abs(res['x'][0][off:end] - d['x_slsqp'][off:end]).max() < 1e-4
    assert_kkt(nlp, tpl, d['p0'], res['x'][0], res['lam_g'][0], 1e-5, 'dubins')


@pytest.mark.gpu
def test_dubins_on_the_device(dubins):
    import omgtools.backend as be
    from oracle import port_binding
    from oracle.nlp_numpy import NumpyNLP
    from oracle.kkt_check import assert_kkt
    tpl, d = dubins
    nlp = NumpyNLP(tpl)
    rng = np.random.default_rng(13)
    B = 2
    ps = np.repeat(d['p0'][None], B, axis=0)
    x = d['x0'][None] + rng.normal(scale=0.2, size=(B, tpl.n_var))
    lam = rng.normal(size=(B, tpl.n_con))
    solver = be.BatchSolver(tpl, B, options=dict(tol=1e-6, max_iter=500))
    try:
        got = solver.eval(ps, x, lam)
        res = solver.solve(ps, np.repeat(d['x0'][None], B, axis=0))
    finally:
        solver.close()
    c = nlp.term_coefs(d['p0'])
    for b in range(B):
        f, g = nlp.fg(x[b], c)
        J, H = nlp.jac(x[b], c), nlp.hess(x[b], lam[b], c)
        assert np.abs(got['g'][b] - g).max() < 1e-10 * max(1.0, np.abs(g).max())
        assert np.abs(got['jac'][b] - J).max() < 1e-10 * max(1.0, np.abs(J).max())
        assert np.abs(got['hess'][b] - H).max() < 1e-10 * max(1.0, np.abs(H).max())
    assert (res['status'] == 0).all() and np.array_equal(res['x'][0], res['x'][1])
    f = nlp.fg(res['x'][0], c)[0]
    assert abs(f - float(d['f_slsqp'])) < 1e-5 * (1 + abs(f))
    assert_kkt(nlp, tpl, d['p0'], res['x'][0], res['lam_g'][0], 1e-5, 'dubins')
    port = port_binding.solve(tpl, d['p0'][None], d['x0'][None], tol=1e-6, max_iter=500)
    assert abs(int(port['iters'][0]) - int(res['iters'][0])) <= 6
    assert np.abs(port['x'][0] - res['x'][0]).max() < 1e-4


# ---- the same class with the substituted velocity splines: two-sided rows (round 4) -------------------------------------
# tests/golden/dubins_subst.npz: `options['substitution']` of the reference's Dubins class (`vehicles/dubins.py:92-115`): the
# position is the integral of separate velocity splines, tied to the tangent-half-angle expressions by 118 rows
# -1e-3 <= x - int(v_til (1 - tg_ha^2)) <= 1e-3 (`basics/optilayer.py:634-666`).  The library solves such a row doubled (ABI 5).  A
# tube of quartic rows 2e-3 wide is hard ground for an interior-point iteration: 717 iterations from the reference's guess at
# tol 1e-3 (at 1e-6 the stall test of phase I gives up at iteration 60: step lengths of 1e-3 inside the tube) -- the fixture
# pins that the class builds, reproduces the reference's graphs and is solved to the optimum SLSQP finds, not that it is fast.
# (Written before templates off the wave path had the second-order correction: with it 149 iterations, and 159 at 1e-6.)
@pytest.fixture(scope='module')
def dubins_subst():
    import os
    from omgtools.template import NLPTemplate
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'dubins_subst.npz')
    tpl = NLPTemplate.from_npz(path)
    d = np.load(path)
    two_sided = np.isfinite(tpl.lb) & np.isfinite(tpl.ub) & (tpl.lb < tpl.ub)
    assert (tpl.n_var, tpl.n_con, tpl.n_par) == (96, 414, 19) and two_sided.sum() == 118 and int(d['slsqp_ok']) == 1
    return tpl, d


def _check_subst(tpl, d, res):
    from oracle.nlp_numpy import NumpyNLP
    from oracle.kkt_check import assert_kkt
    nlp = NumpyNLP(tpl)
    assert res['status'][0] == 0 and res['lam_g'].shape[1] == tpl.n_con
    f, g = nlp.fg(res['x'][0], nlp.term_coefs(d['p0']))
    assert (g - tpl.ub).max() < 1e-6 and (tpl.lb - g).max() < 1e-6                 # inside every tube
    assert abs(f - float(d['f_slsqp'])) < 1e-2 * (1 + abs(f))                      # (tol 1e-3: the barrier's share of the objective)
    assert_kkt(nlp, tpl, d['p0'], res['x'][0], res['lam_g'][0], 1e-2, 'dubins_subst')


def test_substituted_dubins_reproduces_the_reference_graphs_and_solves(dubins_subst):
    from oracle import port_binding
    from oracle.nlp_numpy import NumpyNLP
    tpl, d = dubins_subst
    nlp = NumpyNLP(tpl)
    for xv, pv, fs, gs in zip(d['xs'], d['ps'], d['fs'], d['gs']):          # values of the reference's f / g closures
        f, g = nlp.fg(xv, nlp.term_coefs(pv))
        assert abs(f - fs) < 1e-12 * (1 + abs(fs)) and np.abs(g - gs).max() < 1e-12 * (1 + np.abs(gs).max())
    _check_subst(tpl, d, port_binding.solve(tpl, d['p0'][None], d['x0'][None], tol=1e-3, max_iter=3000))


@pytest.mark.gpu
def test_substituted_dubins_on_the_device(dubins_subst):
    import omgtools.backend as be
    from oracle import port_binding
    tpl, d = dubins_subst
    solver = be.BatchSolver(tpl, 2, options=dict(tol=1e-3, max_iter=3000))
    try:
        res = solver.solve(np.repeat(d['p0'][None], 2, axis=0), np.repeat(d['x0'][None], 2, axis=0), lbg=tpl.lb, ubg=tpl.ub)
    finally:
        solver.close()
    assert np.array_equal(res['x'][0], res['x'][1])
    _check_subst(tpl, d, res)
    port = port_binding.solve(tpl, d['p0'][None], d['x0'][None], tol=1e-3, max_iter=3000)
    assert abs(int(port['iters'][0]) - int(res['iters'][0])) <= 40 and np.abs(port['x'][0] - res['x'][0]).max() < 1e-2


# ---- rotating obstacles (`environment/obstacle.py:299-332`, `examples/revolving_door.py`) --------------------------------
# tests/golden/revolving_door.npz: the template of the reference's own example on `omgx_shim`; the orientation of the two
# rotating beams enters through cos / sin of (theta - t omega): COS / SIN atoms of the parameter program.
@pytest.fixture(scope='module')
def door():
    import os
    from omgtools.template import NLPTemplate
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'revolving_door.npz')
    tpl = NLPTemplate.from_npz(path)
    assert (tpl.prog[:, 0] >= 2).sum() >= 4                   # cos and sin of two beams
    return tpl, np.load(path)


def test_rotating_obstacle_template_reproduces_the_reference_graphs_and_solves(door):
    from oracle import port_binding
    from oracle.nlp_numpy import NumpyNLP
    from oracle.kkt_check import assert_kkt
    tpl, d = door
    nlp = NumpyNLP(tpl)
    assert len(set(np.round(d['ps'][:, [off for (lab, name), (off, r, c) in tpl.par_layout.items() if name == 'theta'][0]], 6))) == 3
    for xv, pv, fs, gs in zip(d['xs'], d['ps'], d['fs'], d['gs']):          # the reference's graphs at three orientations
        f, g = nlp.fg(xv, nlp.term_coefs(pv))
        assert abs(f - fs) < 1e-12 * (1 + abs(fs)) and np.abs(g - gs).max() < 1e-12 * (1 + np.abs(gs).max())
    res = port_binding.solve(tpl, d['p0'][None], d['x0'][None], tol=1e-6, max_iter=500)
    assert res['status'][0] == 0
    assert_kkt(nlp, tpl, d['p0'], res['x'][0], res['lam_g'][0], 1e-5, 'door')


@pytest.mark.gpu
def test_rotating_obstacles_on_the_device(door):
    import omgtools.backend as be
    from oracle import port_binding
    from oracle.nlp_numpy import NumpyNLP
    from oracle.kkt_check import assert_kkt
    tpl, d = door
    nlp = NumpyNLP(tpl)
    B = 3
    solver = be.BatchSolver(tpl, B, options=dict(tol=1e-6, max_iter=500))
    try:
        got = solver.eval(d['ps'], d['xs'], np.zeros((B, tpl.n_con)))        # three orientations of the beams
        ps = np.repeat(d['p0'][None], B, axis=0)
        res = solver.solve(ps, np.repeat(d['x0'][None], B, axis=0))
    finally:
        solver.close()
    for b in range(B):
        assert np.abs(got['g'][b] - d['gs'][b]).max() < 1e-10 * (1 + np.abs(d['gs'][b]).max())
        J = nlp.jac(d['xs'][b], nlp.term_coefs(d['ps'][b]))
        assert np.abs(got['jac'][b] - J).max() < 1e-10 * max(1.0, np.abs(J).max())
    assert (res['status'] == 0).all()
    assert_kkt(nlp, tpl, d['p0'], res['x'][0], res['lam_g'][0], 1e-5, 'door')
    port = port_binding.solve(tpl, d['p0'][None], d['x0'][None], tol=1e-6, max_iter=500)
    f_h, f_p = nlp.fg(res['x'][0], nlp.term_coefs(d['p0']))[0], nlp.fg(port['x'][0], nlp.term_coefs(d['p0']))[0]
    assert abs(f_h - f_p) < 1e-5 * (1 + abs(f_p)) and abs(int(port['iters'][0]) - int(res['iters'][0])) <= 6
