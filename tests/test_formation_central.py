"""K10 of SURVEY.md 8c, second half: the fixed point of the formation ADMM against the CENTRALISED formation problem
(`problems/formation_central.py:36-81`), compared the way the reference compares them
(`examples/compare_distributed_optimization_quadrotors.py:52-64, 106-115`): the stacked fleet-centre coefficients of the
ADMM iterate against those of the central solution, ||x_admm - x_central|| / ||x_central||.

The central problem here: all vehicles in one NLP -- every vehicle's own objective and rows (the x-update template with
rho = 0 and zero multipliers is the plain point-to-point problem of that vehicle), and `centre_i - centre_j = 0`
coefficient-wise for the couples of the chain (`formation_central.py:52-60, 79`) -- solved by scipy SLSQP from the
straight-line guess.  (The reference's central problem lets the vehicles share the terminal slacks g0, g1 -- its
expressions are matched by symbol name -- while every ADMM agent has its own: under the formation constraint the
slacks coincide, the objectives differ by the factor N, the minimiser is the same.  The equality rows of vehicles 1..
are implied by vehicle 0's and the centre equalities and are left out: SLSQP needs independent rows.)

Four Holonomic vehicles in a square formation pass a disc that reaches into their straight path: the whole fleet has to
swerve, which no vehicle learns from its own problem alone -- only the consensus can produce the central solution."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
N = 4


def _scenario():
    import omgtools.backend as be
    from omgtools import scenarios
    from omgtools.shapes import Circle
    saved = be.create_nlp
    be.create_nlp = lambda tpl, opt, name='': (None, 0.)
    try:
        return scenarios.formation_holonomic(N, obstacles=[((0.5, -0.5), (0., 0.), Circle(0.4))])
    finally:
        be.create_nlp = saved


def _centres(x, P, lay):
    return np.stack([x[i, lay.x_spl:lay.x_spl + lay.ns] + np.repeat(P['p'][i, lay.p_rel:lay.p_rel + lay.n_dim], lay.L)
                     for i in range(N)])


@pytest.fixture(scope='module')
def central():
    from scipy.optimize import minimize
    from oracle.nlp_numpy import NumpyNLP
    problem, updater, father, lay, P = _scenario()
    tpl = father.template
    nlp = NumpyNLP(tpl)
    nv, ns, L = tpl.n_var, lay.ns, lay.L
    p0 = np.array(P['p'])
    p0[:, lay.p_rho] = 0.0                                    # no ADMM terms: the vehicle's own point-to-point problem
    for off, size in ((lay.p_zi, ns), (lay.p_li, ns), (lay.p_zji, lay.n_nghb * ns), (lay.p_lji, lay.n_nghb * ns)):
        p0[:, off:off + size] = 0.0
    cs = [nlp.term_coefs(p0[i]) for i in range(N)]
    eq = tpl.lb == tpl.ub
    ineq = np.isfinite(tpl.ub) & ~eq
    split = lambda v: v.reshape(N, nv)
    A, b = np.zeros((ns * (N - 1), N * nv)), np.zeros(ns * (N - 1))
    for i in range(N - 1):                                    # couples of the chain 0-1, 1-2, 2-3
        for k in range(ns):
            A[i * ns + k, i * nv + lay.x_spl + k], A[i * ns + k, (i + 1) * nv + lay.x_spl + k] = 1.0, -1.0
        b[i * ns:(i + 1) * ns] = np.repeat(P['p'][i + 1, lay.p_rel:lay.p_rel + 2] - P['p'][i, lay.p_rel:lay.p_rel + 2], L)

    def blockdiag(rows):
        out, a = np.zeros((sum(r.shape[0] for r in rows), N * nv)), 0
        for i, r in enumerate(rows):
            out[a:a + r.shape[0], i * nv:(i + 1) * nv] = r
            a += r.shape[0]
        return out

    def eq_jac(v):
        J0 = np.zeros((int(eq.sum()), N * nv))
        J0[:, :nv] = nlp.jac(split(v)[0], cs[0])[:-1][eq]
        return np.vstack([J0, A])
    cons = [{'type': 'eq', 'fun': lambda v: np.r_[nlp.fg(split(v)[0], cs[0])[1][eq] - tpl.lb[eq], A @ v - b], 'jac': eq_jac},
            {'type': 'ineq', 'fun': lambda v: np.concatenate([(tpl.ub - nlp.fg(x, c)[1])[ineq] for x, c in zip(split(v), cs)]),
             'jac': lambda v: -blockdiag([nlp.jac(x, c)[:-1][ineq] for x, c in zip(split(v), cs)])}]
    out = minimize(lambda v: sum(nlp.fg(x, c)[0] for x, c in zip(split(v), cs)), P['x0'].reshape(-1),
                   jac=lambda v: np.concatenate([nlp.jac(x, c)[-1] for x, c in zip(split(v), cs)]), constraints=cons,
                   method='SLSQP', options={'maxiter': 500, 'ftol': 1e-12})
    assert out.status == 0, out.message
    cen = _centres(split(out.x), P, lay)
    assert np.abs(cen - cen[0]).max() < 1e-12
    # the disc matters: the central plan leaves the straight line by centimetres
    straight = _centres(P['x0'], P, lay)
    assert np.abs(cen - straight).max() > 0.05
    return tpl, lay, P, cen


def _run(admm, ops_x, lay, P, cen, iters):
    errs = []
    admm.initialize()
    for it in range(iters):
        admm.iterate(0.0)
        errs.append(np.linalg.norm(_centres(ops_x(), P, lay) - cen) / np.linalg.norm(cen))
    return errs


def test_admm_converges_to_the_central_solution_on_the_host(central):
    from omgtools.admm import BatchADMM
    from admm_numpy_ops import NumpyAdmmOps
    tpl, lay, P, cen = central
    ops = NumpyAdmmOps(tpl, lay, P['p'], P['x0'], tol=1e-8)
    errs = _run(BatchADMM(lay, P['nbr'], ops, rho=1.0), lambda: ops.x, lay, P, cen, 60)
    assert errs[0] > 1e-2                 # the first x-updates know nothing of each other
    assert errs[9] < 1e-2 and errs[-1] < 1e-4, (errs[9], errs[-1])


@pytest.mark.gpu
def test_admm_converges_to_the_central_solution_on_hip(central):
    import torch
    from omgtools.admm import BatchADMM, HipAdmmOps
    from omgtools.backend import BatchSolver
    tpl, lay, P, cen = central
    solver = BatchSolver(tpl, N, options=dict(tol=1e-8, max_iter=300))
    try:
        ops = HipAdmmOps(solver, tpl, lay, P['p'], P['x0'], torch.device('cuda', 0))
        errs = _run(BatchADMM(lay, P['nbr'], ops, rho=1.0), lambda: ops.x.cpu().numpy(), lay, P, cen, 60)
    finally:
        solver.close()
    assert errs[9] < 1e-2 and errs[-1] < 1e-4, (errs[9], errs[-1])
