"""Known-answer tests of the casadi stand-in (omg-tools_amd/omgx_shim/casadi) against numpy: every golden NLP of this repository
(tests/golden/nlp_*.npz, admm_*.npz, dubins_*.npz) is the reference's own construct code evaluated ON this stand-in, so its
operations -- products, concatenation, indexing, substitution, functions, the linear solve of the ADMM z-update
(`problems/admm.py:154`) -- must be the operations CasADi documents.  Numbers in, numbers out; no /root/reference needed."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def ca():
    shim = os.path.join(ROOT, 'omg-tools_amd', 'omgx_shim')
    sys.path.insert(0, os.path.join(ROOT, 'omg-tools_amd'))
    if shim not in sys.path:
        sys.path.insert(0, shim)
    saved = sys.modules.pop('casadi', None)
    import casadi
    assert casadi.__file__.startswith(shim)
    yield casadi
    if saved is not None:
        sys.modules['casadi'] = saved


def val(ca, expr, env):
    return np.asarray(ca.MX.lift(expr).eval(env), dtype=float)


def test_arithmetic_concatenation_and_indexing(ca):
    rng = np.random.default_rng(5)
    x, y = ca.MX.sym('x', 4), ca.MX.sym('y', 4)
    A = rng.normal(size=(3, 4))
    xv, yv = rng.normal(size=(4, 1)), rng.normal(size=(4, 1))
    env = {x: xv, y: yv}
    assert np.allclose(val(ca, 2 * x - y / 3 + x * y, env), 2 * xv - yv / 3 + xv * yv, atol=0, rtol=1e-15)
    assert np.allclose(val(ca, x**2 - (-y), env), xv**2 + yv, rtol=1e-15)
    assert np.allclose(val(ca, ca.mtimes(A, x), env), A @ xv, rtol=1e-14)
    assert np.allclose(val(ca, ca.mtimes(x.T, y), env), xv.T @ yv, rtol=1e-14)
    assert np.allclose(val(ca, ca.mtimes(ca.mtimes(A, x).T, ca.mtimes(A, y)), env), (A @ xv).T @ (A @ yv), rtol=1e-13)
    v = ca.vertcat(x, 1.5, y[1:3])
    assert v.size1() == 7 and np.allclose(val(ca, v, env).reshape(-1), np.r_[xv.reshape(-1), 1.5, yv[1:3].reshape(-1)])
    h = ca.horzcat(x, y)
    assert (h.size1(), h.size2()) == (4, 2) and np.allclose(val(ca, h, env), np.hstack((xv, yv)))
    assert np.allclose(val(ca, h[2, 1], env), yv[2]) and np.allclose(val(ca, h[:, 0], env), xv)
    r = ca.reshape(h, 2, 4)                                  # column-major, like CasADi
    assert np.allclose(val(ca, r, env), np.hstack((xv, yv)).reshape((2, 4), order='F'))
    assert np.allclose(val(ca, ca.vec(h), env).reshape(-1), np.hstack((xv, yv)).reshape(-1, order='F'))
    parts = ca.vertsplit(x)
    assert len(parts) == 4 and np.allclose(val(ca, parts[3], env), xv[3])


def test_substitute_function_and_symvar(ca):
    rng = np.random.default_rng(6)
    x, p = ca.MX.sym('x', 3), ca.MX.sym('p', 2)
    expr = ca.vertcat(x[0] * p[1] + x[2]**2, p[0] - x[1])
    xv, pv = rng.normal(size=(3, 1)), rng.normal(size=(2, 1))
    want = np.array([[xv[0, 0] * pv[1, 0] + xv[2, 0]**2], [pv[0, 0] - xv[1, 0]]])
    assert np.allclose(val(ca, expr, {x: xv, p: pv}), want, rtol=1e-15)
    # substitute: x -> 2 x + 1 leaves p alone
    sub = ca.substitute(expr, x, 2 * x + 1)
    xs = 2 * xv + 1
    assert np.allclose(val(ca, sub, {x: xv, p: pv}), [[xs[0, 0] * pv[1, 0] + xs[2, 0]**2], [pv[0, 0] - xs[1, 0]]], rtol=1e-15)
    f = ca.Function('f', [x, p], [expr, ca.mtimes(x.T, x)])
    out = f(xv, pv)
    assert np.allclose(np.asarray(out[0], float), want, rtol=1e-15) and np.allclose(np.asarray(out[1], float), xv.T @ xv, rtol=1e-15)
    names = sorted(s.name() for s in ca.symvar(expr))
    assert names == ['p', 'x']


def test_linear_solve_and_jacobian_pattern(ca):
    rng = np.random.default_rng(7)
    G = rng.normal(size=(5, 5)) + 5 * np.eye(5)
    h = ca.MX.sym('h', 5)
    hv = rng.normal(size=(5, 1))
    assert np.allclose(val(ca, ca.solve(ca.MX.const(G), h), {h: hv}), np.linalg.solve(G, hv), rtol=1e-12)
    # the sparsity pattern the reference reads off a Jacobian (`problems/distributedproblem.py:105-169`): which outputs depend on which inputs
    x = ca.MX.sym('x', 4)
    g = ca.vertcat(x[0] + x[3], 2 * x[1], x[2] * x[2], 7.0)
    f = ca.Function('g', [x], [g])
    sp = f.sparsity_jac(0, 0)
    assert (sp.size1(), sp.size2()) == (4, 4)
    # (row() of the transpose lists, row by row of the Jacobian, the inputs every output touches: x0, x3 | x1 | x2 | none)
    assert sorted(sp.T.row()) == [0, 1, 2, 3]
