"""ORACLE (test infrastructure): optimality conditions of a returned solution, evaluated with the numpy
restatement of the NLP (oracle/nlp_numpy.py, pinned to the reference's construct code by tests/golden).
Independent of how the solution was computed: this is the parity statement that holds without IPOPT
outputs -- "a KKT point of the reference's NLP at `ipopt.tol`" (`problems/problem.py:57`).
Only tests/, __graft_entry__.smoke() and bench.py may import this module."""
import numpy as np


def kkt_report(nlp, tpl, p, x, lam):
    """(max violation, min multiplier sign, max |lam * slack|, max |grad L|) of one agent, rows scaled
    like IPOPT's gradient-based scaling (g_max = 100), multipliers like its s_d."""
    c = nlp.term_coefs(p)
    f, g = nlp.fg(x, c)
    J = nlp.jac(x, c)
    gmax = np.abs(J[:-1]).max(axis=1)
    rho = np.where(gmax > 100., 100. / np.maximum(gmax, 1e-300), 1.0)
    up, lo = np.isfinite(tpl.ub), np.isfinite(tpl.lb)
    viol = max(((g - tpl.ub) * rho)[up].max(initial=0.), ((tpl.lb - g) * rho)[lo].max(initial=0.))
    ineq_up, ineq_lo = up & ~lo, lo & ~up
    sign = min(lam[ineq_up].min(initial=0.), (-lam[ineq_lo]).min(initial=0.))
    slack = np.where(ineq_up, tpl.ub - g, np.where(ineq_lo, g - tpl.lb, 0.0))
    comp = np.abs(lam * slack).max()
    sd = max(100., np.abs(lam / rho).mean()) / 100.
    stat = np.abs(J[-1] + J[:-1].T @ lam).max() / sd
    return viol, sign, comp / sd, stat


def assert_kkt(nlp, tpl, p, x, lam, tol, who=''):
    viol, sign, comp, stat = kkt_report(nlp, tpl, p, x, lam)
    assert viol < 2 * tol, (who, 'violation', viol)              # h <= t v with t at the phase-I floor
    assert sign > -1e-12, (who, 'multiplier sign', sign)
    assert comp < 3 * tol, (who, 'complementarity', comp)        # s z ~ mu <= kappa_eps * tol / 10 ... tol
    assert stat < 1.5 * tol, (who, 'stationarity', stat)
    return viol, sign, comp, stat


def second_order_report(nlp, tpl, p, x, lam, lb=None, ub=None, mult_tol=1e-4, slack_tol=1e-2):
    """Second-order NECESSARY condition at a KKT point, independent of any solver: the Lagrangian Hessian of the reference's NLP
    (numpy restatement, the multipliers as returned) restricted to the tangent space of the active rows -- equality rows and
    inequality rows with a multiplier above `mult_tol` and a slack below `slack_tol` -- has no negative eigenvalue.  Together with the
    first-order conditions (`assert_kkt`) this excludes saddle points and maxima; the L1 objectives of these problems leave flat
    directions (zero eigenvalues), so strict sufficiency is not to be had.  Returns (smallest eigenvalue, largest eigenvalue,
    dimension of the tangent space, number of active rows)."""
    lb = tpl.lb if lb is None else np.asarray(lb, float)
    ub = tpl.ub if ub is None else np.asarray(ub, float)
    c = nlp.term_coefs(p)
    g = nlp.fg(x, c)[1]
    J = nlp.jac(x, c)[:-1]
    H = nlp.hess(x, lam, c)
    eq = np.isfinite(lb) & (lb == ub)
    up, lo = np.isfinite(ub) & ~eq, np.isfinite(lb) & ~eq
    slack = np.where(up, ub - g, np.where(lo, g - lb, np.inf))
    act = eq | ((np.abs(lam) > mult_tol) & (slack < slack_tol))
    Ja = J[act]
    if Ja.shape[0] == 0:
        Z = np.eye(J.shape[1])
    else:
        _, S, Vt = np.linalg.svd(Ja, full_matrices=True)
        rank = int((S > 1e-9 * S[0]).sum())
        Z = Vt[rank:].T
    if Z.shape[1] == 0:
        return 0.0, 0.0, 0, int(act.sum())
    Hr = Z.T @ H @ Z
    ev = np.linalg.eigvalsh(0.5 * (Hr + Hr.T))
    return float(ev.min()), float(ev.max()), int(Z.shape[1]), int(act.sum())
