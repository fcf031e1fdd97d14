"""oracle/slsqp_numpy.py -- TEST INFRASTRUCTURE (checker / reported CPU baseline), never a product path.

An INDEPENDENT solver (scipy SLSQP: dense SQP with its own QP solver and line search) on the restated NLP
(oracle/nlp_numpy.py) -- the stand-in SURVEY.md 8c prescribes for the unobtainable CasADi / IPOPT outputs behind
`problems/problem.py:113`.  From the same initial guess it must reach the same local minimum as the interior-point path.
Imported by tests/ (parity fixtures, `tests/slsqp_reference.py`) and by bench.py's `cpu_baseline` leg only."""
import numpy as np
from scipy.optimize import minimize


def solve_slsqp(nlp, tpl, x0, p, maxiter=400, accept=(0,), viol_tol=1e-8):
    """accept: SLSQP exit codes taken as converged (8 = 'positive directional derivative for linesearch' is where it
    stops on the larger classes once no step improves the objective any more; the feasibility bound still applies)."""
    c = nlp.term_coefs(p)
    eq = tpl.lb == tpl.ub
    ineq = np.isfinite(tpl.ub) & ~eq
    lo = np.isfinite(tpl.lb) & ~eq          # lower sides: rows g >= lb and the lower side of two-sided rows lb <= g <= ub
    cons = [{'type': 'eq', 'fun': lambda v: nlp.fg(v, c)[1][eq] - tpl.lb[eq], 'jac': lambda v: nlp.jac(v, c)[:-1][eq]},
            {'type': 'ineq', 'fun': lambda v: (tpl.ub - nlp.fg(v, c)[1])[ineq], 'jac': lambda v: -nlp.jac(v, c)[:-1][ineq]}]
    if lo.any():
        cons.append({'type': 'ineq', 'fun': lambda v: (nlp.fg(v, c)[1] - tpl.lb)[lo], 'jac': lambda v: nlp.jac(v, c)[:-1][lo]})
    out = minimize(lambda v: nlp.fg(v, c)[0], x0, jac=lambda v: nlp.jac(v, c)[-1], constraints=cons,
                   method='SLSQP', options={'maxiter': maxiter, 'ftol': 1e-12})
    g = nlp.fg(out.x, c)[1]
    viol = max((g - tpl.ub)[ineq].max(), np.abs(g[eq] - tpl.lb[eq]).max(), (tpl.lb - g)[lo].max() if lo.any() else 0.0)
    return out.x, float(out.fun), bool(out.status in accept and viol < viol_tol)
