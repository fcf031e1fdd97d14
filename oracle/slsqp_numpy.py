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


def solve_slsqp_reduced(nlp, tpl, x0_user, p, maxiter=400, accept=(0,), viol_tol=1e-8):
    """SLSQP on the CALLER'S OWN problem of a template with lifted auxiliaries (template.py `_append_lifted`: products of more
    than four variable factors, quotients by a variable): the unknowns are the caller's variables only; the auxiliaries are
    functions of them -- their defining rows solved level by level (`NumpyNLP.project_lifted`) -- and the derivatives follow by
    the implicit-function rule,  da/dx = -(dL/da)^-1 dL/dx  with L the defining rows.  The solver under test never sees this
    form: it iterates on the lifted system.  Returns the full vector (auxiliaries included), the objective and the flag."""
    nl = nlp.n_lift
    nv, nc = nlp.n_var - nl, nlp.n_con - nl
    c = nlp.term_coefs(p)
    lb, ub = tpl.lb[:nc], tpl.ub[:nc]
    memo = {}

    def at(v):
        key = v.tobytes()
        if memo.get('key') != key:
            xf = nlp.project_lifted(np.r_[v, np.zeros(nl)], c)
            f, g = nlp.fg(xf, c)
            J = nlp.jac(xf, c)
            dadx = -np.linalg.solve(J[nc:nc + nl, nv:], J[nc:nc + nl, :nv])
            R = J[:, :nv] + J[:, nv:] @ dadx
            memo.update(key=key, x=xf, f=f, g=g[:nc], Jg=R[:nc], df=R[-1])
        return memo
    eq = lb == ub
    ineq = np.isfinite(ub) & ~eq
    lo = np.isfinite(lb) & ~eq
    cons = [{'type': 'ineq', 'fun': lambda v: (ub - at(v)['g'])[ineq], 'jac': lambda v: -at(v)['Jg'][ineq]}]
    if eq.any():
        cons.append({'type': 'eq', 'fun': lambda v: at(v)['g'][eq] - lb[eq], 'jac': lambda v: at(v)['Jg'][eq]})
    if lo.any():
        cons.append({'type': 'ineq', 'fun': lambda v: (at(v)['g'] - lb)[lo], 'jac': lambda v: at(v)['Jg'][lo]})
    out = minimize(lambda v: at(v)['f'], np.asarray(x0_user, float)[:nv], jac=lambda v: at(v)['df'], constraints=cons,
                   method='SLSQP', options={'maxiter': maxiter, 'ftol': 1e-12})
    m = at(out.x)
    g = m['g']
    viol = max((g - ub)[ineq].max() if ineq.any() else 0.0, np.abs(g[eq] - lb[eq]).max() if eq.any() else 0.0,
               (lb - g)[lo].max() if lo.any() else 0.0)
    return m['x'].copy(), float(out.fun), bool(out.status in accept and viol < viol_tol)
