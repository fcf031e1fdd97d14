"""ORACLE (test infrastructure): two-sided rows lb < g < ub for the checker side -- the restatement of what the library
does around its solve kernel (omg-tools_amd/csrc/omgx.hip `expand_range_rows`, the `range_*` kernels): every row whose
default bounds are both finite and different is solved twice, row r as g <= ub and a copy behind the last row as g >= lb;
the multiplier the caller sees is the sum of the two.  Reference: `basics/optilayer.py:634-666` (define_constraint with
both bounds) -- CasADi / IPOPT takes such rows natively."""
import copy

import numpy as np


def range_rows(tpl):
    return np.nonzero(np.isfinite(tpl.lb) & np.isfinite(tpl.ub) & (tpl.lb < tpl.ub))[0]


def expand_template(tpl):
    """(template with the two-sided rows doubled, src [n_range], dup [n_con] (-1: none)); the input itself when it has none."""
    src = range_rows(tpl)
    dup = np.full(tpl.n_con, -1, dtype=np.int64)
    if len(src) == 0:
        return tpl, src, dup
    dup[src] = tpl.n_con + np.arange(len(src))
    t2 = copy.copy(tpl)
    rp = np.asarray(tpl.row_ptr)
    order = np.r_[np.arange(tpl.n_con), src, tpl.n_con]                    # the objective row stays last
    counts = (rp[1:] - rp[:-1])[order]
    idx = np.concatenate([np.arange(rp[r], rp[r + 1]) for r in order]) if counts.sum() else np.zeros(0, int)
    t2.row_ptr = np.r_[0, np.cumsum(counts)].astype(np.int32)
    t2.t_coef, t2.t_slot, t2.t_nv = np.asarray(tpl.t_coef)[idx], np.asarray(tpl.t_slot)[idx], np.asarray(tpl.t_nv)[idx]
    t2.t_var = np.asarray(tpl.t_var)[idx]
    t2.n_con, t2.n_terms = tpl.n_con + len(src), len(idx)
    t2.lb = np.r_[np.where(dup >= 0, -np.inf, tpl.lb), tpl.lb[src]]
    t2.ub = np.r_[tpl.ub, np.full(len(src), np.inf)]
    t2.con_layout = dict(tpl.con_layout)
    t2.con_layout[('range', 'lower_sides')] = (tpl.n_con, len(src), 1)
    if getattr(tpl, 'n_lift', 0):
        t2.lift_row0 = getattr(tpl, 'lift_row0', tpl.n_con - tpl.n_lift)      # (the defining rows of lifted auxiliaries keep their place)
    if hasattr(t2, 'plan'):
        t2.plan = None
    return t2, src, dup


def expand_bounds(lbg, ubg, src, dup):
    lbg, ubg = np.atleast_2d(np.asarray(lbg, float)), np.atleast_2d(np.asarray(ubg, float))
    lb = np.concatenate([np.where(dup >= 0, -np.inf, lbg), lbg[:, src]], axis=1)
    ub = np.concatenate([ubg, np.full((ubg.shape[0], len(src)), np.inf)], axis=1)
    return lb, ub


def expand_lam(lam, src, dup):
    lam = np.atleast_2d(np.asarray(lam, float))
    return np.concatenate([np.where(dup >= 0, np.maximum(lam, 0.0), lam), np.minimum(lam[:, src], 0.0)], axis=1)


def contract_lam(lam2, n_con, src, dup):
    out = lam2[:, :n_con].copy()
    out[:, src] += lam2[:, n_con:n_con + len(src)]
    return out
