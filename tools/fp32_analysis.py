"""Error analysis of fp32 evaluation inside the config-5 solve (BASELINE.json configs[4]: "fp32 basis eval + fp64 KKT",
SURVEY.md section 7: "needs an error analysis of fp32 T (a x) rows").  CPU only (numpy restatement of the NLP, oracle
port for the reference points).  Three candidate placements of fp32 are emulated and compared with the quantities the
interior-point iteration has to resolve:

  (A) the parameter stage only: B-spline basis rows at t / T and the coefficient slots in fp32, everything after in fp64;
  (B) the row values g(x) in fp32 (term products and row sums), their Jacobian in fp64;
  (C) rows and Jacobian entries in fp32.

What has to be resolved: the slack of an active row on the central path, mu / z (mu = tol / 10 at the end of a solve), and
the tolerance on the dual residual, tol.  Output: a table (stdout; profiles/r04_fp32_error_cfg5.txt is this output)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'omg-tools_amd')); sys.path.insert(0, ROOT)
import numpy as np
from omgtools import workloads
from oracle import port_binding
from oracle.nlp_numpy import NumpyNLP

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
problem, P = workloads.holonomic3d_p2p(n)
tpl = problem.father.template
nlp = NumpyNLP(tpl)
f32 = np.float32


def fg32(x, c):
    """g, f with every term product and every row sum in fp32 (sequential fp32 accumulation per row, term order)."""
    xx = np.r_[x, 1.0].astype(f32)
    mono = np.ones(len(nlp.t_var), dtype=f32)
    for k in range(nlp.t_var.shape[1]):
        mono = (mono * xx[nlp.t_var[:, k]]).astype(f32)
    t = (c.astype(f32) * mono).astype(f32)
    vals = np.zeros(nlp.n_con + 1, dtype=f32)
    for r in range(nlp.n_con + 1):
        acc = f32(0)
        for i in range(nlp.row_ptr[r], nlp.row_ptr[r + 1]):
            acc = f32(acc + t[i])
        vals[r] = acc
    return float(vals[-1]), vals[:-1].astype(float)


def coefs32(p):
    """term coefficients with the parameter stage in fp32 (atoms, basis rows, slot monomials)."""
    a = nlp.atoms(p).astype(f32)          # basis rows recomputed in fp32 below
    from oracle.nlp_numpy import eval_basis_row, OP_BSPL
    for op, i0, i1, i2, i3, i4 in nlp.prog:
        if op == OP_BSPL:
            nfun = i1 - i2 - 1
            a[i4:i4 + nfun] = eval_basis_row(nlp.knots[i0:i0 + i1].astype(f32), i2, f32(a[i3])).astype(f32)
    slots = np.zeros(nlp.n_slots, dtype=f32)
    for s_, pp in enumerate(nlp.slot_pp):
        tot = f32(0)
        for m in range(nlp.pp_ptr[pp], nlp.pp_ptr[pp + 1]):
            v = f32(nlp.pm_coef[m])
            for q in range(nlp.pm_ptr[m], nlp.pm_ptr[m + 1]):
                v = f32(v * a[nlp.pm_atom[q]])
            tot = f32(tot + v)
        slots[s_] = tot
    scale = np.where(nlp.t_slot >= 0, slots[np.maximum(nlp.t_slot, 0)].astype(float), 1.0)
    return nlp.t_coef * scale


rows = []
for tol in (1e-3, 1e-6):
    res = port_binding.solve(tpl, P['p'], P['x0'], n_threads=8, tol=tol, max_iter=500)
    for b in range(n):
        if res['status'][b] != 0:
            continue
        p, x, lam = P['p'][b], res['x'][b], res['lam_g'][b]
        c64 = nlp.term_coefs(p)
        f64_, g64 = nlp.fg(x, c64)
        ub = np.where(np.isfinite(tpl.ub), tpl.ub, np.inf); lb = np.where(np.isfinite(tpl.lb), tpl.lb, -np.inf)
        slack = np.minimum(ub - g64, g64 - lb)
        act = (np.abs(lam) > 1e-3) & np.isfinite(slack) & (tpl.lb != tpl.ub)
        J64 = nlp.jac(x, c64)
        # (A) parameter stage in fp32
        cA = coefs32(p)
        _, gA = nlp.fg(x, cA)
        JA = nlp.jac(x, cA)
        # (B) rows in fp32
        _, gB = fg32(x, c64)
        # (C) Jacobian entries in fp32: rounding every entry and the products lam_r J_rq of the dual residual
        JC = J64.astype(f32).astype(float)
        rd64 = J64[-1] + J64[:-1].T @ lam
        rdA = JA[-1] + JA[:-1].T @ lam
        rdC = (JC[-1].astype(f32) + (JC[:-1].astype(f32).T @ lam.astype(f32))).astype(float)
        rows.append(dict(tol=tol, b=b, act=int(act.sum()), smin=float(slack[act].min()) if act.any() else np.nan,
                         smed=float(np.median(slack[act])) if act.any() else np.nan,
                         gA=float(np.abs(gA - g64).max()), gB=float(np.abs(gB - g64).max()),
                         gB_act=float(np.abs(gB - g64)[act].max()) if act.any() else np.nan,
                         rdA=float(np.abs(rdA - rd64).max()), rdC=float(np.abs(rdC - rd64).max()), rd=float(np.abs(rd64).max()),
                         zmax=float(np.abs(lam).max()), gmax=float(np.abs(g64).max())))

print('config 5 (Holonomic3D, K = 15, 10 spheres: n_var %d, n_con %d), %d agents, points = solutions of the host build' % (tpl.n_var, tpl.n_con, n))
print('%-7s %-3s %-7s %-10s %-10s | %-10s %-10s %-10s | %-10s %-10s %-10s %-8s' % (
    'tol', 'b', 'active', 'min slack', 'med slack', 'A: |dg|', 'B: |dg|', 'B: act', 'A: |d rd|', 'C: |d rd|', '|rd| fp64', 'max |z|'))
for r in rows:
    print('%-7g %-3d %-7d %-10.2e %-10.2e | %-10.2e %-10.2e %-10.2e | %-10.2e %-10.2e %-10.2e %-8.1f' % (
        r['tol'], r['b'], r['act'], r['smin'], r['smed'], r['gA'], r['gB'], r['gB_act'], r['rdA'], r['rdC'], r['rd'], r['zmax']))
for tol in (1e-3, 1e-6):
    sel = [r for r in rows if r['tol'] == tol]
    if not sel:
        continue
    print('tol %g: slack of the active rows min %.1e (central path: mu / z with mu = tol / 10); fp32 rows are off by up to %.1e on them '
          '(%.0f %% of the smallest slack); parameter stage in fp32: rows off by %.1e, dual residual by %.1e (tolerance %g); fp32 '
          'Jacobian: dual residual off by %.1e' % (tol, min(r['smin'] for r in sel), max(r['gB_act'] for r in sel),
                                                   100 * max(r['gB_act'] / r['smin'] for r in sel), max(r['gA'] for r in sel),
                                                   max(r['rdA'] for r in sel), tol, max(r['rdC'] for r in sel)))
