"""ORACLE (test infrastructure, never shipped or measured as product).

Dense numpy statement of the primal-dual interior-point iteration that the HIP
kernel (omg-tools_amd/csrc/omgx_kernels.hip, `ipm_solve_kernel`) runs per agent.
It replaces -- it does not restate -- IPOPT, which is the third-party solver
the reference calls at `problems/problem.py:113` through CasADi
(`basics/optilayer.py:60`; casadi>=3.1.1.post3, `setup.py:29`, not vendored and
not installable here).  PARITY UNPINNED for the solver itself: no IPOPT output is
obtainable in this environment; results are cross-checked against scipy's
independent SLSQP/trust-constr in tests/test_oracle_solver.py instead.

The iteration (same constants as the kernel):
  rows are classified from (lb, ub): equality (lb==ub), upper (ub finite),
  lower (lb finite), free; h = sigma*(g - bound) <= 0 with slack s>0, h+s=0.
  Newton on the perturbed KKT system, condensed to
      [ H + Jh' diag(z/s) Jh + dw I    Je' ] [dx]   [ -(grad f + Jh'(mu/s + (z/s) r_p)) ]
      [ Je                        -dc I ] [y+] = [ -r_E                             ]
  factorised by LDL' without pivoting (quasi-definite); dw is raised until
  all primal pivots are positive (inertia correction).  Fraction-to-boundary
  step, l1-merit backtracking line search, monotone barrier update.
"""
import numpy as np

DEFAULTS = dict(dw_cap_floor=0.03, tol=1e-8, max_iter=200, mu_init=0.1, kappa_eps=10.0, kappa_mu=0.2,
                theta_mu=1.5, tau_min=0.99, s_push=1.0, delta_c=1e-8, eta=1e-4,
                rho=0.1, dw_first=1e-4, dw_inc=10.0, dw_dec=1. / 3., dw_max=1e10, dw_zero=1e-9, dw_heavy=10.0, kappa_eps_heavy=100.0,
                s_max=100.0, kappa_sigma=1e10, max_backtrack=25, max_soc=2,
                slack_reset=True, kappa_push=1.0, stall_iters=20, warm_zmin=1e-8, warm_z_floor=0.1, warm_z_cap=0.01, nu_init=100.0, nu_max=1e8, e_push=1.0, scale_gmax=100.0, s_phi=2.3, s_theta=1.1, delta_sw=1.0,
                gamma_theta=1e-5, gamma_phi=1e-5, filter_size=8, warm_mu_factor=1.0,
                expand_max=16.0, expand_dw=1e-2, expand_from=2)

STATUS = {0: 'Solve_Succeeded', 1: 'Maximum_Iterations_Exceeded',
          2: 'Infeasible_Problem_Detected', 3: 'Unsupported_Bounds',
          4: 'Numerical_Failure'}


def ldl_nopivot(K):
    """In-place-style LDL' without pivoting; returns (L, d)."""
    n = K.shape[0]
    L = np.tril(K).astype(float)
    d = np.zeros(n)
    for j in range(n):
        d[j] = L[j, j]
        if d[j] == 0.0:
            return L, d
        L[j + 1:, j] /= d[j]
        L[j, j] = 1.0
        col = L[j + 1:, j]
        L[j + 1:, j + 1:] -= np.tril(np.outer(col * d[j], col))
    return L, d


def ldl_solve(L, d, b):
    n = len(b)
    y = b.astype(float).copy()
    for j in range(n):
        y[j + 1:] -= L[j + 1:, j] * y[j]
    y /= d
    for j in range(n - 1, -1, -1):
        y[j] -= L[j + 1:, j] @ y[j + 1:]
    return y


def solve_filter(nlp, x0, p, lb, ub, opts=None, z0=None, trace=None):
    o = dict(DEFAULTS)
    o.update(opts or {})
    n = nlp.n_var
    c = nlp.term_coefs(p)
    lb, ub = np.asarray(lb, float), np.asarray(ub, float)
    eq = np.isfinite(lb) & (lb == ub)
    up = np.isfinite(ub) & ~np.isfinite(lb)
    lo = np.isfinite(lb) & ~np.isfinite(ub)
    if np.any(np.isfinite(lb) & np.isfinite(ub) & (lb != ub)):
        return dict(x=np.array(x0, float), lam_g=np.zeros(nlp.n_con), status=3, iters=0)
    iE = np.nonzero(eq)[0]
    iH = np.nonzero(up | lo)[0]
    sig = np.where(up, 1.0, -1.0)[iH]
    bnd = np.where(up, ub, lb)[iH]
    mE, mH = len(iE), len(iH)
    # gradient-based row scaling at x0 (as IPOPT's nlp_scaling_method, g_max=100)
    J0 = nlp.jac(np.array(x0, float), c)
    gmax = np.abs(J0[:nlp.n_con]).max(axis=1)
    rho = np.where(gmax > o['scale_gmax'], o['scale_gmax'] / np.maximum(gmax, 1e-300), 1.0) \
        if o['scale_gmax'] > 0 else np.ones(nlp.n_con)
    bnd = bnd * rho[iH]
    bE = lb[iE] * rho[iE]

    def evaluate(xv):
        fv, gv = nlp.fg(xv, c)
        gv = gv * rho
        return fv, sig * (gv[iH] - bnd), gv[iE] - bE

    x = np.array(x0, float).copy()
    mu = o['mu_init']
    f, h, cE = evaluate(x)
    s = np.maximum(-h, o['s_push'] * np.maximum(1.0, np.abs(h)))
    z = mu / s if z0 is None else np.maximum(np.asarray(z0, float)[iH] * sig, 1e-12)
    y = np.zeros(mE)
    nu = 1.0
    dw_last = 0.0
    status, it, nfact = 1, 0, 0
    filt = None
    eps = np.finfo(float).eps

    def kkt_error(mu_):
        sd = max(o['s_max'], (np.abs(y).sum() + np.abs(z).sum()) / max(1, mE + mH)) / o['s_max']
        return max(np.abs(r_d).max() / sd, np.abs(r_p).max() if mH else 0.0,
                   np.abs(cE).max() if mE else 0.0,
                   (np.abs(s * z - mu_).max() / sd) if mH else 0.0)

    for it in range(o['max_iter'] + 1):
        J = nlp.jac(x, c)
        J[:nlp.n_con] *= rho[:, None]
        gf, Jh, Je = J[nlp.n_con], sig[:, None] * J[iH], J[iE]
        r_p = h + s
        r_d = gf + Je.T @ y + Jh.T @ z
        err0 = kkt_error(0.0)
        if trace is not None:
            trace.append(dict(it=it, f=f, mu=mu, err=err0, inf_pr=max(np.abs(r_p).max() if mH else 0,
                              np.abs(cE).max() if mE else 0), inf_du=np.abs(r_d).max(), dw=dw_last))
        if err0 <= o['tol']:
            status = 0
            break
        if it == o['max_iter']:
            break
        while mu > o['tol'] / 10. and kkt_error(mu) <= o['kappa_eps'] * mu:
            mu = max(o['tol'] / 10., min(o['kappa_mu'] * mu, mu ** o['theta_mu']))
            if filt is not None:
                filt = []
        lam = np.zeros(nlp.n_con)
        lam[iH] = sig * z
        lam[iE] = y
        H = nlp.hess(x, lam * rho, c)
        Sig = z / s
        M = H + Jh.T @ (Sig[:, None] * Jh)
        dw, tries = 0.0, 0
        while True:
            K = np.zeros((n + mE, n + mE))
            K[:n, :n] = M + dw * np.eye(n)
            K[n:, :n] = Je
            K[:n, n:] = Je.T
            K[n:, n:] = -o['delta_c'] * np.eye(mE)
            L, d = ldl_nopivot(K)
            nfact += 1
            if np.all(d[:n] > 0) and np.all(d[n:] < 0):
                break
            dw = (o['dw_first'] if dw_last == 0.0 else max(1e-10, dw_last * o['dw_dec'])) \
                if dw == 0.0 else dw * o['dw_inc']
            tries += 1
            if dw > o['dw_max']:
                status = 4
                break
        if status == 4:
            break
        if dw > 0:
            dw_last = dw

        def newton(rp_, rE_):
            rhs = np.r_[-(gf + Jh.T @ (mu / s + Sig * rp_)), -rE_]
            sol = ldl_solve(L, d, rhs)
            for _ in range(o.get('n_refine', 2)):
                res_ = rhs - K @ sol
                res_[n:] -= o['delta_c'] * sol[n:]
                sol += ldl_solve(L, d, res_)
            dx_ = sol[:n]
            return dx_, sol[n:], -rp_ - Jh @ dx_

        def ftb(v, dv, tau_):
            neg = dv < 0
            return min(1.0, (-tau_ * v[neg] / dv[neg]).min()) if neg.any() else 1.0

        dx, y_new, ds = newton(r_p, cE)
        dz = mu / s - z - Sig * ds
        tau = max(o['tau_min'], 1.0 - mu)
        a_p, a_d = ftb(s, ds, tau), ftb(z, dz, tau)
        theta0 = np.abs(r_p).sum() + np.abs(cE).sum()
        phi0 = f - mu * np.log(s).sum()
        dphi = gf @ dx - mu * (ds / s).sum()
        if filt is None:
            theta_max = 1e4 * max(1.0, theta0)
            theta_min = 1e-4 * max(1.0, theta0)
            filt = []

        def trial(dx_, ds_, alpha_):
            xt_ = x + alpha_ * dx_
            st_ = s + alpha_ * ds_
            ft_, ht_, cEt_ = evaluate(xt_)
            if o['slack_reset']:
                st_ = np.maximum(st_, -ht_)
            th_ = np.abs(ht_ + st_).sum() + np.abs(cEt_).sum()
            ph_ = ft_ - mu * np.log(st_).sum()
            return xt_, st_, ft_, ht_, cEt_, th_, ph_

        def acceptable(th_, ph_, alpha_):
            """(accepted, f_type)"""
            if th_ > theta_max:
                return False, False
            for (tj, pj) in filt:
                if th_ >= tj and ph_ >= pj:
                    return False, False
            ftype = (theta0 <= theta_min and dphi < 0 and
                     alpha_ * (-dphi) ** o['s_phi'] > o['delta_sw'] * theta0 ** o['s_theta'])
            if ftype:
                return ph_ <= phi0 + o['eta'] * alpha_ * dphi + 10 * eps * abs(phi0), True
            return (th_ <= (1 - o['gamma_theta']) * theta0 or
                    ph_ <= phi0 - o['gamma_phi'] * theta0), False

        alpha, ok, nsoc, ftype = a_p, False, 0, False
        if o.get('debug_it') == it:
            ratio = np.where(ds < 0, -s / np.minimum(ds, -1e-300), np.inf)
            idx = np.argsort(ratio)[:12]
            for i in idx:
                print('row %d s=%.3e ds=%.3e z=%.3e dz=%.3e h=%.3e r_p=%.3e ratio=%.3e Jdx=%.3e' % (iH[i], s[i], ds[i], z[i], dz[i], h[i], r_p[i], ratio[i], (Jh @ dx)[i]))
        alpha_min = 1e-10
        for bt in range(o['max_backtrack']):
            xt, st, ft, ht, cEt, tht, pht = trial(dx, ds, alpha)
            ok, ftype = acceptable(tht, pht, alpha)
            if ok:
                break
            if bt == 0 and o['max_soc'] > 0 and tht >= theta0:
                csoc_h, csoc_E, th_prev = alpha * r_p + (ht + st), alpha * cE + cEt, tht
                for nsoc in range(1, o['max_soc'] + 1):
                    dxs, y_s, dss = newton(csoc_h, csoc_E)
                    a_s = ftb(s, dss, tau)
                    xs, ss, fs, hs, cEs, ths, phs = trial(dxs, dss, a_s)
                    ok, ftype = acceptable(ths, phs, a_s)
                    if ok:
                        xt, st, ft, ht, cEt, tht, pht = xs, ss, fs, hs, cEs, ths, phs
                        alpha, y_new = a_s, y_s
                        break
                    if ths > 0.99 * th_prev:
                        break
                    th_prev = ths
                    csoc_h, csoc_E = a_s * csoc_h + (hs + ss), a_s * csoc_E + cEs
                if ok:
                    break
            alpha *= 0.5
        if ok and not ftype:
            filt.append(((1 - o['gamma_theta']) * theta0, phi0 - o['gamma_phi'] * theta0))
            if len(filt) > o['filter_size']:
                filt.pop(0)
        nu = float(len(filt))
        if trace is not None:
            trace[-1].update(alpha=alpha, a_p=a_p, a_d=a_d, ok=ok, nu=nu, tries=tries, nsoc=nsoc, bt=bt)
        x, s, f, h, cE = xt, st, ft, ht, cEt
        z = z + a_d * dz
        y = y + alpha * (y_new - y)
        z = np.minimum(np.maximum(z, mu / (o['kappa_sigma'] * s)), o['kappa_sigma'] * mu / s)
    lam = np.zeros(nlp.n_con)
    lam[iH] = sig * z
    lam[iE] = y
    lam = lam * rho
    return dict(x=x, lam_g=lam, status=status, iters=it, f=f, mu=mu, nfact=nfact)


def solve(nlp, x0, p, lb, ub, opts=None, z0=None, trace=None):
    """Feasible primal-dual interior point with an embedded phase I ("big-M"):
    one extra variable t >= 0 relaxes every row that x0 does not satisfy with
    margin,   h_i(x) - t*v_i <= 0,  c_E(x) - t*c_E(x0) = 0,  cost f + nu*t,
    so (x0, t=1) is strictly feasible and every iterate stays strictly feasible
    (slacks are reset to s = t*v - h after each step).  Plain Armijo on the
    barrier function globalises it."""
    o = dict(DEFAULTS)
    o.update(opts or {})
    n = nlp.n_var
    c = nlp.term_coefs(p)
    lb, ub = np.asarray(lb, float), np.asarray(ub, float)
    eq = np.isfinite(lb) & (lb == ub)
    up = np.isfinite(ub) & ~np.isfinite(lb)
    lo = np.isfinite(lb) & ~np.isfinite(ub)
    if np.any(np.isfinite(lb) & np.isfinite(ub) & (lb != ub)):
        return dict(x=np.array(x0, float), lam_g=np.zeros(nlp.n_con), status=3, iters=0)
    iE = np.nonzero(eq)[0]
    iH = np.nonzero(up | lo)[0]
    sig = np.where(up, 1.0, -1.0)[iH]
    bnd = np.where(up, ub, lb)[iH]
    mE, mH = len(iE), len(iH)
    if getattr(nlp, 'n_lift', 0):          # (omgx_core.h: the solve starts on the defining rows of the lifted auxiliaries)
        x0 = nlp.project_lifted(x0, c)
    J0 = nlp.jac(np.array(x0, float), c)
    gmax = np.abs(J0[:nlp.n_con]).max(axis=1)
    rho = np.where(gmax > o['scale_gmax'], o['scale_gmax'] / np.maximum(gmax, 1e-300), 1.0) \
        if o['scale_gmax'] > 0 else np.ones(nlp.n_con)
    bnd = bnd * rho[iH]
    bE = lb[iE] * rho[iE]

    def evaluate(xv):
        fv, gv = nlp.fg(xv, c)
        gv = gv * rho
        return fv, sig * (gv[iH] - bnd), gv[iE] - bE

    x = np.array(x0, float).copy()
    mu, nu = o['mu_init'], o['nu_init']
    f, h, cE = evaluate(x)
    v = np.maximum(h + o['kappa_push'], 0.0)
    cE0 = cE.copy()
    use_t = bool((v > 0).any() or (np.abs(cE0) > 0).any())
    t = 1.0 if use_t else 0.0
    s = t * v - h
    z = mu / s
    zt = mu / t if use_t else 0.0
    y = np.zeros(mE)
    # (omgx_options compl_inf_tol / constr_viol_tol, version 8: IPOPT's absolute tolerances on the unscaled problem; 0 = not tested)
    compl_tol, viol_tol = float(o.get('compl_inf_tol', 0.0)), float(o.get('constr_viol_tol', 0.0))
    tol_c = compl_tol if 0.0 < compl_tol < o['tol'] else o['tol']
    if z0 is not None:
        # primal-dual warm start: multipliers of the previous solve (unscaled lam_g),
        # barrier parameter from the average complementarity
        lam0 = np.asarray(z0, float) / rho
        z = np.maximum(sig * lam0[iH], np.maximum(o['warm_zmin'], np.minimum(o['warm_z_floor'] * tol_c, o['warm_z_cap'] * tol_c / s if o['warm_z_cap'] > 0 else np.inf)))
        y = lam0[iE].copy()
        mu = float(min(o['mu_init'], max(tol_c / 10., o['warm_mu_factor'] * (s * z).mean())))
    # multiplier of t >= 0: dual feasible in t (nu - v'z - c0'y - zt = 0) rather than on the central
    # path, so that the first Newton step in t is O(t) instead of O(nu t^2 / mu)
    zt = max(mu / t, nu - v @ z - cE0 @ y) if use_t else 0.0
    dw_last = 0.0
    dw_hold, dw_backoff = 0, 1
    status, it, nfact = 1, 0, 0
    ls_fail, full_steps = 0, 0
    alpha_prev = 1.0
    # (omgx_core.h `hess_approx`: no constraint curvature in the Hessian, a damping weight that follows the accepted step length)
    gn, lm = bool(o.get('hess_approx', 0)), 1.0
    N = n + 1                      # (x, t)
    t_check = t
    # inertia correction acts on the variables that appear in a nonlinear term only: the rows
    # and columns of the Lagrangian Hessian of the others are zero, so negative curvature
    # cannot come from them (their block of J' Sigma J is positive definite)
    tv = np.asarray(nlp.t_var)
    nl = np.zeros(N, bool)
    multi = (tv >= 0).sum(axis=1) >= 2
    nl[np.unique(tv[multi][tv[multi] >= 0])] = True
    if o.get('reg_t'):
        nl[n] = True
    reg = np.where(nl, 1.0, o.get('dw_linear', 1e-8)) if o.get('dw_selective', True) else np.ones(N)
    if z0 is None:
        # cold start: leaf (hyperplane) variables are damped less, root variables more (same product)
        ratio = o.get('dw_leaf_ratio_cold', 1.0)
        is_leaf = np.zeros(N, bool)
        is_leaf[np.asarray(getattr(nlp, 'leaf_vars', []), dtype=np.int64)] = True
        reg = np.where(nl, np.where(is_leaf, ratio, 1.0 / ratio), reg)

    def ftb(vv, dv, tau_, cap=1.0):
        neg = dv < 0
        return min(cap, (-tau_ * vv[neg] / dv[neg]).min()) if neg.any() else cap

    for it in range(o['max_iter'] + 1):
        J = nlp.jac(x, c)
        J[:nlp.n_con] *= rho[:, None]
        gf = np.r_[J[nlp.n_con], nu]
        Jh = np.c_[sig[:, None] * J[iH], -v]
        Je = np.c_[J[iE], -cE0]
        rE = cE - t * cE0
        r_d = gf + Je.T @ y + Jh.T @ z
        r_d[n] -= zt
        sd = max(o['s_max'], (np.abs(y).sum() + np.abs(z).sum()) / max(1, mE + mH)) / o['s_max']
        viol = max(np.maximum(h, 0).max() if mH else 0.0, np.abs(cE).max() if mE else 0.0)
        if viol_tol > 0.0:      # the unscaled violation, folded into the same maximum (omgx_core.h `viol_fold`)
            fold = o['tol'] / viol_tol
            viol = max((np.maximum(h, 0) * np.maximum(1.0, fold / np.abs(rho[iH]))).max() if mH else 0.0,
                       (np.abs(cE) * np.maximum(1.0, fold / np.abs(rho[iE]))).max() if mE else 0.0)

        def kkt_error(mu_):
            comp = np.abs(s * z - mu_).max() if mH else 0.0
            if use_t:
                comp = max(comp, abs(t * zt - mu_))
            return max(np.abs(r_d).max() / sd, np.abs(rE).max() if mE else 0.0, comp / sd)

        zh_now = np.abs(z * h).max() if mH else 0.0
        err0 = max(np.abs(r_d[:n]).max() / sd, viol, zh_now / sd)
        if trace is not None:
            trace.append(dict(it=it, f=f, mu=mu, err=err0, inf_pr=viol, inf_du=np.abs(r_d).max(),
                              dw=dw_last, nu=nu, t=t, zt=zt, imax=int(np.argmax(np.abs(r_d)))))
            if o.get('trace_full'):
                trace[-1].update(r_d=r_d.copy(), y=y.copy(), z=z.copy(), s=s.copy(), x=x.copy(), iH=iH, iE=iE,
                                 Jh=Jh.copy(), Je=Je.copy())
        if err0 <= o['tol'] and (compl_tol <= 0.0 or zh_now <= compl_tol):
            status = 0
            break
        if it == o['max_iter']:
            break
        while True:
            r_d[n] = (nu - v @ z - cE0 @ y - zt) if use_t else 0.0
            emu = kkt_error(mu)
            if mu > tol_c / 10. and emu <= (o['kappa_eps_heavy'] if dw_last > o['dw_heavy'] else o['kappa_eps']) * mu:
                mu = max(tol_c / 10., min(o['kappa_mu'] * mu, mu ** o['theta_mu']))
                continue
            if use_t and zt < 0.1 * nu and t > o['tol'] and emu <= o.get('esc_factor', 100) * o['kappa_eps'] * mu:
                if nu >= o['nu_max']:
                    status = 2          # phase I stalls at t > 0: local infeasibility
                    break
                nu *= 10.0
                zt += 0.9 * nu
                continue
            break
        if use_t and viol_tol > 0.0 and status != 2 and mu <= tol_c / 10. and nu < o['nu_max'] and viol > o['tol'] and \
                max(np.abs(r_d[:n]).max(), zh_now) / sd <= o['tol'] and (compl_tol <= 0.0 or zh_now <= compl_tol):
            # (omgx_core.h: only the violation of the rows is left, i.e. the shift t v phase I still holds)
            nu *= 10.0
            zt += 0.9 * nu
        if use_t and it > 0 and it % o['stall_iters'] == 0:
            # phase I stalls: a warm-started solve raises the penalty weight first (local infeasibility
            # only at nu_max), a cold solve gives up at once
            if t > max(o['tol'], 10.0 * mu / nu) and t > o.get('stall_factor', 0.9) * t_check:
                if z0 is None or nu >= o['nu_max']:
                    status = 2
                else:
                    nu *= 10.0
                    zt += 0.9 * nu
            t_check = t
        if status == 2:
            break
        gf[n] = nu
        delta_c = o['delta_c'] * mu ** 0.25          # (omgx_core.h: the equality block's regularisation follows the barrier parameter)
        lam = np.zeros(nlp.n_con)
        lam[iH] = sig * z
        lam[iE] = y
        H = np.zeros((N, N))
        lam_h = np.zeros_like(lam) if gn else lam
        H[:n, :n] = nlp.hess(x, lam_h * rho, c)
        # no variable is damped more than diagonal dominance of the Lagrangian Hessian needs
        gersh = np.r_[nlp.hess_gershgorin(x, lam_h * rho, c), 0.0]
        Sig = z / s
        M = H + Jh.T @ (Sig[:, None] * Jh)
        if use_t:
            M[n, n] += zt / t
        else:
            M[n, n] += 1.0
        # inertia correction tracking (same policy as omgx_core.h): skip the doomed dw = 0 attempt
        # when the previous iteration needed dw > 0; try dw_last/3 only every dw_backoff-th
        # iteration (doubling on a failed decrease, reset by a success); a failed decrease falls
        # back to the value that worked last before escalating by dw_inc
        decreasing = False
        if dw_last < o['dw_zero']:
            dw = 0.0
        elif dw_hold > 0:
            dw = dw_last
            dw_hold -= 1
        else:
            dw = dw_last * o['dw_dec']
            decreasing = True
        if gn and dw < lm:
            dw = lm
        tries = 0
        lv = np.asarray(getattr(nlp, 'leaf_vars', []), dtype=np.int64)
        # (warm starts only, like the kernel)
        root_retry = o.get('root_retry', bool(getattr(nlp, 'wave_ok', False)) and z0 is not None) and len(lv) > 0
        is_root = np.ones(N, bool)
        is_root[lv] = False

        def shift(dw_vec):
            return np.where(nl, np.minimum(dw_vec * reg, gersh + o['dw_cap_floor'] * dw_vec), dw_vec * reg)

        def factor(dw_vec):
            K = np.zeros((N + mE, N + mE))
            K[:N, :N] = M + np.diag(shift(dw_vec))
            K[N:, :N] = Je
            K[:N, N:] = Je.T
            K[N:, N:] = -delta_c * np.eye(mE)
            L, d = ldl_nopivot(K)
            return K, L, d, bool(np.all(d[:N] > 0) and np.all(d[N:] < 0))
        # the Gershgorin guarantee (omgx_core.h gersh_cap, round 5): with dw >= g_q / f_q for every nonlinear variable the capped
        # correction is at least g_q everywhere and the primal block is positive definite -- the escalation ladder is cut at
        # that value, and a correction carried over that exceeds it (> dw_clamp_from) is taken back to it; not after a failed
        # line search (that retry wants the heavier direction)
        gcap = 1.01 * (np.max(gersh[nl] / reg[nl]) if nl.any() else 0.0) + o['dw_first']
        if ls_fail == 0 and dw > o.get('dw_clamp_from', 1.0) and not gn:
            dw = min(dw, gcap)

        def escalated(v, floor_v):
            nxt = o['dw_first'] if v == 0.0 else v * o['dw_inc']
            return gcap if (ls_fail == 0 and v < gcap and nxt > gcap) else nxt
        dw_used = np.zeros(N)          # the inertia correction the factors carry (per variable, before weights and caps)
        while True:
            dw_used = np.full(N, dw)
            K, L, d, ok = factor(dw_used)
            nfact += 1
            if ok:
                if decreasing:
                    dw_backoff = 1
                break
            leaves_pd = False
            if root_retry:
                try:
                    np.linalg.cholesky(K[np.ix_(lv, lv)])
                    leaves_pd = True
                except np.linalg.LinAlgError:
                    leaves_pd = False
            if leaves_pd:
                # (omgx_core.h: the leaf blocks are positive definite at this dw, the root alone has the wrong
                # inertia: raise the inertia correction of the root variables only)
                dwr = dw
                leave_root = False
                while True:
                    if decreasing:
                        decreasing = False
                        dwr = dw_last
                        dw_backoff = min(2 * dw_backoff, o.get('dw_backoff_max', 8))
                        dw_hold = dw_backoff
                    else:
                        # (the root at the guarantee and still the wrong inertia: the leaves lack damping -- back to the
                        # full factorisation at gcap)
                        if ls_fail == 0 and dwr >= gcap and dw < gcap:
                            leave_root = True
                            break
                        dwr = escalated(dwr, None)
                    if dwr > o['dw_max']:
                        status = 4
                        break
                    dw_used = np.where(is_root, dwr, dw)
                    K, L, d, ok = factor(dw_used)
                    nfact += 1
                    if ok:
                        break
                if status == 4:
                    break
                if leave_root:
                    dw = gcap
                    decreasing = False
                    continue
                dw = dwr
                break
            if decreasing:
                decreasing = False
                dw = dw_last
                dw_backoff = min(2 * dw_backoff, o.get('dw_backoff_max', 8))
                dw_hold = dw_backoff
            else:
                dw = escalated(dw, None)
            tries += 1
            if dw > o['dw_max']:
                status = 4
                break
        if status == 4:
            break
        dw_last = dw
        # r_p == 0 by construction (slack reset)
        g_bar = gf + Jh.T @ (mu / s)
        if use_t:
            g_bar[n] -= mu / t
        rhs = np.r_[-g_bar, -rE]
        sol = ldl_solve(L, d, rhs)
        for _ in range(o.get('n_refine', 0)):
            res_ = rhs - K @ sol
            res_[N:] -= delta_c * sol[N:]
            sol += ldl_solve(L, d, res_)
        # iterative refinement of a regularised step (omgx_core.h, option refine: version 9): with D the inertia correction the
        # factors carry, K s1 = -r - D s1, so s1 + s2 with (K + D) s2 = D s1 is one term closer to the step of the unregularised
        # system; taken when the term is no longer than the step, from the second iteration of a solve on, not after an iteration
        # that accepted less than a tenth of its step, in cold solves only once phase I is over; templates on the wave path with
        # the exact Hessian.  A refined step stands only if its first trial is accepted: else the plain step takes the whole
        # line search (the pass loop below), like there
        refine_on = int((opts or {}).get('refine', 0)) > 0      # (omgx_options refine: off by default)
        term = None
        if refine_on and it >= 1 and alpha_prev >= 0.1 and (z0 is not None or not use_t or t <= 1e-4) and \
                bool(getattr(nlp, 'wave_ok', False)) and not getattr(nlp, 'n_lift', 0) and not gn and (dw_used[lv].max() if len(lv) else dw_used.max()) > 0.0:
            cand = ldl_solve(L, d, np.r_[shift(dw_used) * sol[:N], np.zeros(mE)])
            if np.abs(cand[:N]).max() <= np.abs(sol[:N]).max():
                term = cand
                sol = sol + term
        while True:          # (two passes at most: a refined step whose first trial is not accepted gives way to the plain step)
            back_to_plain = False
            dxt, y_new = sol[:N], sol[N:]
            if not use_t:
                dxt[n] = 0.0
            ds = -Jh @ dxt
            dz = mu / s - z - Sig * ds
            dt = dxt[n]
            dzt = (mu / t - zt - (zt / t) * dt) if use_t else 0.0
            tau = max(o['tau_min'], 1.0 - mu)
            a_p = ftb(s, ds, tau, o['expand_max'])            # (the boundary step up to expand_max: omgx_core.h OMGX_EXPAND_MAX)
            a_d = ftb(z, dz, tau)
            if use_t:
                if dt < 0:
                    a_p = min(a_p, -tau * t / dt)
                if dzt < 0:
                    a_d = min(a_d, -tau * zt / dzt)
            thE = np.abs(rE).sum()
            nuE = 2.0 * max(1.0, np.abs(y_new).max() if mE else 0.0)
            # (quasi-definite system: the linearised equality residual after the full step is delta_c * y_new; the merit
            # function counts the equality residual only above that floor -- omgx_core.h)
            floorE = delta_c * (np.abs(y_new).sum() if mE else 0.0)
            phi0 = f + nu * t - mu * np.log(s).sum() - (mu * np.log(t) if use_t else 0.0) + nuE * max(0.0, thE - floorE)
            dphi = g_bar @ dxt - nuE * max(0.0, thE - floorE)
            # a step of the regularised system may be offered longer (omgx_core.h: the crawl of the proximal iteration)
            a_bnd = a_p
            a_p = min(a_bnd, 1.0)
            phi_noise = abs(a_p * dphi) <= o.get('phi_noise', 1e-10) * (1.0 + abs(phi0))      # (omgx_core.h OMGX_PHI_NOISE)
            if not phi_noise and dw_last > o['expand_dw'] and full_steps >= o['expand_from']:
                ex = 1.0
                while 2.0 * ex <= a_bnd and 2.0 * ex <= o['expand_max']:
                    ex *= 2.0
                a_p = min(a_bnd, ex)
            alpha, ok = a_p, False
            # second-order correction (omgx_core.h, option max_soc; templates on the wave path like there): when the first
            # trial is rejected, one more solve with the factors of the iteration for what the rows moved beyond their
            # linearisation; the corrected step is offered once before the halving starts
            n_soc = int((opts or {}).get('max_soc', 1))          # (omgx_options max_soc: a count, default 1; DEFAULTS' entry belongs to solve_filter)
            soc = 0 if n_soc > 0 else 2
            soc_rounds = 0
            soc_levels = n_soc > 0 and bool(getattr(nlp, 'wave_ok', False)) and not gn      # (omgx_core.h: templates on the wave path)
            d_c = None
            for bt in range(o['max_backtrack']):
                step = alpha * dxt + d_c if soc == 1 else alpha * dxt
                xt = x + step[:n]
                if getattr(nlp, 'n_lift', 0):          # (omgx_core.h lift_project: the auxiliaries of lifted products follow the trial point)
                    xt = nlp.project_lifted(xt, c)
                tt = t + step[n]
                ft, ht, cEt = evaluate(xt)
                st = tt * v - ht
                # (omgx_core.h OMGX_FTB_ACTUAL: every row really keeps half of what the linear fraction-to-boundary rule leaves it)
                if (st - o.get('ftb_actual', 0.5) * (1.0 - tau) * s).min() > 0 and (not use_t or tt > 0):
                    phit = ft + nu * tt - mu * np.log(st).sum() - (mu * np.log(tt) if use_t else 0.0) \
                        + nuE * max(0.0, np.abs(cEt - tt * cE0).sum() - floorE)
                    if phi_noise or phit <= phi0 + o['eta'] * alpha * dphi or phit - phi0 <= 10 * np.finfo(float).eps * abs(phi0):
                        ok = True
                        break
                if term is not None and bt == 0:      # (omgx_core.h: the refined step stands only if its first trial is accepted)
                    sol = sol - term
                    term = None
                    back_to_plain = True
                    break
                row_rejected = not ((st - o.get('ftb_actual', 0.5) * (1.0 - tau) * s).min() > 0)
                if soc == 1 and soc_rounds < n_soc and row_rejected:
                    soc = 0                      # (omgx_core.h, max_soc > 1: one more correction from where the corrected trial landed)
                if soc == 1:
                    soc = 2
                    if row_rejected and soc_levels:             # (omgx_core.h, round 6: the correction is offered again at the shorter step when a row rejected this one)
                        soc, soc_rounds, d_c = 0, 0, None
                    alpha *= 0.5
                    continue
                if soc == 0:
                    soc = 1
                    soc_rounds += 1
                    e = (s + alpha * ds) - st
                    eE = (cEt - tt * cE0) - (1.0 - alpha) * rE
                    rhs2 = np.r_[-(Jh.T @ (Sig * e)), -eE]
                    if not use_t:
                        rhs2[n] = 0.0
                    d_new = ldl_solve(L, d, rhs2)[:N]
                    if not use_t:
                        d_new[n] = 0.0
                    d_c = d_new if d_c is None else d_c + d_new
                    continue
                if soc == 2 and soc_levels and row_rejected:
                    soc, soc_rounds, d_c = 0, 0, None
                alpha *= 0.5
            if not back_to_plain:
                break
        if trace is not None:
            trace[-1].update(alpha=alpha, a_p=a_p, a_d=a_d, ok=ok, tries=tries, bt=bt, dphi=dphi)
        if not ok:
            # (omgx_core.h: the iteration again with a heavier inertia correction, up to 3 times in a row)
            if ls_fail < o.get('ls_retry', 3):
                ls_fail += 1
                dw_last = max(dw_last, o['dw_first']) * o.get('ls_retry_dw', 100.0)
                dw_hold, dw_backoff = 2, o.get('dw_backoff_max', 8)
                continue
            status = 4
            break
        if gn:
            lm = min(lm * 4.0, 1e8) if alpha < 0.25 else (max(lm * 0.5, 1e-6) if alpha >= 1.0 else lm)
        ls_fail = 0
        full_steps = full_steps + 1 if (alpha >= 1.0 and alpha == a_p) else 0
        alpha_prev = alpha
        x, t, s, f, h, cE = xt, tt, st, ft, ht, cEt
        if z0 is not None:
            # component-wise dual step: every multiplier takes its full Newton step, clipped at the
            # fraction-to-boundary rule on its own (one multiplier on its way to zero does not hold
            # back the others)
            z = np.maximum(z + dz, (1.0 - tau) * z)
            zt = max(zt + dzt, (1.0 - tau) * zt)
        else:
            z = z + a_d * dz
            zt = zt + a_d * dzt
        y = y + min(alpha, 1.0) * (y_new - y)
        z = np.minimum(np.maximum(z, mu / (o['kappa_sigma'] * s)), o['kappa_sigma'] * mu / s)
        if use_t:
            zt = min(max(zt, mu / (o['kappa_sigma'] * t)), o['kappa_sigma'] * mu / t)
    lam = np.zeros(nlp.n_con)
    lam[iH] = sig * z
    lam[iE] = y
    lam = lam * rho
    return dict(x=x, lam_g=lam, status=status, iters=it, f=f, mu=mu, nfact=nfact, nu=nu, t=t)
