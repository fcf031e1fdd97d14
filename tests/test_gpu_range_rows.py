"""Two-sided rows lb < g < ub (`basics/optilayer.py:634-666`; include/omgx.h version 5) on the device: the velocity limits of
the Holonomic model are two one-sided rows per coefficient in the reference (`vehicles/holonomic.py:73-78`: -v + T vmin <= 0 and
v - vmax <= 0); merged into ONE row  -(vmax - vmin) T <= v - vmax T <= 0  the problem is the same, so the library -- which
solves with such a row doubled, once per side -- must return the same plans, and the multiplier of the merged row is the
difference of the two it replaces."""
import copy

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-6


def _merged(tpl, nlp, p, x):
    """(template with the x- and y-velocity rows merged, [(upper rows, lower rows)])."""
    t2 = copy.copy(tpl)
    t2.lb, t2.ub = tpl.lb.copy(), tpl.ub.copy()
    g = nlp.fg(x, nlp.term_coefs(p))[1]
    pairs = []
    for lo_name, up_name in (('c_0_', 'c_2_'), ('c_1_', 'c_3_')):      # (-dx + T vxmin, dx - T vxmax), (-dy + T vymin, dy - T vymax)
        (lo, r, _), = [v for (lab, nm), v in tpl.con_layout.items() if lab.startswith('vehicle') and nm.startswith(lo_name)]
        (up, r2, _), = [v for (lab, nm), v in tpl.con_layout.items() if lab.startswith('vehicle') and nm.startswith(up_name)]
        assert r == r2
        k = g[lo:lo + r] + g[up:up + r]                 # (-v + vmin T) + (v - vmax T): the same constant in every row
        assert np.abs(k - k[0]).max() < 1e-12 and k[0] < 0
        t2.lb[up:up + r] = k[0]                         # v - vmax T >= (vmin - vmax) T  <=>  -v + vmin T <= 0
        t2.ub[lo:lo + r] = np.inf                       # the lower-limit rows become free rows
        pairs.append((np.arange(up, up + r), np.arange(lo, lo + r)))
    return t2, pairs


def test_merged_velocity_rows_give_the_same_solution():
    import torch
    from omgtools import workloads
    from omgtools.backend import BatchSolver
    from oracle.nlp_numpy import NumpyNLP
    B = 8
    problem, P = workloads.holonomic_p2p(B)
    tpl = problem.father.template
    nlp = NumpyNLP(tpl)
    t2, pairs = _merged(tpl, nlp, P['p'][0], np.random.default_rng(0).normal(size=tpl.n_var))
    assert np.sum(np.isfinite(t2.lb) & np.isfinite(t2.ub) & (t2.lb < t2.ub)) == 26
    a = BatchSolver(tpl, B, options=dict(tol=TOL, max_iter=500))
    b = BatchSolver(t2, B, options=dict(tol=TOL, max_iter=500))
    try:
        ra = a.solve(P['p'], P['x0'])
        rb = b.solve(P['p'], P['x0'], lbg=t2.lb, ubg=t2.ub)
        assert (ra['status'] == 0).all() and (rb['status'] == 0).all()
        assert rb['lam_g'].shape == (B, tpl.n_con)
        assert np.abs(ra['x'] - rb['x']).max() < 1e-6
        for up, lo in pairs:
            assert np.abs(rb['lam_g'][:, up] - (ra['lam_g'][:, up] - ra['lam_g'][:, lo])).max() < 1e-6
            assert np.abs(rb['lam_g'][:, lo]).max() == 0.0                    # free rows
        rest = np.setdiff1d(np.arange(tpl.n_con), np.concatenate([np.r_[u, l] for u, l in pairs]))
        assert np.abs(ra['lam_g'][:, rest] - rb['lam_g'][:, rest]).max() < 1e-6
        # the receding-horizon call shape: device pointers, primal-dual warm start from the multipliers of the caller's rows
        dev = torch.device('cuda', 0)
        f64 = dict(dtype=torch.float64, device=dev)
        b.set_options(warm_start=1)
        b.set_stream(torch.cuda.current_stream().cuda_stream)
        p, x0 = torch.as_tensor(P['p'], **f64), torch.as_tensor(rb['x'], **f64)
        lam = torch.as_tensor(rb['lam_g'], **f64)
        x = torch.empty_like(x0)
        status = torch.zeros(B, dtype=torch.int32, device=dev); iters = torch.zeros(B, dtype=torch.int32, device=dev)
        lb, ub = torch.as_tensor(t2.lb, **f64), torch.as_tensor(t2.ub, **f64)
        b.solve_device(p, x0, lb, ub, x, lam, status, iters, bounds_shared=True)
        torch.cuda.synchronize()
        assert (status.cpu().numpy() == 0).all() and np.median(iters.cpu().numpy()) <= 3      # warm from the solution: it stays (most agents: no iteration)
        # ... the plan, that is: the separating hyperplanes float on flat faces, and the multipliers of their rows with them
        lo_s, hi_s = tpl.entry_range(problem.vehicles[0].label, 'splines_seg0', 'var')
        xw = x.cpu().numpy()
        assert np.abs(xw[:, lo_s:hi_s] - rb['x'][:, lo_s:hi_s]).max() < 1e-4
        for i in range(B):
            fa, fb = (nlp.fg(v[i], nlp.term_coefs(P['p'][i]))[0] for v in (xw, rb['x']))
            assert abs(fa - fb) < 1e-5 * (1 + abs(fb))
        for up, lo in pairs:
            assert np.abs(lam.cpu().numpy()[:, up] - rb['lam_g'][:, up]).max() < 1e-3
        # a two-sided row the template's default bounds did not announce is still refused, loudly
        bad_lb = tpl.lb.copy(); bad_lb[0] = -5.0
        rc = a.solve(P['p'], P['x0'], lbg=bad_lb, ubg=tpl.ub)
        assert (rc['status'] == 3).all()
    finally:
        a.close(); b.close()
