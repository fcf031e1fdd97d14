"""The receding-horizon glue either side of the solve on the device: prediction with
higher derivatives and with the reference's RK4 statements, `Vehicle.store` as a kernel and fused behind the
solve -- each against the host front end's own functions (omgtools.vehicles / omgtools.splines, which
tests/test_front_end_* pin to the reference)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _setup(n, scen='holonomic_p2p'):
    import omgtools.backend as be
    from omgtools import scenarios
    saved = be.create_nlp
    be.create_nlp = lambda tpl, opt, name='': (None, 0.)
    try:
        problem, P = getattr(scenarios, scen)(n)
    finally:
        be.create_nlp = saved
    return problem, P


def _random_plan(problem, P, rng):
    """x with random spline coefficients (smooth enough to be a plan)."""
    tpl, veh = problem.father.template, problem.vehicles[0]
    lo, hi = tpl.entry_range(veh.label, 'splines_seg0', 'var')
    x = np.array(P['x0'])
    x[:, lo:hi] += rng.normal(scale=0.3, size=(x.shape[0], hi - lo))
    return x, lo


def test_predict_with_second_derivative_matches_the_spline_algebra():
    """Quadrotor: spl0, dspl0, ddspl0 (`vehicles/quadrotor.py:76-85`) from the plan at tau."""
    import torch
    from omgtools.backend import BatchSolver
    from omgtools.splines import BSpline
    problem, P = _setup(6, 'quadrotor_p2p')
    tpl, veh = problem.father.template, problem.vehicles[0]
    T = float(problem.options['horizon_time'])
    x, lo = _random_plan(problem, P, np.random.default_rng(3))
    L, nd = len(veh.basis), veh.n_spl
    solver = BatchSolver(tpl, 6)
    xd = torch.as_tensor(x, device='cuda:0')
    pd = torch.zeros((6, tpl.n_par), dtype=torch.float64, device='cuda:0')
    offs = [tpl.entry_range(veh.label, nm, 'par')[0] for nm in ('spl0', 'dspl0', 'ddspl0')]
    o_t = tpl.entry_range(problem.label, 't', 'par')[0]
    for tau in (0.0, 0.013, 1.0 / 13, 0.37, 1.0):
        solver.predict_ex(xd, pd, lo, nd, veh.degree, veh.basis.knots, tau, 1.0 / T, offs, o_t, 0.25)
        got = pd.cpu().numpy()
        for b in range(6):
            for k in range(nd):
                s = BSpline(veh.basis, x[b, lo + k * L:lo + (k + 1) * L])
                for o in range(3):
                    want = (s.derivative(o) if o else s)(tau) / T ** o
                    assert abs(got[b, offs[o] + k] - want) < 1e-10 * max(1.0, abs(want)), (tau, b, k, o)
        assert np.all(got[:, o_t] == 0.25)
    solver.close()


def test_rk4_prediction_follows_the_reference_statements():
    """`Vehicle::integrate` (`export/vehicles/Vehicle.cpp:82-110`) with `ode` = input (holonomic.py:126-127)."""
    import torch
    from omgtools.backend import BatchSolver, PREDICT_RK4
    from omgtools.splines import BSpline
    problem, P = _setup(5)
    tpl, veh = problem.father.template, problem.vehicles[0]
    T = float(problem.options['horizon_time'])
    x, lo = _random_plan(problem, P, np.random.default_rng(4))
    L, nd = len(veh.basis), veh.n_dim
    solver = BatchSolver(tpl, 5)
    xd = torch.as_tensor(x, device='cuda:0')
    pd = torch.zeros((5, tpl.n_par), dtype=torch.float64, device='cuda:0')
    state = np.random.default_rng(5).normal(size=(5, nd))
    sd = torch.as_tensor(state, device='cuda:0')
    o_s = tpl.entry_range(veh.label, 'state0', 'par')[0]
    o_i = tpl.entry_range(veh.label, 'input0', 'par')[0]
    sample_time, steps = 0.01, 10                                  # update_time 0.1 s at 100 Hz
    tau = 0.3 + steps * sample_time / T
    solver.predict_ex(xd, pd, lo, nd, veh.degree, veh.basis.knots, tau, 1.0 / T, [o_s, o_i], -1, 0.,
                      mode=PREDICT_RK4, state_in=sd, n_sub=steps, dtau=sample_time / T)
    got = pd.cpu().numpy()
    for b in range(5):
        spl = [BSpline(veh.basis, x[b, lo + k * L:lo + (k + 1) * L]) for k in range(nd)]
        inp = np.array([[s.derivative(1)(0.3 + i * sample_time / T) / T for s in spl] for i in range(steps + 1)])
        state0 = state[b].copy()
        stateT = state0.copy()
        for i in range(steps):                                     # the C++ loop, line by line
            k1 = inp[i]
            k2 = inp[i]
            k3 = inp[i]
            k4 = inp[i + 1]
            stateT += (sample_time / 6.0) * (k1 + 2 * k2 + 2 * k3 + k4)
        assert np.abs(got[b, o_s:o_s + nd] - stateT).max() < 1e-12
        assert np.abs(got[b, o_i:o_i + nd] - inp[steps]).max() < 1e-12
    solver.close()


def test_store_kernel_and_fused_store_match_splines2signals():
    """`Vehicle.store` -> `splines2signals` (state, input, dinput, v_tot on the sample grid): the stand-alone
    kernel on a given x, and the same arrays written by the solve kernel for the solution it found."""
    import torch
    from omgtools.backend import BatchSolver
    from omgtools.splines import BSpline
    B = 8
    problem, P = _setup(B)
    tpl, veh = problem.father.template, problem.vehicles[0]
    T = float(problem.options['horizon_time'])
    lo = tpl.entry_range(veh.label, 'splines_seg0', 'var')[0]
    L, nd, n_samp, sample_time = len(veh.basis), veh.n_dim, 1001, 0.01
    dev = torch.device('cuda', 0)
    f64 = dict(dtype=torch.float64, device=dev)
    solver = BatchSolver(tpl, B, options=dict(tol=1e-6, max_iter=300))
    out = torch.zeros((B, 3, nd, n_samp), **f64)
    vt = torch.zeros((B, n_samp), **f64)
    t0 = torch.zeros(B, **f64)
    knots = veh.basis.knots
    # fused: the solve writes the trajectories of its own solution
    solver.set_store(out, vt, t0, lo, nd, veh.degree, knots, 3, n_samp, sample_time / T, 1.0 / T)
    xd, pd = torch.as_tensor(P['x0'], **f64), torch.as_tensor(P['p'], **f64)
    lb, ub = torch.as_tensor(tpl.lb, **f64), torch.as_tensor(tpl.ub, **f64)
    xs, lam = torch.empty_like(xd), torch.zeros((B, tpl.n_con), **f64)
    st, it = torch.zeros(B, dtype=torch.int32, device=dev), torch.zeros(B, dtype=torch.int32, device=dev)
    solver.solve_device(pd, xd, lb, ub, xs, lam, st, it, bounds_shared=True)
    torch.cuda.synchronize()
    assert (st == 0).all()
    fused, fused_v = out.cpu().numpy().copy(), vt.cpu().numpy().copy()
    x = xs.cpu().numpy()
    time = np.linspace(0., (n_samp - 1) * sample_time, n_samp)
    for b in range(B):
        spl = [BSpline(veh.basis, x[b, lo + k * L:lo + (k + 1) * L]) for k in range(nd)]
        # reference units: the stored splines live on [0, T]; derivative order o scales with 1 / T^o
        sig = {'state': np.array([s(time / T) for s in spl]),
               'input': np.array([s.derivative(1)(time / T) / T for s in spl]),
               'dinput': np.array([s.derivative(2)(time / T) / T ** 2 for s in spl])}
        for o, key in enumerate(('state', 'input', 'dinput')):
            scale = max(1.0, np.abs(sig[key]).max())
            assert np.abs(fused[b, o] - sig[key]).max() < 1e-10 * scale, (b, key)
        assert np.abs(fused_v[b] - np.sqrt((sig['input'] ** 2).sum(axis=0))).max() < 1e-10
    # stand-alone kernel on the same x: the same bits
    solver.set_store(None)
    out2, vt2 = torch.zeros_like(out), torch.zeros_like(vt)
    solver.store(xs, out2, vt2, t0, lo, nd, veh.degree, knots, 3, n_samp, sample_time / T, 1.0 / T)
    torch.cuda.synchronize()
    assert np.array_equal(out2.cpu().numpy(), fused) and np.array_equal(vt2.cpu().numpy(), fused_v)
    # and a solve after set_store(None) leaves the arrays alone
    out.zero_()
    solver.solve_device(pd, xd, lb, ub, xs, lam, st, it, bounds_shared=True)
    torch.cuda.synchronize()
    assert float(out.abs().max()) == 0.0
    solver.close()


def test_fused_store_survives_a_knot_crossing_shift():
    """set_store -> shift (first use of the handle's shift tables) -> solve -> the trajectories the solve wrote equal
    the stand-alone kernel's on the same solution -> destroy.  (Round-2 advice: the first shift on a handle used to
    free the device copy of the store specification; the receding-horizon loop of INTEGRATION.md 4 is this
    sequence.)"""
    import torch
    from omgtools.backend import BatchSolver
    from omgtools.splines import shiftoverknot_T
    B = 8
    problem, P = _setup(B)
    tpl, veh = problem.father.template, problem.vehicles[0]
    T = float(problem.options['horizon_time'])
    lo = tpl.entry_range(veh.label, 'splines_seg0', 'var')[0]
    L, nd, n_samp, sample_time = len(veh.basis), veh.n_dim, 501, 0.01
    dev = torch.device('cuda', 0)
    f64 = dict(dtype=torch.float64, device=dev)
    solver = BatchSolver(tpl, B, options=dict(tol=1e-6, max_iter=300))
    out, vt, t0 = torch.zeros((B, 3, nd, n_samp), **f64), torch.zeros((B, n_samp), **f64), torch.zeros(B, **f64)
    knots = veh.basis.knots
    solver.set_store(out, vt, t0, lo, nd, veh.degree, knots, 3, n_samp, sample_time / T, 1.0 / T)
    xd, pd = torch.as_tensor(P['x0'], **f64), torch.as_tensor(P['p'], **f64)
    lb, ub = torch.as_tensor(tpl.lb, **f64), torch.as_tensor(tpl.ub, **f64)
    xs, lam = torch.empty_like(xd), torch.zeros((B, tpl.n_con), **f64)
    st, it = torch.zeros(B, dtype=torch.int32, device=dev), torch.zeros(B, dtype=torch.int32, device=dev)
    solver.solve_device(pd, xd, lb, ub, xs, lam, st, it, bounds_shared=True)
    torch.cuda.synchronize()
    assert (st == 0).all()
    # knot crossing: the plan of every agent is shifted on the device (allocates the handle's shift tables)
    Tm = shiftoverknot_T(veh.basis)
    ents = np.array([[lo, L, nd, 0]], dtype=np.int32)
    mask = torch.ones(B, dtype=torch.uint8, device=dev)
    want = np.concatenate([Tm @ xs.cpu().numpy()[:, lo + k * L:lo + (k + 1) * L].T for k in range(nd)]).T
    solver.shift(xs, mask, ents, np.asarray(Tm, dtype=np.float64).reshape(-1), device=True)
    torch.cuda.synchronize()
    assert np.abs(xs.cpu().numpy()[:, lo:lo + nd * L] - want).max() < 1e-12
    # the next solve (from the shifted plan) still writes its trajectories where set_store said
    out.zero_(); vt.zero_()
    x2 = torch.empty_like(xd)
    solver.solve_device(pd, xs, lb, ub, x2, lam, st, it, bounds_shared=True)
    torch.cuda.synchronize()
    assert (st == 0).all()
    fused, fused_v = out.cpu().numpy().copy(), vt.cpu().numpy().copy()
    assert np.abs(fused).max() > 0.1
    # a second set_store (copies into the same device block) and the stand-alone kernel: the same bits
    out2, vt2 = torch.zeros_like(out), torch.zeros_like(vt)
    solver.set_store(out2, vt2, t0, lo, nd, veh.degree, knots, 3, n_samp, sample_time / T, 1.0 / T)
    solver.solve_device(pd, xs, lb, ub, x2, lam, st, it, bounds_shared=True)
    torch.cuda.synchronize()
    assert np.array_equal(out2.cpu().numpy(), fused) and np.array_equal(vt2.cpu().numpy(), fused_v)
    solver.set_store(None)
    out3, vt3 = torch.zeros_like(out), torch.zeros_like(vt)
    solver.store(x2, out3, vt3, t0, lo, nd, veh.degree, knots, 3, n_samp, sample_time / T, 1.0 / T)
    torch.cuda.synchronize()
    assert np.array_equal(out3.cpu().numpy(), fused) and np.array_equal(vt3.cpu().numpy(), fused_v)
    solver.close()


def test_restart_pass_solves_only_the_failed_agents():
    """OMGX_ONLY_FAILED: a second call with other initial guesses touches only the agents that failed; BatchP2P's
    cold solve uses it with the straight-line guess bent sideways (Quadrotor class: phase-I stalls)."""
    import torch
    from omgtools.batch import BatchP2P
    problem, P = _setup(64, 'quadrotor_p2p')
    opts = dict(P.get('solver_options', {}), tol=1e-3, max_iter=300)
    plain = BatchP2P(problem, P, ops='hip', options=opts)
    assert plain.solve_cold(bends=()) == 0
    st0, x0 = plain.status.cpu().numpy().copy(), plain.x.cpu().numpy().copy()
    failed = np.nonzero(st0 != 0)[0]
    assert 0 < len(failed) < 16                                     # the class has its stragglers ...
    mpc = BatchP2P(problem, P, ops='hip', options=opts)
    passes = mpc.solve_cold()
    st1, x1 = mpc.status.cpu().numpy(), mpc.x.cpu().numpy()
    assert passes >= 1 and (st1 == 0).all()                         # ... and the bent guesses get all of them through
    ok = st0 == 0
    assert np.array_equal(x1[ok], x0[ok])                           # untouched: the same bits as without restarts
    assert np.abs(x1[failed] - x0[failed]).max() > 1e-3
    # the restarted solutions satisfy the optimality conditions of the reference's NLP
    from oracle.nlp_numpy import NumpyNLP
    from oracle.kkt_check import assert_kkt
    nlp = NumpyNLP(problem.father.template)
    lam = mpc.lam.cpu().numpy()
    for b in failed[:4]:
        assert_kkt(nlp, problem.father.template, P['p'][b], x1[b], lam[b], 1e-2, ('restart', b))
    # the same restarts as separate OMGX_ONLY_FAILED passes: same guesses, same bits
    sep = BatchP2P(problem, P, ops='hip', options=opts)
    passes_sep = sep.solve_cold(fused=False)
    assert passes_sep == passes
    assert np.array_equal(sep.x.cpu().numpy(), x1) and np.array_equal(sep.status.cpu().numpy(), st1)
    assert np.array_equal(sep.iters.cpu().numpy(), mpc.iters.cpu().numpy())
    assert np.array_equal(sep.lam.cpu().numpy(), lam)
    plain.solver.close()
    mpc.solver.close()
    sep.solver.close()


def test_launch_events_time_the_solve_kernel_itself():
    """omgx_batch_set_launch_events: the caller's events ride on the dispatch packet of the next solve kernel (one
    launch only); the handle's own timing (last_kernel_ms) uses the same mechanism and agrees."""
    import time
    import torch
    from omgtools.batch import BatchP2P
    problem, P = _setup(64)
    mpc = BatchP2P(problem, P, ops='hip', options=dict(tol=1e-3, max_iter=300))
    mpc.solver.set_timing(True)
    mpc.solve_cold(bends=())
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with pytest.raises(ValueError):
        mpc.solver.set_launch_events(a, b)                 # no handle before the first record
    a.record()
    b.record()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mpc.step(events=(a, b))
    torch.cuda.synchronize()
    wall_ms = (time.perf_counter() - t0) * 1e3
    ms = a.elapsed_time(b)
    assert 0.01 < ms < wall_ms
    assert (mpc.status == 0).all()
    # one shot: the next launch is timed by the handle's own pair again
    mpc.step()
    torch.cuda.synchronize()
    own = mpc.solver.last_kernel_ms()
    assert 0.01 < own < 50.0 and a.elapsed_time(b) == ms
    mpc.solver.close()


def test_launch_statistics_match_the_per_agent_outputs():
    """omgx_batch_set_stats: the counts the solve kernel adds up per launch equal those of the status / iters arrays."""
    import torch
    from omgtools.batch import BatchP2P
    problem, P = _setup(96)
    mpc = BatchP2P(problem, P, ops='hip', options=dict(tol=1e-3, max_iter=300))
    stats = torch.zeros((3, 4), dtype=torch.int64, device='cuda')
    mpc.solver.set_stats(stats)
    mpc.solve_cold(bends=())
    ref = [(int((mpc.status == 0).sum()), int(mpc.iters.sum()), int(mpc.iters.max()), 96)]
    for _ in range(3):                                   # the fourth launch wraps around into row 0
        mpc.step()
        ref.append((int((mpc.status == 0).sum()), int(mpc.iters.sum()), int(mpc.iters.max()), 96))
    got = stats.cpu().numpy()
    assert tuple(got[1]) == ref[1] and tuple(got[2]) == ref[2]
    assert got[0, 0] == ref[0][0] + ref[3][0] and got[0, 1] == ref[0][1] + ref[3][1]
    assert got[0, 2] == max(ref[0][2], ref[3][2]) and got[0, 3] == 192
    mpc.solver.set_stats(None)
    mpc.step()
    assert np.array_equal(stats.cpu().numpy(), got)
    mpc.solver.close()


def test_quadrotor_ode_prediction_matches_the_reference_statements():
    """Non-ideal prediction of the Quadrotor model on the device (`omgx_batch_predict_quadrotor`) against the reference's
    statements replayed in numpy / scipy: inputs from the plan's sampled derivatives (`vehicles/quadrotor.py:121-140`),
    `integrate_ode` = odeint on their linear interpolation (`vehicles/vehicle.py:412-423`) with the model's ode
    (`quadrotor.py:149-152`), spl0 from the integrated state, dspl0 / ddspl0 from the plan (`quadrotor.py:110-114`)."""
    import torch
    from scipy.integrate import odeint
    from scipy.interpolate import interp1d
    from omgtools import workloads
    from omgtools.backend import BatchSolver
    B = 6
    problem, P = workloads.quadrotor_p2p(B)
    tpl = problem.father.template
    veh = problem.vehicles[0]
    basis, T, g = veh.basis, float(problem.options['horizon_time']), 9.81
    L = len(basis)
    o_spl = tpl.entry_range(veh.label, 'splines_seg0', 'var')[0]
    rng = np.random.default_rng(5)
    x = P['x0'].copy()
    x[:, o_spl:o_spl + 2 * L] += 0.05 * rng.normal(size=(B, 2 * L))         # a plan with some curvature
    dev = torch.device('cuda', 0)
    solver = BatchSolver(tpl, B)
    f64 = dict(dtype=torch.float64, device=dev)
    xd, pd = torch.as_tensor(x, **f64), torch.as_tensor(P['p'].copy(), **f64)
    p_offs = [tpl.entry_range(veh.label, nm, 'par')[0] for nm in ('spl0', 'dspl0', 'ddspl0')]
    o_t = tpl.entry_range(problem.label, 't', 'par')[0]
    sample_time, update_time, t0 = 0.01, 0.1, 0.2
    n_sub = int(round(update_time / sample_time))
    tau = (t0 + update_time) / T
    state0 = np.zeros((B, 5))
    state_in = torch.as_tensor(state0, **f64)
    state_out = torch.zeros((B, 5), **f64)
    # the vehicles' current states: on the plan at t0 (position, velocity, pitch) plus a small disturbance
    der = [basis.eval_basis([t0 / T])[0]]
    for o in (1, 2, 3):
        db, Po = basis.derivative(o)
        der.append(db.eval_basis([t0 / T])[0] @ Po / T ** o)
    c = x[:, o_spl:o_spl + 2 * L].reshape(B, 2, L)
    pos, vel, acc = c @ der[0], c @ der[1], c @ der[2]
    state0 = np.c_[pos, vel, np.arctan2(acc[:, 0], acc[:, 1] + g)] + 1e-3 * rng.normal(size=(B, 5))
    state_in.copy_(torch.as_tensor(state0, **f64))
    try:
        solver.predict_quadrotor(xd, pd, o_spl, basis.degree, basis.knots, tau, 1.0 / T, p_offs, o_t, 0.3, state_in, state_out, n_sub,
                                 sample_time / T, g)
        torch.cuda.synchronize()
    finally:
        got_p, got_s = pd.cpu().numpy(), state_out.cpu().numpy()
        solver.close()
    times = t0 + sample_time * np.arange(n_sub + 1)
    rows = []
    for o in (0, 1, 2, 3):
        if o == 0:
            rows.append(basis.eval_basis(times / T))
        else:
            db, Po = basis.derivative(o)
            rows.append(db.eval_basis(times / T) @ Po / T ** o)
    for b in range(B):
        cx, cy = c[b]
        ddx, ddy, dddx, dddy = rows[2] @ cx, rows[2] @ cy, rows[3] @ cx, rows[3] @ cy
        u1 = np.sqrt(ddx ** 2 + (ddy + g) ** 2)
        u2 = (dddx * (ddy + g) - ddx * dddy) / ((ddy + g) ** 2 + ddx ** 2)
        inp = interp1d(times - t0, np.c_[u1, u2].T, kind='linear', bounds_error=False, fill_value=np.r_[u1[-1], u2[-1]])

        def ode(s, t):
            u = inp(t)
            return np.r_[s[2:4], u[0] * np.sin(s[4]), u[0] * np.cos(s[4]) - g, u[1]]
        ref = odeint(ode, state0[b], times - t0, rtol=1e-12, atol=1e-12)[-1]
        assert np.abs(got_s[b] - ref).max() < 1e-7, (b, np.abs(got_s[b] - ref).max())
        assert np.abs(got_p[b, p_offs[0]:p_offs[0] + 2] - ref[:2]).max() < 1e-7
        assert np.abs(got_p[b, p_offs[1]:p_offs[1] + 2] - np.r_[rows[1][-1] @ cx, rows[1][-1] @ cy]).max() < 1e-9
        assert np.abs(got_p[b, p_offs[2]:p_offs[2] + 2] - np.r_[rows[2][-1] @ cx, rows[2][-1] @ cy]).max() < 1e-8
        assert got_p[b, o_t] == 0.3


def test_transfer_moves_segments_between_device_and_pinned_host_memory():
    """`omgx_batch_transfer` (ABI 6): the host boundary without the copy engine -- one launch on the handle's stream moves up
    to six segments between device memory and pinned host memory, in either direction, ordered with the handle's other calls."""
    import torch
    from omgtools import workloads
    from omgtools.backend import BatchSolver, OmgxError
    problem, P = workloads.holonomic_p2p(4)
    solver = BatchSolver(problem.father.template, 4)
    solver.set_stream(torch.cuda.current_stream().cuda_stream)
    dev = torch.device('cuda', 0)
    g = torch.Generator(device='cpu').manual_seed(3)
    a_d = torch.randn((512, 164), dtype=torch.float64, generator=g).to(dev)
    b_d = torch.arange(514, dtype=torch.int32, device=dev)
    c_d = torch.randn(7, dtype=torch.float64, generator=g).to(dev)                      # an odd number of doubles
    a_h, b_h, c_h = (torch.zeros(t.shape, dtype=t.dtype).pin_memory() for t in (a_d, b_d, c_d))
    solver.transfer([(a_h, a_d), (b_h, b_d), (c_h, c_d)])                              # down: the kernel writes host memory
    solver.sync()
    assert torch.equal(a_h, a_d.cpu()) and torch.equal(b_h, b_d.cpu()) and torch.equal(c_h, c_d.cpu())
    a2, b2 = torch.zeros_like(a_d), torch.zeros_like(b_d)
    a_h.mul_(2.0)
    solver.transfer([(a2, a_h), (b2, b_h)])                                            # up: the kernel reads host memory
    a3 = a2 + 1.0                                                                       # (stream order: a torch kernel behind it sees the data)
    torch.cuda.synchronize()
    assert torch.equal(a2.cpu(), a_h) and torch.equal(b2, b_d) and torch.equal(a3.cpu(), a_h + 1.0)
    v = a_d.view(-1)[1:1 + 6 * 164]                                                     # 8-byte aligned only: the narrow path
    w_h = torch.zeros(6 * 164, dtype=torch.float64).pin_memory()
    solver.transfer([(w_h, v.contiguous().view(-1)[:]), ])
    solver.sync()
    assert torch.equal(w_h, v.cpu())
    with pytest.raises(ValueError):
        solver.transfer([(torch.zeros(4, dtype=torch.float64), c_d[:4].contiguous())])  # pageable host memory
    with pytest.raises(OmgxError):
        solver.transfer([(torch.zeros(3, dtype=torch.int32).pin_memory(), b_d[:3].contiguous())])   # not a multiple of 8 bytes
    solver.close()
