"""sample_kernel / shift_kernel against the oracle (numpy spline algebra pinned to
the reference in tests/test_golden_splines.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _solver(cfg2_small):
    from omgtools.backend import BatchSolver
    problem, P = cfg2_small
    return problem, P, BatchSolver(problem.father.template, 8)


def test_sample_matches_numpy(cfg2_small):
    from omgtools.splines import BSpline
    problem, P, solver = _solver(cfg2_small)
    tpl = problem.father.template
    veh = problem.vehicles[0]
    rng = np.random.default_rng(3)
    x = rng.normal(size=(8, tpl.n_var))
    lo, hi = tpl.entry_range(veh.label, 'splines_seg0', 'var')
    L = len(veh.basis)
    T = 10.0
    knots = veh.basis.knots * T                        # real-time axis as in Vehicle.store
    t0 = rng.uniform(0., 0.9, size=8)
    n_samp, dt = 1001, 0.009
    out = solver.sample(x, lo, 2, veh.degree, knots, 3, t0, dt, n_samp)
    assert out.shape == (8, 3, 2, n_samp)
    for b in range(8):
        tau = t0[b] + dt * np.arange(n_samp)
        for k in range(2):
            s = BSpline(veh.basis.scale(T), x[b, lo + k * L: lo + (k + 1) * L])
            for o in range(3):
                ref = s.derivative(o)(tau)
                assert np.abs(out[b, o, k] - ref).max() <= 1e-10 * (1 + np.abs(ref).max())
    out32 = solver.sample(x, lo, 2, veh.degree, knots, 3, t0, dt, n_samp, as_f32=True)
    assert out32.dtype == np.float32 and np.abs(out32 - out).max() <= 1e-5 * (1 + np.abs(out).max())
    # end point convention: tau == last knot is inside the last span
    t_end = np.full(8, T)
    end = solver.sample(x, lo, 2, veh.degree, knots, 1, t_end, 0.0, 1)
    assert np.abs(end[:, 0, :, 0] - x[:, [lo + L - 1, lo + 2 * L - 1]]).max() < 1e-12
    solver.close()


def test_sample_all_orders_nonuniform_knots_device_pointers(cfg2_small):
    """Degree 4, every derivative order, non-uniform breakpoints, device
    pointers (asynchronous on the handle's stream): the per-span power series equal scipy-free
    de Boor evaluation (`omgtools.splines`, pinned to the reference in tests/test_golden_splines.py)."""
    import torch
    from omgtools.splines import BSpline, BSplineBasis
    dev = torch.device('cuda', 0)
    torch.zeros(1, device=dev)                           # torch initialises the HIP runtime first (as in BatchP2P)
    problem, P, solver = _solver(cfg2_small)
    tpl = problem.father.template
    rng = np.random.default_rng(11)
    deg = 4
    brk = np.r_[0., np.sort(rng.uniform(0.05, 0.95, size=6)), 1.]
    knots = np.r_[np.zeros(deg), brk, np.ones(deg)]
    basis = BSplineBasis(knots, deg)
    L, n_spl, n_der, n_samp = len(basis), 3, deg + 1, 777
    assert n_spl * L <= tpl.n_var
    x = rng.normal(size=(8, tpl.n_var))
    t0 = rng.uniform(0., 0.2, size=8)
    dt = 0.8 / (n_samp - 1)
    xd = torch.as_tensor(x, device=dev)
    td = torch.as_tensor(t0, device=dev)
    out = torch.empty((8, n_der, n_spl, n_samp), dtype=torch.float64, device=dev)
    solver.sample(xd, 5, n_spl, deg, knots, n_der, td, dt, n_samp, out=out, device=True)
    solver.sync()
    out = out.cpu().numpy()
    for b in range(8):
        tau = t0[b] + dt * np.arange(n_samp)
        for k in range(n_spl):
            s = BSpline(basis, x[b, 5 + k * L: 5 + (k + 1) * L])
            for o in range(n_der):
                ref = s.derivative(o)(tau)
                assert np.abs(out[b, o, k] - ref).max() <= 1e-9 * (1 + np.abs(ref).max()), (b, k, o)
    solver.close()


def test_shift_matches_matrix(cfg2_small):
    from omgtools.splines import shiftoverknot_T, BSplineBasis
    problem, P, solver = _solver(cfg2_small)
    tpl = problem.father.template
    father = problem.father
    rng = np.random.default_rng(4)
    x = rng.normal(size=(8, tpl.n_var))
    mask = np.array([1, 0, 1, 1, 0, 0, 1, 0], dtype=np.uint8)
    entries, tmats, off = [], [], 0
    want = x.copy()
    for label, name, spl in father.shifted_entries():
        lo, rows, cols = tpl.var_layout[(label, name)]
        Tm = shiftoverknot_T(spl['basis'])
        entries.append([lo, rows, cols, off])
        tmats.append(Tm.reshape(-1))
        off += Tm.size
        for b in np.nonzero(mask)[0]:
            blk = x[b, lo:lo + rows * cols].reshape((rows, cols), order='F')
            want[b, lo:lo + rows * cols] = (Tm @ blk).reshape(-1, order='F')
    assert len(entries) == 7                      # splines_seg0 + (a, b) x 3 obstacles; g* are not shifted
    got = x.copy()
    solver.shift(got, mask, np.array(entries), np.concatenate(tmats))
    assert np.abs(got - want).max() < 1e-13
    solver.close()


def test_shift_table_sets_are_cached_and_evicted(cfg2_small):
    """The handle keeps the table sets it has seen on the device (16, found again by content; the least recently used one goes):
    twenty different sets, then the first ones again -- every shift equals the matrix product."""
    problem, P, solver = _solver(cfg2_small)
    tpl = problem.father.template
    rng = np.random.default_rng(11)
    lo, rows, cols = tpl.var_layout[(problem.vehicles[0].label, 'splines_seg0')]
    mask = np.ones(8, dtype=np.uint8)
    sets = [rng.normal(size=(rows, rows)) for _ in range(20)]
    for k in list(range(20)) + [0, 1, 19, 5]:
        Tm = sets[k]
        x = rng.normal(size=(8, tpl.n_var))
        want = x.copy()
        for b in range(8):
            blk = x[b, lo:lo + rows * cols].reshape((rows, cols), order='F')
            want[b, lo:lo + rows * cols] = (Tm @ blk).reshape(-1, order='F')
        got = x.copy()
        solver.shift(got, mask, np.array([[lo, rows, cols, 0]]), Tm.reshape(-1))
        assert np.abs(got - want).max() < 1e-12, k
    solver.close()


def test_predict_matches_basis_evaluation(cfg2_small):
    """`omgx_batch_predict` (ideal prediction, `vehicles/vehicle.py:323-326`): state0 = spline(tau),
    input0 = spline'(tau)/T, t written into p -- against the host basis matrices."""
    import torch
    from omgtools.backend import BatchSolver
    problem, P = cfg2_small
    tpl = problem.father.template
    veh = problem.vehicles[0]
    basis, L, nd = veh.basis, len(veh.basis), veh.n_dim
    B = P['p'].shape[0]
    rng = np.random.default_rng(5)
    x = rng.standard_normal((B, tpl.n_var))
    o_spl = tpl.entry_range(veh.label, 'splines_seg0', 'var')[0]
    o_s0 = tpl.entry_range(veh.label, 'state0', 'par')[0]
    o_i0 = tpl.entry_range(veh.label, 'input0', 'par')[0]
    o_t = tpl.entry_range(problem.label, 't', 'par')[0]
    solver = BatchSolver(tpl, B)
    dev = torch.device('cuda', 0)
    solver.set_stream(torch.cuda.current_stream().cuda_stream)      # same stream as the torch copies
    for tau in (0.004, 1. / 11. - 1e-3, 0.5, 0.97):
        xd = torch.as_tensor(x, dtype=torch.float64, device=dev)
        pd = torch.as_tensor(P['p'], dtype=torch.float64, device=dev).clone()
        solver.predict(xd, pd, o_spl, nd, basis.degree, basis.knots, tau, 1. / 10., o_s0, o_i0, o_t, 0.123)
        solver.sync()
        p = pd.cpu().numpy()
        E = basis.eval_basis([tau])[0]
        dbasis, P1 = basis.derivative(1)
        Ed = dbasis.eval_basis([tau])[0] @ P1 / 10.
        c = x[:, o_spl:o_spl + nd * L].reshape(B, nd, L)
        assert np.abs(p[:, o_s0:o_s0 + nd] - c @ E).max() < 1e-12
        assert np.abs(p[:, o_i0:o_i0 + nd] - c @ Ed).max() < 1e-11
        assert np.all(p[:, o_t] == 0.123)
        keep = np.ones(tpl.n_par, bool); keep[o_s0:o_s0 + nd] = False; keep[o_i0:o_i0 + nd] = False; keep[o_t] = False
        assert np.array_equal(p[:, keep], P['p'][:, keep])
    solver.close()


def test_order_by_iters_is_a_bucketed_permutation(cfg2_small):
    """`omgx_batch_order_by_iters`: a permutation of the agents, iteration counts (clamped to 15: sixteen classes, each split
    four ways by the inertia correction the agent carries -- none here: a fresh handle) non-increasing along it; installing it
    as the launch order does not change any result."""
    import torch
    from omgtools.backend import BatchSolver
    problem, P = cfg2_small
    tpl = problem.father.template
    B = 64
    p = np.tile(P['p'], (B // 8, 1)); x0 = np.tile(P['x0'], (B // 8, 1))
    solver = BatchSolver(tpl, B)
    dev = torch.device('cuda', 0)
    solver.set_stream(torch.cuda.current_stream().cuda_stream)
    rng = np.random.default_rng(3)
    iters = torch.as_tensor(rng.integers(0, 40, size=B).astype(np.int32), device=dev)
    order = torch.zeros(B, dtype=torch.int32, device=dev)
    solver.order_by_iters(iters, order)
    solver.sync()
    o = order.cpu().numpy(); it = np.minimum(iters.cpu().numpy(), 15)
    assert sorted(o.tolist()) == list(range(B))
    assert np.all(np.diff(it[o]) <= 0)
    res = solver.solve(p, x0)                       # launched in that order
    solver.set_order(None)
    ref = solver.solve(p, x0)
    assert np.array_equal(res['status'], ref['status']) and np.ptp(res['iters'] - ref['iters']) <= 1
    assert np.abs(res['x'] - ref['x']).max() < 1e-6
    solver.close()
