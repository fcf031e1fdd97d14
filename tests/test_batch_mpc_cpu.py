"""Receding-horizon protocol of BatchP2P on the host (oracle port as solver):
warm-started steps need far fewer iterations than the cold solve, a knot crossing
shifts the plan consistently, and the fleet makes progress to its targets."""
import numpy as np


def test_receding_horizon_protocol(cfg2_small):
    from omgtools.batch import BatchP2P, dual_shift_perm
    from oracle import port_binding
    problem, P = cfg2_small
    tpl = problem.father.template
    perm = dual_shift_perm(problem.father, extrapolate=False)
    # velocity rows (13, simple knots) shift by 1, vehicle-hyperplane rows (45, 4 per interval) by 4
    off = tpl.con_layout[(problem.vehicles[0].label, 'c_0_' + problem.vehicles[0].label)][0]
    assert list(perm[off:off + 3]) == [off + 1, off + 2, off + 3] and perm[off + 12] == -1
    off = tpl.con_layout[(problem.vehicles[0].label, 'c_8_' + problem.vehicles[0].label)][0]
    assert perm[off] == off + 4 and perm[off + 44] == -1
    # default: rows entering at the end of the horizon start from the last multiplier that has a predecessor
    perm_x = dual_shift_perm(problem.father)
    assert perm_x[off] == off + 4 and list(perm_x[off + 41:off + 45]) == [off + 44] * 4 and perm_x.min() >= 0
    mpc = BatchP2P(problem, P, ops=port_binding, options=dict(tol=1e-3, max_iter=300))
    mpc.solve_cold()
    ok = mpc.status == 0
    assert ok.sum() >= 5
    cold_iters = mpc.iters[ok].mean()
    lo = mpc.o_spl
    goal = P['p'][:, tpl.entry_range(problem.vehicles[0].label, 'poseT', 'par')[0]:][:, :2]
    d0 = np.linalg.norm(mpc.p[:, mpc.o_state0:mpc.o_state0 + 2] - goal, axis=1)
    warm_iters, crossings = [], 0
    for k in range(12):
        x_before = mpc.x.copy()
        crossed = mpc.step()
        crossings += crossed
        assert np.all(mpc.status[ok] == 0)
        warm_iters.append(mpc.iters[ok].mean())
        if not crossed:          # consecutive plans agree closely where they describe the future
            L = mpc.L                # (the first coefficients only shape the already-travelled piece)
            for a in range(2):
                sl = slice(lo + a * L + 2, lo + (a + 1) * L)
                assert np.abs(mpc.x[ok, sl] - x_before[ok, sl]).max() < 5e-2
    assert crossings == 1                                   # t = 1.0 passes the first knot (0.909)
    assert np.mean(warm_iters) < 0.5 * cold_iters
    d1 = np.linalg.norm(mpc.p[:, mpc.o_state0:mpc.o_state0 + 2] - goal, axis=1)
    assert np.all(d1[ok] < d0[ok] - 0.3)                    # 1.2 s of motion towards the goal


def test_moving_obstacles_and_multi_vehicle_guard():
    """Obstacle parameters are the values at the time of the solve (`environment/obstacle.py:142-155`): the
    loop advances x and v of moving obstacles between steps (constant acceleration, `obstacle.py:246-264`
    without bouncing); multi-vehicle problems are refused instead of predicted for vehicle 0 only."""
    import pytest
    import omgtools.backend as be
    from omgtools.batch import BatchP2P
    from omgtools.scenarios import holonomic_p2p
    from omgtools import Holonomic, Fleet, Environment, Obstacle, Circle, Square, Point2point
    from oracle import port_binding
    saved = be.create_nlp
    be.create_nlp = lambda tpl, opt, name='': (None, 0.)
    try:
        problem, P = holonomic_p2p(2, n_obs=2)
        veh1, veh2 = Holonomic(), Holonomic()
        for v, s in ((veh1, [-1., -1.]), (veh2, [-1., 1.])):
            v.set_initial_conditions(s); v.set_terminal_conditions([1., 0.])
        env = Environment(room={'shape': Square(5.)})
        prob2 = Point2point(Fleet([veh1, veh2]), env, options={'verbose': 0})
        prob2.init()
    finally:
        be.create_nlp = saved
    tpl = problem.father.template
    obs = problem.environment.obstacles[0]
    ox, ov = tpl.entry_range(obs.label, 'x', 'par')[0], tpl.entry_range(obs.label, 'v', 'par')[0]
    P['p'][:, ov:ov + 2] = [[0.05, -0.02], [-0.03, 0.04]]         # obstacle 0 moves, obstacle 1 stays
    mpc = BatchP2P(problem, P, ops=port_binding, options=dict(tol=1e-3, max_iter=5))
    assert len(mpc.obst) == 1
    x_start, v = P['p'][:, ox:ox + 2].copy(), P['p'][:, ov:ov + 2].copy()
    assert np.abs(v).max() > 0
    for k in range(3):
        mpc.step()
    assert np.abs(mpc.p[:, ox:ox + 2] - (x_start + 0.3 * v)).max() < 1e-12
    assert np.array_equal(mpc.p[:, ov:ov + 2], v)                 # no acceleration: velocity unchanged
    P2 = {'p': np.zeros((1, prob2.father.template.n_par)), 'x0': np.zeros((1, prob2.father.template.n_var))}
    with pytest.raises(NotImplementedError):
        BatchP2P(prob2, P2, ops=port_binding)


def test_pool_step_glue_in_c_equals_the_numpy_glue():
    """bench.py's CPU baseline runs the step glue per agent inside the pinned workers of oracle/port
    (`omgx_port_pool_solve`): same iterates as the numpy statements of BatchP2P.step, knot crossing included."""
    import omgtools.backend as be
    from omgtools.scenarios import holonomic_p2p
    from omgtools.batch import BatchP2P
    from oracle import port_binding
    saved = be.create_nlp
    be.create_nlp = lambda tpl, opt, name='': (None, 0.)
    try:
        problem, P = holonomic_p2p(6)
    finally:
        be.create_nlp = saved
    opts = dict(tol=1e-3, max_iter=300)
    a = BatchP2P(problem, P, ops=port_binding, options=opts)
    b = BatchP2P(problem, P, ops=port_binding, options=opts)
    b.pool = port_binding.PortPool(problem.father.template, n_threads=2)
    a.solve_cold()
    b.solve_cold()
    assert np.array_equal(a.x, b.x)
    crossings = 0
    for _ in range(11):
        ca, cb = a.step(), b.step()
        assert ca == cb
        crossings += int(ca)
        assert np.array_equal(a.status, b.status) and np.array_equal(a.iters, b.iters)
        assert np.abs(a.x - b.x).max() < 1e-8 and np.abs(a.p - b.p).max() < 1e-11
    assert crossings >= 1
    b.pool.close()


def test_sub_batch_count_of_the_product_path():
    """Three sub-batches, with or without a torch.distributed process group: the sub-batch streams live in a hardware-queue pool
    of their own (`omgtools.batch.sub_batch_streams`, high-priority streams), so the count no longer depends on what else the
    process created (rounds 5 / 6 picked two or four under a process group; measured now: 2.20 M solves/s with three either way)."""
    from omgtools.batch import product_path_streams, PRODUCT_PATH_STREAMS
    assert PRODUCT_PATH_STREAMS == 3
    assert product_path_streams() == product_path_streams(process_group=True, hw_queues=4) == product_path_streams(process_group=True, hw_queues=8) == 3


def test_arrived_is_the_reference_stop_criterion():
    """`BatchP2P.arrived` = `vehicles/holonomic.py:145-151` on the predicted state: nobody has arrived after the cold solve and a few
    updates; an agent whose predicted state is put on its target pose at rest has."""
    from omgtools import workloads
    from omgtools.batch import BatchP2P
    from oracle import port_binding
    problem, P = workloads.holonomic_p2p(6)
    m = BatchP2P(problem, P, ops=port_binding, options=dict(tol=1e-3, max_iter=300))
    m.solve_cold()
    for _ in range(3):
        m.step()
    assert not m.arrived().any()
    tpl = m.tpl
    o_pose = tpl.entry_range(m.veh.label, 'poseT', 'par')[0]
    m.p[2, m.o_state0:m.o_state0 + 2] = m.p[2, o_pose:o_pose + 2] + np.array([6e-4, 0.0])
    m.p[2, m.o_input0:m.o_input0 + 2] = np.array([0.0, -8e-4])
    m.p[4, m.o_state0:m.o_state0 + 2] = m.p[4, o_pose:o_pose + 2]
    m.p[4, m.o_input0:m.o_input0 + 2] = np.array([2e-3, 0.0])              # on the target but still moving
    a = m.arrived()
    assert a.tolist() == [False, False, True, False, False, False]
    assert m.arrived(stop_tol=5e-4).sum() == 0


def test_stop_at_arrival_ends_an_agents_loop_like_the_reference():
    """`BatchP2P.stop_at_arrival` (host loop; the device loop's rule is the solve kernel's, tests/test_gpu_batch_mpc.py): the reference's
    `Simulator.run` leaves a vehicle's loop at the first update for which `stop_criterium` holds (`execution/simulator.py:39-62`).  With a
    wide tolerance the six vehicles arrive at different updates: until then an agent is solved exactly as without the rule, from then on
    it keeps its plan, its multipliers and its status and `iters` reads 0 -- also when the criterion stops holding later."""
    from omgtools import workloads
    from omgtools.batch import BatchP2P
    from oracle import port_binding
    problem, P = workloads.holonomic_p2p(6)
    opts = dict(tol=1e-3, max_iter=300)
    free, ruled = (BatchP2P(problem, P, ops=port_binding, options=opts) for _ in range(2))
    ruled.stop_at_arrival(stop_tol=2.5)            # (|state - poseT| <= 2.5 m and |input| <= 2.5 m/s: under way, some metres out)
    for m in (free, ruled):
        m.solve_cold()
    assert ruled.under_way.all() and np.array_equal(free.x, ruled.x)
    ended_at = np.full(6, -1)
    frozen = {}
    for k in range(40):
        free.step(); ruled.step()
        crit = free.arrived(2.5)                     # (the free loop's p equals the ruled one's for every agent still under way)
        for b in range(6):
            if ended_at[b] < 0 and crit[b]:
                ended_at[b] = k
                frozen[b] = (ruled.x[b].copy(), ruled.lam[b].copy())
        run = ended_at < 0
        assert np.array_equal(ruled.under_way, run), k
        assert np.array_equal(ruled.x[run], free.x[run]) and np.array_equal(ruled.lam[run], free.lam[run]) and np.array_equal(ruled.iters[run], free.iters[run])
        assert (ruled.iters[~run] == 0).all() and (ruled.status[~run] == 0).all()
    assert (ended_at >= 0).sum() >= 3 and len(set(ended_at[ended_at >= 0])) >= 2 and (ended_at != 0).all()
    for b, (xb, lb) in frozen.items():             # shifted over the knots it crossed since, never solved again
        assert ruled.iters[b] == 0
    ruled.stop_at_arrival(on=False)
    assert ruled.under_way is None


def test_the_knot_clock_is_consistent_where_the_reference_statement_is_not():
    """`omgtools.splines.since_knot`: the reference's `np.round(t, 6) % knot_time` (`problems/point2point.py:177`) everywhere but
    where it contradicts the reference's own crossing test (`point2point.py:190-193`): T = 10 s, 11 knot intervals, t = 10.0 -- the
    crossing test counts the 11th knot, the remainder reads 0.909 s.  The receding-horizon loop crosses that instant with consistent
    initial conditions: no agent needs more than a handful of iterations there (62 with the verbatim statement)."""
    from omgtools.splines import since_knot
    kt = (int(10.0 * 1000.) / 11) / 1000.                                     # (`point2point.py:134`: 0.9090909090909092, one ulp above 10 / 11)
    assert abs(float(np.round(10.0, 6) % kt) - kt) < 1e-12                  # the reference statement at t = 10.0: one interval off
    assert int(np.round(9.9 / kt, 6)) < int(np.round(10.0 / kt, 6))         # ... while its crossing test shifts the plan
    assert since_knot(10.0, kt) == 0.0 and since_knot(20.0, kt) == 0.0
    for t in (0.0, 0.1, 0.5, 0.9, 1.0, 4.5, 9.9, 10.1, 17.3):                 # elsewhere: the reference's statement, bit for bit
        assert since_knot(t, kt) == float(np.round(t, 6) % kt)
    assert since_knot(1.0, 0.5) == 0.0 and since_knot(0.75, 0.5) == 0.25
    from omgtools import workloads
    from omgtools.batch import BatchP2P
    from oracle import port_binding
    problem, P = workloads.holonomic_p2p(1024)
    sel = np.array([159, 35, 0, 1])                                           # (159: the 62-iteration agent of the verbatim clock)
    m = BatchP2P(problem, dict(P, p=P['p'][sel], x0=P['x0'][sel]), ops=port_binding, options=dict(tol=1e-3, max_iter=300))
    m.solve_cold(bends=())
    m.time = 9.0                                                              # (the clock only: ten updates up to and across t = 10.0)
    worst = 0
    for k in range(12):
        crossed = m.step()
        if abs(m.time - 10.0) < 1e-9:
            assert crossed and m.p[0, m.o_t] == 0.0
        worst = max(worst, int(m.iters.max()))
    assert (m.status == 0).all() and worst <= 12
