"""`omgx_batch_rollout` (K receding-horizon steps of every agent in one launch) against the stepwise loop `BatchP2P.step`
(`execution/deployer.py:43-79`: one update per call): per agent the same statements in the same order, so plans, multipliers,
parameters, statuses and iteration counts of every step must be THE SAME BITS -- across a knot crossing, with moving
obstacles, and when the batch is not a multiple of the resident workgroups."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pair(n, mutate=None, **kw):
    import torch
    from omgtools import workloads
    from omgtools.batch import BatchP2P
    out = []
    for _ in range(2):
        problem, P = workloads.holonomic_p2p(n, **kw)
        if mutate is not None:
            mutate(problem, P)
        mpc = BatchP2P(problem, P, ops='hip', device=torch.device('cuda', 0), options=dict(tol=1e-3, max_iter=300))
        mpc.solve_cold(bends=())
        out.append(mpc)
    return out


@pytest.mark.parametrize('n', [40, 1024])
def test_rollout_equals_the_stepwise_loop_bit_for_bit(n):
    import torch
    a, b = _pair(n)
    K = 12                                               # update_time 0.1 s, knot_time 1 s: one crossing inside
    it_log = torch.zeros((K, n), dtype=torch.int32, device=a.dev)
    st_log = torch.full((K, n), -1, dtype=torch.int32, device=a.dev)
    crossings = a.rollout(K, iters_log=it_log, status_log=st_log)
    it_ref, st_ref, crossed_ref = [], [], 0
    for _ in range(K):
        crossed_ref += int(b.step())
        it_ref.append(b.iters.clone()); st_ref.append(b.status.clone())
    torch.cuda.synchronize()
    assert crossings == crossed_ref == 1
    assert abs(a.time - b.time) < 1e-12
    assert torch.equal(it_log, torch.stack(it_ref)) and torch.equal(st_log, torch.stack(st_ref))
    assert (st_log == 0).all()
    for name in ('x', 'lam', 'p', 'status', 'iters'):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    # and the loop goes on from there either way
    a.step(); b.rollout(1)
    torch.cuda.synchronize()
    assert torch.equal(a.x, b.x) and torch.equal(a.lam, b.lam)
    a.solver.close(); b.solver.close()


def test_rollout_fills_the_per_step_statistics():
    import torch
    (a, b) = _pair(64)
    K = 5
    stats = torch.zeros((K + 2, 4), dtype=torch.int64, device=a.dev)
    a.solver.set_stats(stats)
    a.rollout(K)
    a.step()
    a.solver.set_stats(None)
    torch.cuda.synchronize()
    s = stats.cpu().numpy()
    assert (s[:K + 1, 3] == 64).all() and (s[:K + 1, 0] == 64).all() and s[K + 1].sum() == 0
    assert (s[:K + 1, 2] >= 1).all() and (s[:K + 1, 1] >= 64).all()
    a.solver.close(); b.solver.close()


def test_rollout_with_moving_obstacles():
    """Obstacles with a velocity and an acceleration are advanced inside the launch exactly as the tensor statements of `step` do
    (each product and sum rounded on its own)."""
    import torch

    def drift(problem, P):
        tpl = problem.father.template
        rng = np.random.default_rng(5)
        for obs in problem.environment.obstacles:
            ov, oa = (tpl.entry_range(obs.label, nm, 'par') for nm in ('v', 'a'))
            P['p'][:, ov[0]:ov[1]] = rng.uniform(-0.03, 0.03, size=(len(P['p']), ov[1] - ov[0]))
            P['p'][:, oa[0]:oa[1]] = rng.uniform(-0.01, 0.01, size=(len(P['p']), oa[1] - oa[0]))
    a, b = _pair(32, mutate=drift)
    assert len(a.obst) == 3
    a.rollout(6)
    for _ in range(6):
        b.step()
    torch.cuda.synchronize()
    for name in ('x', 'lam', 'p', 'status', 'iters'):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    a.solver.close(); b.solver.close()


def test_rollout_refuses_what_it_cannot_do():
    """A crossing without shift tables, and a template class without a rollout kernel (spill mode), are errors -- not silent
    fallbacks."""
    import torch
    from omgtools import workloads
    from omgtools.backend import OmgxError
    from omgtools.batch import BatchP2P
    (a, b) = _pair(8)
    with pytest.raises(OmgxError):
        a.solver.rollout(a.p, a.x, a.lb, a.ub, a.lam, a.status, a.iters, [0.05], [0.0], [1], a.o_spl, a.n_spl, a.basis.degree,
                         a.basis.knots, 1.0 / a.T, a.p_offs, a.o_t)
    a.solver.close(); b.solver.close()
    problem, P = workloads.quadrotor_p2p(4)
    q = BatchP2P(problem, P, ops='hip', device=torch.device('cuda', 0), options=dict(P['solver_options'], tol=1e-3, max_iter=300))
    q.solve_cold(bends=())
    with pytest.raises(OmgxError):
        q.rollout(2)
    q.step()                                             # the stepwise loop is what such a class uses
    torch.cuda.synchronize()
    q.solver.close()


def test_two_streams_give_the_same_plans():
    """`StreamedP2P`: two sub-batches on two HIP streams -- per agent the launches of `BatchP2P`, so the same bits."""
    import torch
    from omgtools import workloads
    from omgtools.batch import BatchP2P, StreamedP2P
    n = 64
    problem, P = workloads.holonomic_p2p(n)
    one = BatchP2P(problem, P, ops='hip', device=torch.device('cuda', 0), options=dict(tol=1e-3, max_iter=300))
    problem2, P2 = workloads.holonomic_p2p(n)
    two = StreamedP2P(problem2, P2, n_streams=2, device=torch.device('cuda', 0), options=dict(tol=1e-3, max_iter=300))
    one.solve_cold(bends=()); two.solve_cold(bends=())
    crossed = 0
    for _ in range(11):
        c1, c2 = one.step(), two.step()
        assert bool(c1) == bool(c2)
        crossed += int(bool(c1))
    torch.cuda.synchronize()
    assert crossed == 1
    for name in ('x', 'lam', 'p', 'status', 'iters'):
        assert torch.equal(getattr(one, name), two.gather(name)), name
    one.solver.close(); two.close()


def test_rollout_writes_the_stored_trajectories():
    """`omgx_batch_set_store` holds for the rollout launch too: after K steps the sampled state / input / dinput of every agent are
    those the stepwise loop leaves (the same routine in both epilogues: the same bits)."""
    import torch
    a, b = _pair(16)
    outs = []
    for m in (a, b):
        veh, tpl = m.veh, m.tpl
        f64 = dict(dtype=torch.float64, device=m.dev)
        out, vt, t0 = torch.zeros((16, 3, veh.n_dim, 101), **f64), torch.zeros((16, 101), **f64), torch.zeros(16, **f64)
        m.solver.set_store(out, vt, t0, m.o_spl, veh.n_dim, m.basis.degree, m.basis.knots, 3, 101, 0.1 / m.T, 1.0 / m.T)
        outs.append((out, vt, t0))
    a.rollout(5)
    for _ in range(5):
        b.step()
    torch.cuda.synchronize()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and float(outs[0][0].abs().max()) > 0.1
    for m in (a, b):
        m.solver.set_store(None)
        m.solver.close()


def test_back_to_back_rollouts_on_a_caller_stream():
    """Round-4 advisor: the rollout's step table, its argument block and the multiplier map were written with null-stream
    copies; on a non-blocking caller stream (`set_stream`) a second rollout issued without a sync rewrote them under the first
    one's persistent kernel.  The copies are ordered on the handle's stream now: two rollouts of different step tables issued
    back to back on a side stream (different tau / t_rel / crossed per call) equal the stepwise loop bit for bit."""
    import torch
    from omgtools import workloads
    from omgtools.batch import BatchP2P
    n = 600                                              # (more agents than resident workgroups: the first rollout is still under way when the second is issued)
    side = torch.cuda.Stream()
    objs = []
    for st in (side, torch.cuda.current_stream()):
        problem, P = workloads.holonomic_p2p(n)
        with torch.cuda.stream(st):
            m = BatchP2P(problem, P, ops='hip', device=torch.device('cuda', 0), options=dict(tol=1e-3, max_iter=300))
            m.solve_cold(bends=())
        objs.append(m)
    a, b = objs
    with torch.cuda.stream(side):
        c1 = a.rollout(7)                                # steps 1-7
        c2 = a.rollout(6)                                # steps 8-13: crosses the knot at t = 1 s, issued while the first is running
    crossed = sum(int(b.step()) for _ in range(13))
    side.synchronize(); torch.cuda.synchronize()
    assert c1 + c2 == crossed == 1
    for name in ('x', 'lam', 'p', 'status', 'iters'):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    a.solver.close(); b.solver.close()


def test_the_per_step_product_path_picks_two_half_launches_for_two_rounds_of_workgroups():
    """`receding_horizon_batch`: one handle for a batch below two rounds of resident workgroups, three stream-ordered sub-batch
    launches per step (384 + 384 + 256 agents: sizes in quarters of the resident workgroups, `split_bounds`) at 1024 agents (512 resident workgroups) -- with the plans of the single handle, bit
    for bit, also across the crossing."""
    import torch
    from omgtools import workloads
    from omgtools.batch import BatchP2P, StreamedP2P, receding_horizon_batch, PRODUCT_PATH_STREAMS
    dev = torch.device('cuda', 0)
    opts = dict(tol=1e-3, max_iter=300)
    problem, P = workloads.holonomic_p2p(64)
    small = receding_horizon_batch(problem, P, device=dev, options=opts)
    assert isinstance(small, BatchP2P)
    small.solver.close()
    n = 1024
    problem, P = workloads.holonomic_p2p(n)
    rh = receding_horizon_batch(problem, P, device=dev, options=opts)
    assert isinstance(rh, StreamedP2P) and len(rh.parts) == PRODUCT_PATH_STREAMS == 3 and rh.B == n
    assert [m.B for m in rh.parts] == [384, 384, 256]
    problem1, P1 = workloads.holonomic_p2p(n)
    one = receding_horizon_batch(problem1, P1, device=dev, n_streams=1, options=opts)
    assert isinstance(one, BatchP2P)
    rh.solve_cold(bends=()); one.solve_cold(bends=())
    for _ in range(11):
        assert bool(rh.step()) == bool(one.step())
    for name in ('x', 'lam', 'p', 'status', 'iters'):
        assert torch.equal(getattr(rh, name), getattr(one, name)), name       # (the properties join the streams)
    assert abs(rh.time - one.time) < 1e-12
    rh.close(); one.solver.close()


def test_rollout_applies_the_stop_rule_like_the_stepwise_loop():
    """`omgx_batch_rollout` with `omgx_batch_set_stop` on (`BatchP2P.stop_at_arrival`): inside the one launch an agent's loop ends at the
    step its state meets the reference's stop criterion -- the step at which the solve kernel of the stepwise loop stops solving it.
    Same flags, the same iteration counts step by step (0 from the stop on), the same bits for the agents still under way at the end;
    an agent that stopped keeps the plan it had at that step (the stepwise loop goes on shifting it at later knot crossings)."""
    import torch
    from omgtools import workloads
    from omgtools.batch import BatchP2P
    B, K = 24, 40
    problem, P = workloads.holonomic_p2p(B)
    dev = torch.device('cuda', 0)
    opts = dict(tol=1e-3, max_iter=300)
    step, roll = (BatchP2P(problem, P, ops='hip', device=dev, options=opts) for _ in range(2))
    try:
        for m in (step, roll):
            m.stop_at_arrival(stop_tol=2.5)
            m.solve_cold(bends=())
        it_hist = np.zeros((K, B), dtype=np.int32)
        x_at_stop = {}
        was = np.ones(B, dtype=bool)
        for k in range(K):
            step.step()
            it_hist[k] = step.host('iters')
            now = step.host('under_way') != 0
            for b in np.flatnonzero(was & ~now):
                x_at_stop[int(b)] = step.host('x')[b].copy()
            was = now
        iters_log = torch.zeros((K, B), dtype=torch.int32, device=dev)
        status_log = torch.full((K, B), -1, dtype=torch.int32, device=dev)
        stats = torch.zeros((K, 4), dtype=torch.int64, device=dev)
        roll.solver.set_stats(stats)
        roll.rollout(K, iters_log=iters_log, status_log=status_log)
        torch.cuda.synchronize()
        roll.solver.set_stats(None)
        under = roll.host('under_way') != 0
        assert np.array_equal(under, was) and 0 < under.sum() < B and len(x_at_stop) == int((~under).sum())
        assert np.array_equal(iters_log.cpu().numpy(), it_hist) and (status_log.cpu().numpy() == 0).all()
        assert np.array_equal(stats[:, 3].cpu().numpy(), (it_hist > 0).sum(axis=1)) and np.array_equal(stats[:, 1].cpu().numpy(), it_hist.sum(axis=1))
        for name in ('x', 'lam', 'p'):
            assert np.array_equal(roll.host(name)[under], step.host(name)[under]), name
        for b, xb in x_at_stop.items():
            assert np.array_equal(roll.host('x')[b], xb), b
    finally:
        step.solver.close(); roll.solver.close()
