"""Device-resident receding-horizon loop (BatchP2P on the HIP path) against the
same protocol on host arrays with the oracle port (replay shaped like the
reference's `export/tests/point2point/test.cpp`: N MPC steps, two-sided
tolerance on the trajectories)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_mpc_replay_matches_port(cfg2_small):
    from omgtools.batch import BatchP2P
    from oracle import port_binding
    problem, P = cfg2_small
    tpl = problem.father.template
    from oracle.nlp_numpy import NumpyNLP
    nlp = NumpyNLP(tpl)
    opts = dict(tol=1e-5, max_iter=300)
    gpu = BatchP2P(problem, P, ops='hip', options=opts, max_iter_step=300)
    cpu = BatchP2P(problem, P, ops=port_binding, options=opts, max_iter_step=300)
    gpu.solve_cold()
    cpu.solve_cold()
    ok = (gpu.host('status') == 0) & (cpu.status == 0)
    assert ok.sum() >= 4
    crossings = 0
    for k in range(12):
        cg = gpu.step()
        cc = cpu.step()
        assert cg == cc
        crossings += cg
        sg, sc = gpu.host('status'), cpu.status
        assert np.array_equal(sg, sc)
        ok = ok & (sg == 0)
        assert ok.sum() >= 3
        pg, pc = gpu.host('p'), cpu.p
        # initial conditions handed to the next solve: positions to 2e-4, velocities to 2e-3
        # (iterates agree to rounding; the stopping points of two runs differ within the tolerance
        # and flat directions of this L1-type objective amplify that; the reference's own replay test
        # accepts 1e-4 relative, export/tests/point2point/test.cpp:131,138)
        s0, i0 = gpu.o_state0, gpu.o_input0
        assert np.abs(pg[ok, s0:s0 + 2] - pc[ok, s0:s0 + 2]).max() < 2e-4
        assert np.abs(pg[ok, i0:i0 + 2] - pc[ok, i0:i0 + 2]).max() < 2e-3
        # the plans themselves: equal objective, both feasible (the optimal face of this L1-type
        # objective is flat in places, so coefficients may differ by millimetres at equal cost)
        xg = gpu.host('x')
        for b in np.nonzero(ok)[0][:3]:
            # (each plan with the parameters of its own run: the initial-condition rows move with the predicted state,
            # which the two runs agree on to 2e-4 only)
            fg_, gg_ = nlp.fg(xg[b], nlp.term_coefs(pg[b]))
            fc_, gc_ = nlp.fg(cpu.x[b], nlp.term_coefs(pc[b]))
            assert abs(fg_ - fc_) < 1e-4 * (1 + abs(fc_))
            assert (gg_ - tpl.ub).max() < 1e-5 and (tpl.lb - gg_).max() < 1e-5
            assert (gc_ - tpl.ub).max() < 1e-5 and (tpl.lb - gc_).max() < 1e-5
    assert crossings == 1
    gpu.solver.close()


def test_arrived_on_the_device_is_the_reference_stop_criterion(cfg2_small):
    """`BatchP2P.arrived` (device loop) against the statement of `vehicles/holonomic.py:145-151` on the downloaded parameters, along a run
    long enough for the vehicles to reach their targets (T = 10 s, update time 0.1 s)."""
    import torch
    from omgtools import workloads
    from omgtools.batch import BatchP2P
    problem, P = workloads.holonomic_p2p(16)
    m = BatchP2P(problem, P, ops='hip', device=torch.device('cuda', 0), options=dict(tol=1e-3, max_iter=300))
    try:
        m.solve_cold()
        tpl = m.tpl
        o_pose = tpl.entry_range(m.veh.label, 'poseT', 'par')[0]
        seen = 0
        for k in range(125):
            m.step()
            if k % 10 == 4 or k > 100:
                p = m.host('p')
                ref = (np.linalg.norm(p[:, m.o_state0:m.o_state0 + 2] - p[:, o_pose:o_pose + 2], axis=1) <= 1e-3) & \
                      (np.linalg.norm(p[:, m.o_input0:m.o_input0 + 2], axis=1) <= 1e-3)
                got = m.arrived().cpu().numpy()
                assert np.array_equal(got, ref), k
                seen = max(seen, int(got.sum()))
        assert (m.host('status') == 0).all()
        assert seen >= 12                                   # (the manoeuvre ends: most vehicles are on their targets at rest)
    finally:
        m.solver.close()


def test_stop_rule_of_the_solve_kernel_ends_loops_like_the_reference():
    """`omgx_batch_set_stop` (ABI 9) through `BatchP2P.stop_at_arrival`: the solve kernel tests the reference's stop criterion
    (`vehicles/holonomic.py:145-151` on the state the prediction wrote) before it solves an agent and never solves it again once it
    held (`execution/simulator.py:39-62`).  Against a loop without the rule: the agents under way get the same bits, the flags clear
    at the update at which `arrived()` first holds, an agent whose loop has ended keeps its plan (x <- x0 in the double-buffered loop),
    multipliers and status, `iters` 0, and the launch statistics count only the agents that were solved.  Also through the
    three-stream product path."""
    import torch
    from omgtools import workloads
    from omgtools.batch import BatchP2P, receding_horizon_batch
    B = 24
    problem, P = workloads.holonomic_p2p(B)
    dev = torch.device('cuda', 0)
    opts = dict(tol=1e-3, max_iter=300)
    free = BatchP2P(problem, P, ops='hip', device=dev, options=opts)
    ruled = BatchP2P(problem, P, ops='hip', device=dev, options=opts)
    try:
        ruled.stop_at_arrival(stop_tol=2.5)
        stats = torch.zeros((60, 4), dtype=torch.int64, device=dev)
        for m in (free, ruled):
            m.solve_cold(bends=())
        assert np.array_equal(free.host('x'), ruled.host('x')) and ruled.host('under_way').all()
        ruled.solver.set_stats(stats)
        ended = np.zeros(B, dtype=bool)
        firsts = []
        for k in range(40):
            x_prev, lam_prev = ruled.host('x'), ruled.host('lam')
            crossed = free.step()
            assert ruled.step() == crossed
            now = free.arrived(2.5).cpu().numpy() & ~ended
            firsts += [k] * int(now.sum())
            ended |= now
            run = ~ended
            assert np.array_equal(ruled.host('under_way') != 0, run), k
            for name in ('x', 'lam', 'iters', 'status'):
                assert np.array_equal(ruled.host(name)[run], free.host(name)[run]), (k, name)
            assert (ruled.host('iters')[ended] == 0).all() and (ruled.host('status')[ended] == 0).all()
            if not crossed and ended.any():       # (a crossing shifts the kept plan and moves its multipliers by index like everyone's)
                assert np.array_equal(ruled.host('x')[ended & ~now], x_prev[ended & ~now])
                assert np.array_equal(ruled.host('lam')[ended], lam_prev[ended])
            assert int(stats[k, 3].item()) == int(run.sum()) and int(stats[k, 0].item()) == int(run.sum())
        assert ended.sum() >= B // 2 and len(set(firsts)) >= 3 and min(firsts) > 0
        ruled.solver.set_stats(None)
    finally:
        free.solver.close(); ruled.solver.close()
    # the product path: three sub-batches on their own streams, the rule in each
    a = receding_horizon_batch(problem, P, device=dev, n_streams=3, options=opts)
    b = BatchP2P(problem, P, ops='hip', device=dev, options=opts)
    try:
        for m in (a, b):
            m.stop_at_arrival(stop_tol=2.5)
            m.solve_cold(bends=())
        for _ in range(30):
            a.step(); b.step()
        ua = a.under_way.cpu().numpy()
        assert np.array_equal(ua, b.host('under_way')) and 0 < ua.sum() < B
        assert np.array_equal(a.x.cpu().numpy(), b.host('x')) and np.array_equal(a.iters.cpu().numpy(), b.host('iters'))
    finally:
        a.close(); b.solver.close()
