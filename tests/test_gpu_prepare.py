"""Round 6: the setup of a batch of solves as a kernel of its own (`ipm_prepare_kernel`, `omgx_batch_set_prepare`, ABI 8) -- the
parameter stage (basis rows at t / T, coefficient slots), the Jacobian and the rows at x0, row classification, gradient-based
scaling and the start values of the multipliers for ALL agents in one launch with many workgroups per CU, ahead of the solve
kernel whose persistent workgroups then start every solve by loading the agent's record.  The setup kernel runs the statements of
the in-kernel setup (`omgx::ipm_setup`, one source) with the same thread count: every workspace mode must return the SAME BITS
with the setup kernel on and off -- cold solves, warm-started solves, restart passes (`OMGX_ONLY_FAILED`), the receding-horizon
loop across a knot crossing."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _both(tpl, B, opts, p, x0, lbg=None, ubg=None, warm_from=None):
    from omgtools.backend import BatchSolver
    out = []
    for on in (True, False):
        s = BatchSolver(tpl, B, options=opts)
        try:
            s.set_prepare(on)
            kw = {} if lbg is None else dict(lbg=lbg, ubg=ubg)
            r = s.solve(p, x0, **kw)
            if warm_from is not None:           # a second, warm-started solve from the first one's result at moved parameters
                s.set_options(**dict(opts, warm_start=1))
                r2 = s.solve(warm_from, r['x'], lam_g0=r['lam_g'], status0=r['status'], **kw)
                r = dict(r, x2=r2['x'], lam2=r2['lam_g'], status2=r2['status'], iters2=r2['iters'])
            out.append((r, s.workspace()['mode']))
        finally:
            s.close()
    return out


def _same(a, b):
    for k in a:
        assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k


@pytest.mark.parametrize('name,B,mode', [('holonomic_p2p', 24, 5), ('quadrotor_p2p', 6, 1), ('holonomic3d_p2p', 5, 3)])
def test_cold_solves_same_bits_with_and_without_the_setup_kernel(name, B, mode):
    from omgtools import workloads
    problem, P = getattr(workloads, name)(B)
    tpl = problem.father.template
    (on, m_on), (off, m_off) = _both(tpl, B, dict(tol=1e-4, max_iter=120), P['p'], P['x0'])
    assert m_on == m_off == mode
    assert (on['status'] == 0).sum() >= B - 2
    _same(on, off)


def test_warm_solves_same_bits_with_and_without_the_setup_kernel():
    from omgtools import workloads
    B = 16
    problem, P = workloads.holonomic_p2p(B)
    tpl = problem.father.template
    p2 = P['p'].copy()
    o_t = tpl.entry_range(problem.label, 't', 'par')[0]
    p2[:, o_t] += 0.1                                  # the horizon clock moved on: every row that depends on t / T changes
    (on, _), (off, _) = _both(tpl, B, dict(tol=1e-3, max_iter=300), P['p'], P['x0'], warm_from=p2)
    assert (on['status2'] == 0).all() and on['iters2'].mean() < 0.5 * on['iters'].mean()
    _same(on, off)


def test_lifted_class_same_bits_with_and_without_the_setup_kernel():
    """Workspace mode 6, the general kernel instance: the start point is projected onto the defining rows of the lifted
    auxiliaries by the setup (the record carries the projected point)."""
    from omgtools import workloads
    tpl, P = workloads.agv_loop(4)
    (on, m_on), (off, _) = _both(tpl, 4, dict(tol=1e-3, max_iter=25), P['p'], P['x0'], lbg=P['lbg'], ubg=P['ubg'])
    assert m_on == 6
    _same(on, off)


def test_receding_horizon_loop_same_bits_across_a_crossing():
    import torch
    from omgtools import workloads
    from omgtools.batch import BatchP2P
    B = 32
    problem, P = workloads.holonomic_p2p(B)
    dev = torch.device('cuda', 0)
    logs = []
    for on in (True, False):
        mpc = BatchP2P(problem, P, ops='hip', device=dev, options=dict(tol=1e-3, max_iter=300))
        try:
            mpc.solver.set_prepare(on)
            mpc.solve_cold()
            crossed = 0
            for _ in range(12):
                crossed += int(bool(mpc.step()))
            logs.append(dict(x=mpc.host('x'), lam=mpc.host('lam'), status=mpc.host('status'), iters=mpc.host('iters'), crossed=crossed))
        finally:
            mpc.solver.close()
    assert logs[0]['crossed'] >= 1
    _same(logs[0], logs[1])
