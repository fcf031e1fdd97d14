"""Parity of the HIP solve path (through the C ABI) against the oracle:
numpy interior point (independent statement) and the single-thread host port."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# fp64 tolerances (stated): the three implementations run the same iteration in
# different summation orders, so iterates agree to rounding amplified by the
# conditioning of the KKT systems.
TOL_X = 1e-6


def test_library_loads_and_reports_lds(cfg2_small):
    from omgtools.backend import BatchSolver
    problem, P = cfg2_small
    solver = BatchSolver(problem.father.template, 8)
    assert 0 < solver.lds_bytes <= 160 * 1024
    solver.close()


def test_fixed_iteration_iterates_match_port(cfg2_small):
    """Same number of iterations on both sides -> iterates agree to rounding."""
    from omgtools.backend import BatchSolver
    from oracle import port_binding
    problem, P = cfg2_small
    tpl = problem.father.template
    opts = dict(tol=1e-300, max_iter=12)
    solver = BatchSolver(tpl, 8, options=opts)
    res = solver.solve(P['p'], P['x0'])
    ref = port_binding.solve(tpl, P['p'], P['x0'], **opts)
    assert np.array_equal(res['iters'], ref['iters'])
    assert np.abs(res['x'] - ref['x']).max() < 1e-8
    assert np.abs(res['lam_g'] - ref['lam_g']).max() < 1e-6 * (1 + np.abs(ref['lam_g']).max())
    solver.close()


def test_cfg2_matches_port_and_numpy(cfg2_small):
    from omgtools.backend import BatchSolver
    from oracle import port_binding, ipm_numpy
    from oracle.nlp_numpy import NumpyNLP
    problem, P = cfg2_small
    tpl = problem.father.template
    tol = 1e-6
    solver = BatchSolver(tpl, 8, options=dict(tol=tol, max_iter=300))
    res = solver.solve(P['p'], P['x0'])
    # every sum of the kernel has a fixed order (owner-computes, no floating-point atomics): a second run
    # returns the same bits
    res2 = solver.solve(P['p'], P['x0'])
    assert np.array_equal(res['x'], res2['x']) and np.array_equal(res['lam_g'], res2['lam_g'])
    assert np.array_equal(res['status'], res2['status']) and np.array_equal(res['iters'], res2['iters'])
    ref = port_binding.solve(tpl, P['p'], P['x0'], tol=tol, max_iter=300)
    # same-source check: the kernel and its host build take the same decisions on all 8 agents
    assert np.array_equal(res['status'], ref['status'])
    assert set(np.unique(res['status'])) <= {0, 1, 2}            # never a numerical failure
    good = (res['status'] == 0) & (ref['status'] == 0)
    assert good.sum() >= 6
    lo, hi = tpl.entry_range(problem.vehicles[0].label, 'splines_seg0', 'var')
    nlp = NumpyNLP(tpl)
    for b in np.nonzero(good)[0]:
        # converged solutions: trajectory coefficients (the output the reference
        # consumes) to 1e-4, objective to 1e-6; hyperplane variables are not unique
        assert np.abs(res['x'][b, lo:hi] - ref['x'][b, lo:hi]).max() < 1e-4
        c = nlp.term_coefs(P['p'][b])
        f_gpu, g = nlp.fg(res['x'][b], c)
        f_ref, _ = nlp.fg(ref['x'][b], c)
        assert abs(f_gpu - f_ref) < 1e-6
        assert (g - tpl.ub).max() < 1e-5 and (tpl.lb - g).max() < 1e-5      # tol x row scaling
    # independent dense statement of the same iteration (no block structure, no MFMA): at this
    # tolerance its unpivoted dense LDL' may give up on the last ill-conditioned iterations, so
    # take the first agent it finishes
    checked = 0
    for b in np.nonzero(good)[0]:
        r_np = ipm_numpy.solve(nlp, P['x0'][b], P['p'][b], tpl.lb, tpl.ub,
                               opts={'tol': tol, 'max_iter': 300})
        if r_np['status'] != 0:
            continue
        assert np.abs(r_np['x'][lo:hi] - res['x'][b, lo:hi]).max() < 1e-4
        checked += 1
        break
    assert checked == 1
    solver.close()


def test_hip_reaches_the_slsqp_minimum():
    """Independent-solver parity (tests/test_independent_solver.py) on the HIP path."""
    from omgtools.backend import BatchSolver
    from test_independent_solver import slsqp_cases, check_against_slsqp

    def solve(tpl, p, x0):
        solver = BatchSolver(tpl, 1, options=dict(tol=3e-6, max_iter=500))
        res = solver.solve(p[None], x0[None])
        solver.close()
        return res
    check_against_slsqp(slsqp_cases(), solve)


def test_ragged_batch_and_masked_shift(cfg2_small):
    """Batch sizes that are not a multiple of anything (1, 7, 130 agents) give the same per-agent
    results; a masked warm-start shift leaves the unmasked agents untouched."""
    from omgtools.backend import BatchSolver
    from omgtools.splines import shiftoverknot_T
    problem, P = cfg2_small
    tpl = problem.father.template
    ref = None
    for B in (1, 7, 130):
        idx = np.arange(B) % 8
        solver = BatchSolver(tpl, B, options=dict(tol=1e-3, max_iter=300))
        res = solver.solve(P['p'][idx], P['x0'][idx])
        if ref is None:
            ref = res
        assert np.array_equal(res['status'][:1], ref['status'][:1])
        assert np.abs(res['x'][0] - ref['x'][0]).max() < 1e-6
        for b in range(B):
            assert np.abs(res['x'][b] - res['x'][idx[b]]).max() < 1e-6
        if B == 7:
            veh = problem.vehicles[0]
            lo, rows, cols = tpl.var_layout[(veh.label, 'splines_seg0')]
            Tm = shiftoverknot_T(veh.basis)
            x = res['x'].copy()
            mask = np.array([1, 0, 1, 0, 0, 1, 0], dtype=np.uint8)
            out = np.ascontiguousarray(x.copy())
            solver.shift(out, mask, np.array([[lo, rows, cols, 0]], dtype=np.int32), Tm.reshape(-1))      # in place
            want = x.copy()
            blk = x[:, lo:lo + rows * cols].reshape(B, cols, rows)
            want[:, lo:lo + rows * cols] = np.where(mask[:, None].astype(bool), (blk @ Tm.T).reshape(B, -1),
                                                    x[:, lo:lo + rows * cols])
            assert np.abs(out - want).max() < 1e-12
        solver.close()


def test_ipopt_absolute_tolerances_match_port(cfg2_small):
    """omgx_options version 8 (`compl_inf_tol`, `constr_viol_tol`: IPOPT's absolute tolerances on the unscaled problem, at its
    documented defaults): the HIP path takes the decisions of the host build -- same iteration counts -- and ends with
    complementarity products below 1e-4; without the two options the iteration counts of the plain solve are unchanged."""
    import omgtools.backend as be
    from oracle import port_binding
    problem, P = cfg2_small
    tpl = problem.father.template
    for extra in ({}, be.IPOPT_DEFAULT_TOLERANCES):
        opts = dict(tol=1e-3, max_iter=200, **extra)
        solver = be.BatchSolver(tpl, 8, options=opts)
        res = solver.solve(P['p'], P['x0'])
        solver.close()
        ref = port_binding.solve(tpl, P['p'], P['x0'], **opts)
        assert np.array_equal(res['status'], ref['status']) and (res['status'] == 0).all()
        assert np.abs(res['iters'] - ref['iters']).max() <= 1
        assert np.abs(res['x'] - ref['x']).max() < 1e-6


def test_refined_steps_match_port(cfg2_small):
    """omgx_options version 9, `refine` = 1: iterative refinement of regularised Newton steps (one more solve with the factors of the
    iteration, `kkt_solve2_wave` with the equality multipliers; the fall-back to the plain step when the first trial of the line search
    does not accept the refined one).  The HIP path takes the decisions of the host build -- cold solves and warm-started re-solves at
    moved parameters, tolerances 1e-3 and 1e-6 -- and needs fewer iterations than without the option at the tight tolerance; off (the
    default) nothing changes."""
    import omgtools.backend as be
    from oracle import port_binding
    problem, P = cfg2_small
    tpl = problem.father.template
    o_t = tpl.entry_range(problem.label, 't', 'par')[0]
    p2 = P['p'].copy()
    p2[:, o_t] += 0.1
    total = {}
    for tol in (1e-3, 1e-6):
        for refine in (0, 1):
            opts = dict(tol=tol, max_iter=300, refine=refine)
            solver = be.BatchSolver(tpl, 8, options=opts)
            res = solver.solve(P['p'], P['x0'])
            ref = port_binding.solve(tpl, P['p'], P['x0'], **opts)
            solver.close()
            solver = be.BatchSolver(tpl, 8, options=dict(opts, warm_start=1))      # (a fresh handle: no inertia correction carried over, like the port call)
            warm = solver.solve(p2, ref['x'], lam_g0=ref['lam_g'], status0=ref['status'])      # (both warm solves from the same point)
            solver.close()
            refw = port_binding.solve(tpl, p2, ref['x'], lam_g0=ref['lam_g'], status0=ref['status'], warm_start=1, **opts)
            for got, want in ((res, ref), (warm, refw)):
                assert np.array_equal(got['status'], want['status']) and (got['status'] == 0).all()
                assert np.abs(got['iters'] - want['iters']).max() <= (1 if tol == 1e-3 else 3)
                assert np.abs(got['x'] - want['x']).max() < (1e-6 if tol == 1e-3 else 2e-3)
            total[(tol, refine)] = int(res['iters'].sum() + warm['iters'].sum())
    assert total[(1e-6, 1)] < total[(1e-6, 0)]
