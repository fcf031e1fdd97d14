"""Closed-loop fidelity at the BENCH's solver setting (tol = ipopt.tol = 1e-3, BatchP2P's warm-start options), in the form of the
reference's replay test (`export/tests/point2point/test.cpp:84-141`: after every update the sampled state and input trajectories
are compared with those of the other implementation, relative 1e-4) -- made two-sided, with scipy SLSQP in the role of the other
implementation (tests/golden/closed_loop_cfg2.npz -- and closed_loop_cfg3.npz, closed_loop_cfg5.npz for the Quadrotor and Holonomic3D classes --, generator tests/golden/generate_closed_loop.py: 64 agents of config 2, the loop
closed over SLSQP's own plans for 25 updates with two knot crossings; only the basin of the cold solve comes from the product's
algorithm).  The product runs ITS OWN closed loop -- cold solve from the reference's guess, then `BatchP2P.step` 25 times, every
step predicted from its own previous plan -- so solver error accumulates the way it would in a deployment.

Reported (and bounded): the largest deviation of the sampled position [m] and velocity [m/s] over all agents, updates and the first
20 samples (0.2 s: two update periods) of every new plan, and the two-sided relative figure |a - b| / max(|a|, |b|, floor) with
floor = 0.1 (10 cm, 10 cm/s; the velocity limit of the class is 0.5 m/s: the reference divides by the data value itself,
which is meaningless where a velocity crosses zero).

CPU tier: host build of the kernel source; GPU tier: the HIP path (`BatchP2P(ops='hip')`), same bounds."""
import os

import numpy as np
import pytest

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
FLOOR = 1e-1      # (tools/closed_loop.py FLOOR)

# The NLP is non-convex: now and then two solvers leave a step in different local minima (one passes a disc on the other side, or
# slides along a flat face of the L1 objective) and the two closed loops part for good.  Such agents are COUNTED (objective of
# the step off by more than 2e-2 relative -- a solve at tol 1e-3 ends with a barrier gap of 3e-3 to 5e-3 --, or the sampled position off by more than 0.1 m), not compared: at most MAX_PARTED of
# the 64; every other agent must stay within the bounds below for all 25 updates.
MAX_PARTED = 4
# what the loops achieve (measured, printed by the tests with -s) and what is asserted (a factor ~2 above it)
#            tol    state [m]   input [m/s]   relative
# measured (host build; the HIP path within a few per cent of it):  1e-3: 2.7e-2 m (median over the agents at the end of the
# loop 4e-3 m), 4.7e-2 m/s, relative 0.47 -- a solve at the reference's default tolerance ends with a barrier gap of 3e-3 to 5e-3
# in the objective, on the flat faces of the L1 objective the plan moves by centimetres with it, and the closed loop integrates the
# difference; 1e-6: 3.3e-4 m (median 2.7e-5 m), 1.7e-3 m/s, relative 8.6e-3; after the cold solve alone 7.7e-6 m / 7.5e-5 m/s.
# The reference's own figure (relative 1e-4, one-sided, between two runs of the SAME solver at the SAME tolerance) is not what
# a different solver at 1e-3 can meet against a converged one; these are the figures a user of the replacement gets.
BOUNDS = {1e-3: (6.0e-2, 1.0e-1, 0.75),      # (round 6: the relative entry at 1.5 x what is achieved, 0.47 -- it was 1.0, no bound at all)
          1e-6: (1.0e-3, 4.0e-3, 2.0e-2)}
# the Quadrotor class (closed_loop_cfg3.npz: 8 agents, 12 updates, three knot crossings, five moving circles; SLSQP's own accuracy on
# this class is ~1e-4 on the coefficients: it stops with 'positive directional derivative' at a feasibility of 1e-7)
# measured (host build): 1e-3: 5.4e-2 m (median at the end 2.0e-2), 0.18 m/s (the class flies at 1-2 m/s), relative 0.39;
# 1e-6: 2.3e-4 m (median 6.5e-5), 1.2e-3 m/s, relative 2.3e-3; after the cold solve alone 1.6e-6 m
BOUNDS_CFG3 = {1e-3: (1.2e-1, 4.0e-1, 0.6),     # (achieved 0.39)
               1e-6: (1.0e-3, 4.0e-3, 1.0e-2)}

# the Holonomic3D class (closed_loop_cfg5.npz, round 5: BASELINE config 5's class -- K = 15, ten moving spheres, 748 variables / 1812 rows;
# 8 agents, 18 updates, two knot crossings; SLSQP left 6 of the 152 solves unfinished -- two agents are out of the comparison)
# measured (host build): 1e-3: 2.4e-2 m (median at the end 2.2e-3), 2.3e-2 m/s, relative 8.4e-2; 1e-6: 4.8e-5 m (median 5.7e-6),
# 1.9e-4 m/s, relative 1.9e-3; after the cold solve alone 2.0e-7 m; no agent leaves the reference loop's basin
BOUNDS_CFG5 = {1e-3: (5.0e-2, 5.0e-2, 2.0e-1),
               1e-6: (1.5e-4, 6.0e-4, 6.0e-3)}


import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
import closed_loop as cl          # tools/closed_loop.py: the comparison itself (shared with bench.py's `parity_at_tol`)


def _objective(tpl):
    from oracle.nlp_numpy import NumpyNLP
    nlp = NumpyNLP(tpl)
    return lambda x, p: nlp.fg(x, nlp.term_coefs(p))[0]


def run_loop(make_mpc, tol, cfg='cfg2'):
    worst, first, parted_at, med, _ = cl.run_loop(make_mpc, tol, cfg, objective=_objective, max_capped=0 if tol >= 1e-3 else 2)
    return worst, first, parted_at, med


def check(make_mpc, tol, who, cfg='cfg2', bounds=None):
    worst, first, parted_at, med = run_loop(make_mpc, tol, cfg)
    print('\n%s closed loop, %s, tol %g, against SLSQP in the loop: %d agents parted from the reference '
          'loop %s; the others: position %.2e m (median at the end %.1e), velocity %.2e m/s, two-sided relative %.2e (floor %.0e); after the cold '
          'solve alone: %.2e m / %.2e m/s / %.2e'
          % ((who, {'cfg2': '64 Holonomic agents x 25 updates (two crossings)', 'cfg3': '8 Quadrotor agents x 12 updates (three crossings)',
               'cfg5': '8 Holonomic3D agents x 18 updates (two crossings)'}[cfg],
              tol, len(parted_at), parted_at) + (worst[0], med, worst[1], worst[2], FLOOR) + tuple(first)))
    b = bounds if bounds is not None else {'cfg2': BOUNDS, 'cfg3': BOUNDS_CFG3, 'cfg5': BOUNDS_CFG5}[cfg][tol]
    assert len(parted_at) <= (MAX_PARTED if cfg == 'cfg2' else 1), parted_at
    assert worst[0] < b[0] and worst[1] < b[1] and worst[2] < b[2], (worst, b)


# Round 6: IPOPT's absolute tolerances at their documented defaults beside tol = 1e-3 (omgx_options compl_inf_tol / constr_viol_tol, ABI 8:
# what the reference's solver configuration actually tests, `problems/problem.py:57`) -- measured 9.6e-3 m / 2.3e-2 m/s / 0.156, no agent parted
BOUNDS_IPOPT_DEFAULTS = (2.0e-2, 5.0e-2, 0.3)


CASES = [('cfg2', 1e-3), ('cfg2', 1e-6), ('cfg3', 1e-3), ('cfg3', 1e-6), ('cfg5', 1e-3), ('cfg5', 1e-6)]


@pytest.mark.parametrize('cfg,tol', CASES)
def test_port_closed_loop_follows_slsqp_in_the_loop(cfg, tol):
    from omgtools.batch import BatchP2P
    from oracle import port_binding

    def make(problem, P, opts):
        m = BatchP2P(problem, P, ops=port_binding, options=opts)
        m.n_threads = 8
        return m
    check(make, tol, 'host build', cfg)


@pytest.mark.gpu
@pytest.mark.parametrize('cfg,tol', CASES)
def test_hip_closed_loop_follows_slsqp_in_the_loop(cfg, tol):
    import torch
    from omgtools.batch import BatchP2P
    mpcs = []

    def make(problem, P, opts):
        mpcs.append(BatchP2P(problem, P, ops='hip', device=torch.device('cuda', 0), options=opts))
        return mpcs[-1]
    try:
        check(make, tol, 'HIP', cfg)
    finally:
        for m in mpcs:
            m.solver.close()


def _with(make, extra):
    return lambda problem, P, opts: make(problem, P, dict(opts, **extra))


def test_port_closed_loop_at_ipopt_default_tolerances():
    import omgtools.backend as be
    from omgtools.batch import BatchP2P
    from oracle import port_binding

    def make(problem, P, opts):
        m = BatchP2P(problem, P, ops=port_binding, options=opts)
        m.n_threads = 8
        return m
    check(_with(make, be.IPOPT_DEFAULT_TOLERANCES), 1e-3, 'host build, IPOPT default absolute tolerances', 'cfg2', BOUNDS_IPOPT_DEFAULTS)


@pytest.mark.gpu
def test_hip_closed_loop_at_ipopt_default_tolerances():
    import torch
    import omgtools.backend as be
    from omgtools.batch import BatchP2P
    mpcs = []

    def make(problem, P, opts):
        mpcs.append(BatchP2P(problem, P, ops='hip', device=torch.device('cuda', 0), options=opts))
        return mpcs[-1]
    try:
        check(_with(make, be.IPOPT_DEFAULT_TOLERANCES), 1e-3, 'HIP, IPOPT default absolute tolerances', 'cfg2', BOUNDS_IPOPT_DEFAULTS)
    finally:
        for m in mpcs:
            m.solver.close()
