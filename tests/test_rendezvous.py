"""RendezVous ADMM (`problems/rendezvous.py:26-67`): pinned to the reference through tests/golden/admm_rendezvous.npz
-- values produced by EXECUTING the reference's own `rendezvous.py`, `point2point.py` (FreeEndPoint2point),
`distributedproblem.py`, `dualmethod.py` and `admm.py` on the casadi stand-in (generator
tests/golden/generate_golden_admm.py, scenario `examples/rendezvous_holonomic_export.py:31-53`) -- and run:
the fleet agrees on a meeting point (CPU: numpy ops + host port; GPU: HIP kernels, same numbers)."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
FIX = os.path.join(HERE, 'golden', 'admm_rendezvous.npz')


@pytest.fixture(scope='module')
def fix():
    return np.load(FIX)


@pytest.fixture(scope='module')
def updx():
    import omgtools.backend as be
    from omgtools import Holonomic, Environment, Obstacle, Rectangle, Circle, Square
    from omgtools.rendezvous import build_rendezvous_template
    saved = be.create_nlp
    be.create_nlp = lambda tpl, opt, name='': (None, 0.)
    try:
        vehicle = Holonomic(shapes=Circle(0.1), options={'room_constraints': None})
        vehicle.set_initial_conditions([0., 3.])
        vehicle.set_terminal_conditions([0., 0.])
        environment = Environment(room={'shape': Square(5.), 'position': [1.5, 1.5]})
        environment.add_obstacle(Obstacle({'position': [3.2, 1.0]}, shape=Rectangle(width=3., height=0.2)))
        return build_rendezvous_template(vehicle, environment, 2, {'horizon_time': 10.})
    finally:
        be.create_nlp = saved


def test_xupdate_nlp_equals_the_reference(fix, updx):
    """`ADMM.construct_upd_x` (`admm.py:63-115`) over `FreeEndPoint2point` (`point2point.py:376-418`): the same
    variables (incl. the free end point `conT0`), parameters, rows, bounds, f and g as the reference's graphs."""
    from oracle.nlp_numpy import NumpyNLP
    from test_golden_admm import _ordinal_labels
    problem, updater, father = updx
    tpl = father.template
    assert (tpl.n_var, tpl.n_con, tpl.n_par) == (fix['updx_x0'].size, fix['updx_lb'].size, fix['updx_p0'].size) == (87, 340, 41)
    for which in ('var', 'par', 'con'):
        mine = [(name, off, r, c) for (_, name, off, r, c) in tpl.block_table(which)]
        ref = [(str(n).split('/')[-1], int(o), int(r), int(c)) for n, (o, r, c) in zip(fix['updx_%s_names' % which], fix['updx_%s_layout' % which])]
        assert [(o, r, c) for _, o, r, c in mine] == [(o, r, c) for _, o, r, c in ref], which
        if which != 'con':
            assert _ordinal_labels([n for n, _, _, _ in mine]) == _ordinal_labels([n for n, _, _, _ in ref]), which
    assert np.array_equal(tpl.lb, fix['updx_lb']) and np.array_equal(tpl.ub, fix['updx_ub'])
    nlp = NumpyNLP(tpl)
    for xv, pv, fr, gr in zip(fix['updx_xs'], fix['updx_ps'], fix['updx_fs'], fix['updx_gs']):
        f, g = nlp.fg(xv, nlp.term_coefs(pv))
        assert abs(f - fr) < 1e-9 * (1 + abs(fr))
        assert np.abs(g - gr).max() < 1e-9 * (1 + np.abs(gr).max())


def test_z_l_and_residual_updates_equal_the_reference(fix, updx):
    """Shared quantity = the fleet centre of the free end point, 2 numbers (`rendezvous.py:41-45`); coupling
    z_i - z_ij = 0 (`rendezvous.py:47-58` -> `admm.py:313-354`), neighbour order "next, previous"; closed-form
    z-update, multiplier update and residuals (`admm.py:117-168, 248-307`; no knot transform: not a spline)."""
    from omgtools.rendezvous import consensus_matrix, RendezVousLayout
    problem, updater, father = updx
    assert int(fix['n_shared']) == 2 and [int(i) % 4 for i in fix['nghb_index']] == [1, 3]
    assert np.abs(fix['updz_b']).max() == 0.0
    A, Ar = consensus_matrix(2, 2), fix['updz_A']
    assert A.shape == Ar.shape == (4, 6)
    proj = lambda M: np.eye(M.shape[1]) - M.T @ np.linalg.solve(M @ M.T, M)
    assert np.abs(proj(A) - proj(Ar)).max() < 1e-12
    lay = RendezVousLayout(father.template, problem.vehicles[0], problem, updater, 2)
    ns, nij = 2, 4
    for vin, vout in zip(fix['updz_in'], fix['updz_out']):
        x_i, l_i, l_ij, x_j = vin[:ns], vin[ns:2 * ns], vin[2 * ns:2 * ns + nij], vin[2 * ns + nij:2 * ns + 2 * nij]
        t, T, rho = vin[-3:]
        M, F = lay.zupdate(t / T)
        assert np.abs(M @ (np.r_[x_i, x_j] + np.r_[l_i, l_ij] / rho) - vout).max() < 1e-12 * (1 + np.abs(vout).max())
        assert np.array_equal(F, np.eye(6))
    for vin, vout in zip(fix['updl_in'], fix['updl_out']):
        x_i, z_i, z_ij = vin[:ns], vin[ns:2 * ns], vin[2 * ns:2 * ns + nij]
        l_i, l_ij = vin[2 * ns + nij:3 * ns + nij], vin[3 * ns + nij:3 * ns + 2 * nij]
        x_j, rho = vin[3 * ns + 2 * nij:3 * ns + 3 * nij], vin[-1]
        assert np.abs(np.r_[l_i + rho * (x_i - z_i), l_ij + rho * (x_j - z_ij)] - vout).max() < 1e-12 * (1 + np.abs(vout).max())
    for vin, vout in zip(fix['res_in'], fix['res_out']):
        o, parts = 0, []
        for n in (ns, ns, ns, nij, nij, nij):
            parts.append(vin[o:o + n]); o += n
        x_i, z_i, z_i_p, z_ij, z_ij_p, x_j = parts
        rho = vin[-1]
        pr = np.sum((np.r_[x_i, x_j] - np.r_[z_i, z_ij]) ** 2)
        dr = rho * np.sum((np.r_[z_i, z_ij] - np.r_[z_i_p, z_ij_p]) ** 2)
        assert np.abs(np.array([pr, dr, rho * pr + dr]) - vout).max() < 1e-12 * (1 + np.abs(vout).max())


def _fleet(n):
    import omgtools.backend as be
    from omgtools.scenarios import rendezvous_holonomic
    saved = be.create_nlp
    be.create_nlp = lambda tpl, opt, name='': (None, 0.)
    try:
        return rendezvous_holonomic(n)
    finally:
        be.create_nlp = saved


def test_fleet_agrees_on_a_meeting_point_cpu():
    """Eight vehicles, circular interconnection: the ADMM iteration (numpy ops, x-updates by the host port) drives
    the primal residual down and the fleet centres of the planned end points together; every x-update converges."""
    from omgtools.admm import BatchADMM
    from admm_numpy_ops import NumpyAdmmOps
    problem, updater, father, lay, P = _fleet(8)
    ops = NumpyAdmmOps(father.template, lay, P['p'], P['x0'])
    admm = BatchADMM(lay, P['nbr'], ops, rho=2.0)
    admm.initialize()
    spread = []
    for it in range(40):
        status, _ = admm.iterate(0.0)
        assert np.all(status == 0)
        centre = ops.x[:, lay.x_spl:lay.x_spl + 2] + ops.p[:, lay.p_rel:lay.p_rel + 2]
        spread.append(np.abs(centre - centre.mean(axis=0)).max())
    res = admm.residuals
    assert res[-1][0] < 0.05 * res[0][0] and spread[-1] < 0.05 * spread[0] and spread[-1] < 0.05
    # the planned trajectories end at the agreed points
    L = len(lay.basis)
    ends = ops.x[:, lay.x_traj:lay.x_traj + 2 * L].reshape(8, 2, L)[:, :, -1]
    assert np.abs(ends - ops.x[:, lay.x_spl:lay.x_spl + 2]).max() < 5e-2


@pytest.mark.gpu
def test_rendezvous_hip_matches_numpy_backend():
    import torch
    from omgtools.admm import BatchADMM, HipAdmmOps
    from omgtools.backend import BatchSolver
    from admm_numpy_ops import NumpyAdmmOps
    problem, updater, father, lay, P = _fleet(8)
    tpl = father.template
    # (the options NumpyAdmmOps hands its x-update solver at 1e-6: plain multiplier floor, plain backtracking)
    solver = BatchSolver(tpl, 8, options=dict(tol=1e-6, max_iter=200, warm_z_cap=0.0, max_soc=0))
    gpu = BatchADMM(lay, P['nbr'], HipAdmmOps(solver, tpl, lay, P['p'], P['x0'], torch.device('cuda', 0)), rho=2.0)
    cpu_ops = NumpyAdmmOps(tpl, lay, P['p'], P['x0'])
    cpu = BatchADMM(lay, P['nbr'], cpu_ops, rho=2.0)
    gpu.initialize()
    cpu.initialize()
    for it in range(6):
        st_g, res_g = gpu.iterate(0.0)
        st_c, res_c = cpu.iterate(0.0)
        assert np.all(st_g.cpu().numpy() == 0) and np.all(st_c == 0)
        assert np.allclose(res_g, res_c, rtol=1e-5, atol=1e-8)
        assert np.abs(gpu.ops.x.cpu().numpy()[:, lay.x_spl:lay.x_spl + 2] - cpu_ops.x[:, lay.x_spl:lay.x_spl + 2]).max() < 1e-6
        assert np.abs(gpu.ops.z_ij.cpu().numpy() - cpu_ops.z_ij).max() < 1e-6
    solver.close()


def rendezvous_example(ops, n=4, verbose=0):
    """The script of `examples/rendezvous_holonomic_export.py:31-53` (without the export) through `Simulator`."""
    from omgtools import (Holonomic, Fleet, Environment, Obstacle, RegularPolyhedron, Rectangle, Circle, Square,
                          RendezVous, Simulator)
    vehicles = [Holonomic(shapes=Circle(0.1), options={'room_constraints': None}) for _ in range(n)]
    fleet = Fleet(vehicles)
    configuration = RegularPolyhedron(0.2, n, np.pi / 4.).vertices.T
    fleet.set_configuration(configuration.tolist())
    fleet.set_initial_conditions([[0., 3.], [3., 3.], [3., 0.], [0., 0.]])
    fleet.set_terminal_conditions(np.zeros((n, 2)).tolist())
    environment = Environment(room={'shape': Square(5.), 'position': [1.5, 1.5]})
    environment.add_obstacle(Obstacle({'position': [3.2, 1.0]}, shape=Rectangle(width=3., height=0.2)))
    problem = RendezVous(fleet, environment, options={'rho': 2., 'horizon_time': 10, 'init_iter': 5, 'verbose': verbose},
                         ops=ops)
    problem.init()
    simulator = Simulator(problem)
    simulator.run()
    return problem, vehicles, configuration


def check_rendezvous_run(problem, vehicles, configuration):
    """The vehicles end at rest, in the configured formation around a common point they were not told."""
    ends = np.array([veh.signals['state'][:, -1] for veh in vehicles])
    centre = (ends - configuration).mean(axis=0)
    assert np.abs(ends - configuration - centre).max() < 5e-2
    assert 0.5 < centre[0] < 2.5 and 0.5 < centre[1] < 2.5              # somewhere between the four corners
    for veh in vehicles:
        assert np.linalg.norm(veh.signals["input"][:, -1]) < 0.15          # (the stop criterium looks at positions only)
        assert np.abs(veh.signals['input']).max() <= 0.5 + 3e-2
    assert problem.iteration > 30


def test_rendezvous_example_through_the_simulator_cpu():
    from admm_numpy_ops import NumpyAdmmOps
    out = rendezvous_example(lambda tpl, lay, p, x0, tol: NumpyAdmmOps(tpl, lay, p, x0, tol=tol))
    check_rendezvous_run(*out)


@pytest.mark.gpu
def test_rendezvous_example_through_the_simulator_hip():
    out = rendezvous_example('hip')
    check_rendezvous_run(*out)
