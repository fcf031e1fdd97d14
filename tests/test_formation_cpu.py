"""`FormationPoint2point` drop-in (host bookkeeping + `BatchADMM`) in the CPU tier: the script of
the reference's `examples/formation_holonomic.py:22-57` through `Simulator`, with the numpy ADMM
ops and the ORACLE host port as x-update solver injected for the HIP kernels (the GPU tier runs
the same script on the product path, tests/test_gpu_formation.py)."""
import numpy as np


def formation_example(ops, n=4, verbose=0, extra={}, vehicle_options=None):
    from omgtools import (Holonomic, Fleet, Environment, Obstacle, RegularPolyhedron, Rectangle, Circle, Square,
                          FormationPoint2point, Simulator)
    vehicles = [Holonomic(options=vehicle_options) for _ in range(n)]
    fleet = Fleet(vehicles)
    configuration = RegularPolyhedron(0.2, n, np.pi / 4.).vertices.T
    init_positions = [-1.5, -1.5] + configuration
    terminal_positions = [2., 2.] + configuration
    fleet.set_configuration(configuration.tolist())
    fleet.set_initial_conditions(init_positions.tolist())
    fleet.set_terminal_conditions(terminal_positions.tolist())
    environment = Environment(room={'shape': Square(5.)})
    rectangle = Rectangle(width=3., height=0.2)
    environment.add_obstacle(Obstacle({'position': [-2.1, -0.5]}, shape=rectangle))
    environment.add_obstacle(Obstacle({'position': [1.7, -0.5]}, shape=rectangle))
    trajectories = {'velocity': {'time': [3., 4.], 'values': [[-0.15, 0.0], [0., 0.15]]}}
    environment.add_obstacle(Obstacle({'position': [1.5, 0.5]}, shape=Circle(0.4),
                                      simulation={'trajectories': trajectories}))
    problem = FormationPoint2point(fleet, environment, options=dict({'rho': 1., 'horizon_time': 10, 'verbose': verbose}, **extra),
                                   ops=ops)
    problem.init()
    simulator = Simulator(problem)
    simulator.run()
    return problem, vehicles, terminal_positions, configuration


def check_formation_run(problem, vehicles, terminal_positions, configuration, max_dev=0.15, mean_dev=0.02):
    t_end = vehicles[0].signals['time'][0, -1]
    assert 8. < t_end < 40.
    for veh, target in zip(vehicles, terminal_positions):
        assert np.linalg.norm(veh.signals['state'][:, -1] - target) < 5e-3
        assert np.abs(veh.signals['input']).max() <= 0.5 + 3e-2      # one ADMM iteration per update: x-updates are not exact
    # formation kept along the way: vehicle k minus vehicle 0 stays close to the configured offset
    n = min(v.signals['state'].shape[1] for v in vehicles)
    for k in range(1, len(vehicles)):
        rel = vehicles[k].signals['state'][:, :n] - vehicles[0].signals['state'][:, :n]
        want = (configuration[k] - configuration[0])[:, None]
        dev = np.abs(rel - want)
        # rho = 1 with one iteration per update is soft in the passage between the obstacles
        assert dev.max() < max_dev and dev.mean() < mean_dev
    assert problem.residuals['primal'][-1] < 5e-2 and problem.iteration > 50


def test_formation_holonomic_example_cpu():
    from admm_numpy_ops import NumpyAdmmOps
    out = formation_example(lambda tpl, lay, p, x0, tol: NumpyAdmmOps(tpl, lay, p, x0, tol=tol))
    check_formation_run(*out)


def test_formation_tight_with_larger_rho():
    from admm_numpy_ops import NumpyAdmmOps
    out = formation_example(lambda tpl, lay, p, x0, tol: NumpyAdmmOps(tpl, lay, p, x0, tol=tol), extra={'rho': 5.})
    check_formation_run(*out, max_dev=0.1, mean_dev=0.01)


def test_device_side_prediction_equals_host_packing():
    """`FormationPoint2point` under `Simulator` with ideal prediction: the initial conditions of an update are predicted from
    the resident plan (one launch; the drop-in then packs ONE sub-problem's parameters instead of every vehicle's).  At every
    update the parameter block the device path leaves is the one the host packing would upload (`problems/admm.py:477-491`,
    `vehicles/vehicle.py:323-326`) -- through the knot crossings and the push of the moving obstacle."""
    import omgtools.admm as A
    import omgtools.formation as F
    from admm_numpy_ops import NumpyAdmmOps
    seen = dict(worst=0.0, n=0, problem=None, time=None)
    orig_iter, orig_du = A.BatchADMM.iterate, F.FormationPoint2point.dual_update

    def iterate(self, t_rel, sync=True):
        pr = seen['problem']
        if pr is not None and pr.device_predictions > seen['n']:          # this update was predicted on the device
            seen['n'] = pr.device_predictions
            P = pr._host_parameters(seen['time'])
            rho_col = pr.lay.p_rho                                          # (set by the iteration itself)
            cols = pr.host_cols[pr.host_cols != rho_col]
            seen['worst'] = max(seen['worst'], float(np.abs(P[:, cols] - self.ops.p[:, cols]).max()))
        return orig_iter(self, t_rel, sync)

    def dual_update(self, current_time, update_time):
        seen['problem'], seen['time'] = self, current_time
        return orig_du(self, current_time, update_time)
    A.BatchADMM.iterate, F.FormationPoint2point.dual_update = iterate, dual_update
    try:
        problem, vehicles, _, _ = formation_example(lambda tpl, lay, p, x0, tol: NumpyAdmmOps(tpl, lay, p, x0, tol=tol),
                                                    extra={'max_iter': 60}, vehicle_options={'ideal_prediction': True})
    finally:
        A.BatchADMM.iterate, F.FormationPoint2point.dual_update = orig_iter, orig_du
    assert problem.device_predictions >= 50 and seen['n'] == problem.device_predictions
    assert seen['worst'] < 1e-9, seen['worst']
    assert vehicles[0].signals['time'][0, -1] > 5.0                         # past the crossings at 1 s ... 5 s and the push at 3 s
