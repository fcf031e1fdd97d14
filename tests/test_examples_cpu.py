"""Host logic of the drop-in surface in the CPU tier: scripts of the reference's examples run
through `Simulator` (Deployer -> predict / solve / store / simulate, knot shifts, vehicle dynamics)
with the ORACLE host port standing in for the HIP solver (tests/port_solver.py; the product path
has no CPU fallback).  Shaped like the reference's `tests/test_examples.py`, which runs every
example and asserts that it completes -- here the outcome is asserted too."""
import numpy as np
import pytest


@pytest.fixture(autouse=True)
def port_backend(monkeypatch):
    import omgtools.backend as be
    import port_solver
    monkeypatch.setattr(be, 'create_nlp', port_solver.create_nlp)


def test_p2p_quadrotor_example_reaches_target():
    """`examples/p2p_quadrotor.py:22-43`: Quadrotor over a wall, safety distance, T = 5 s."""
    from omgtools import Quadrotor, Environment, Obstacle, Rectangle, Square, Point2point, Simulator
    vehicle = Quadrotor()
    vehicle.set_options({'safety_distance': 0.1})
    vehicle.set_initial_conditions([-4., -4., 0., 0., 0.])
    vehicle.set_terminal_conditions([4., 4.])
    environment = Environment(room={'shape': Square(10.)})
    environment.add_obstacle(Obstacle({'position': [-0.6, -5.4]}, shape=Rectangle(width=0.2, height=12.)))
    problem = Point2point(vehicle, environment, {'horizon_time': 5, 'verbose': 0})
    problem.init()
    simulator = Simulator(problem)
    problem.plot('scene')
    vehicle.plot('input', knots=True, label=['Thrust force (N/kg)', 'Pitch rate (rad/s)'])
    trajectories, signals = simulator.run()
    state = signals['state']
    assert np.linalg.norm(state[:2, -1] - np.array([4., 4.])) < 2e-2
    # never inside the wall (x in [-0.7, -0.5], y < 0.6) inflated by the vehicle radius
    inside = (np.abs(state[0] + 0.6) < 0.1 + 0.2 - 2e-2) & (state[1] < 0.6 + 0.2 - 2e-2)
    assert not inside.any()
    u = signals['input']
    assert u[0].min() > 2. - 0.3 and u[0].max() < 15. + 0.3            # thrust limits (`quadrotor.py:34-35`)


def test_p2p_holonomic_example_reaches_target():
    """`examples/p2p_holonomic.py:23-51` (the GPU tier runs the same script on the HIP path)."""
    from omgtools import Holonomic, Environment, Obstacle, Circle, Square, Point2point, Simulator
    vehicle = Holonomic()
    vehicle.set_options({'safety_distance': 0.1})
    vehicle.set_options({'ideal_prediction': False})
    vehicle.set_initial_conditions([-1.5, -1.5])
    vehicle.set_terminal_conditions([2., 2.])
    environment = Environment(room={'shape': Square(5.)})
    trajectories = {'velocity': {'time': [0., 40.], 'values': [[-0.35, 0.35], [0., 0.15]]}}
    obstacle = Obstacle({'position': [1.5, -1]}, shape=Circle(0.5), options={'bounce': False},
                        simulation={'trajectories': trajectories})
    environment.add_obstacle(obstacle)
    problem = Point2point(vehicle, environment, options={'verbose': 0}, freeT=False)
    problem.init()
    trajectories, signals = Simulator(problem).run()
    state = signals['state']
    assert np.linalg.norm(state[:, -1] - np.array([2., 2.])) < 1e-2
    n = min(state.shape[1], obstacle.signals['position'].shape[1])
    dist = np.linalg.norm(state[:, :n] - obstacle.signals['position'][:, :n], axis=0)
    assert dist.min() >= 0.5 + 0.1 - 2e-2


def _free_T_run():
    from omgtools import Holonomic, Environment, Obstacle, Circle, Square, Point2point, Simulator
    vehicle = Holonomic()
    vehicle.set_initial_conditions([-1.5, -1.5])
    vehicle.set_terminal_conditions([2., 2.])
    environment = Environment(room={'shape': Square(5.)})
    environment.add_obstacle(Obstacle({'position': [0.2, -0.4]}, shape=Circle(0.4)))
    moving = Obstacle({'position': [1.0, 1.2], 'velocity': [-0.1, 0.05]}, shape=Circle(0.3))
    environment.add_obstacle(moving)
    problem = Point2point(vehicle, environment, options={'verbose': 0}, freeT=True)
    problem.init()
    trajectories, signals = Simulator(problem).run()
    return problem, signals, moving


def check_free_T_run(problem, signals, moving):
    state = signals['state']
    assert np.linalg.norm(state[:, -1] - np.array([2., 2.])) < 1e-2
    assert np.abs(signals['input']).max() <= 0.5 + 1e-3
    # free end time: close to the time-optimal transfer (4.95 m at |v|_inf <= 0.5: >= 7 s), and the
    # objective is the elapsed motion time (`point2point.py:361-365`)
    t_end = signals['time'][0, -1]
    assert 7.0 < t_end < 9.5 and abs(problem.compute_objective() - t_end) < 0.11
    assert np.linalg.norm(state - np.array([[0.2], [-0.4]]), axis=0).min() >= 0.4 + 0.1 - 2e-2
    n = min(state.shape[1], moving.signals['position'].shape[1])
    assert np.linalg.norm(state[:, :n] - moving.signals['position'][:, :n], axis=0).min() >= 0.3 + 0.1 - 2e-2


def test_free_T_point2point_reaches_target():
    """`FreeTPoint2point` (`point2point.py:269-369`): T is a variable and the objective; the NLP
    itself is pinned against the reference in tests/test_golden_nlp.py (freeT_holonomic)."""
    check_free_T_run(*_free_T_run())


def _balls_run():
    """`examples/p2p_holonomic_balls.py:22-54`: free end time, 2-norm velocity / acceleration limits (rows of degree 4 in
    the variables), two moving and two standing circles."""
    from omgtools import Holonomic, Environment, Obstacle, Circle, Square, Point2point, Simulator
    vehicle = Holonomic(shapes=Circle(0.2), options={'syslimit': 'norm_2'})
    vehicle.define_knots(knot_intervals=10)
    vehicle.set_initial_conditions([-4., 0])
    vehicle.set_terminal_conditions([4., 0])
    environment = Environment(room={'shape': Square(10.)})
    trajectories1 = {'velocity': {'time': [0, 4.5], 'values': [[0., 0.0], [0., 0.35]]}}
    trajectories2 = {'velocity': {'time': [0, 5.], 'values': [[0., 0.0], [0., 0.25]]}}
    obstacles = [Obstacle({'position': [0., -0.5]}, shape=Circle(0.75), simulation={'trajectories': trajectories1}),
                 Obstacle({'position': [2., 0.5]}, shape=Circle(0.75)),
                 Obstacle({'position': [-2., 0.5]}, shape=Circle(0.75)),
                 Obstacle({'position': [0., -2.25]}, shape=Circle(0.75), simulation={'trajectories': trajectories2})]
    for obstacle in obstacles:
        environment.add_obstacle(obstacle)
    problem = Point2point(vehicle, environment, options={'verbose': 0}, freeT=True)
    problem.init()
    trajectories, signals = Simulator(problem).run()
    return problem, signals, obstacles


def check_balls_run(problem, signals, obstacles):
    state = signals['state']
    assert problem.father.template.t_nv.max() == 4
    assert np.linalg.norm(state[:, -1] - np.array([4., 0.])) < 1e-2
    assert np.linalg.norm(signals['input'], axis=0).max() <= 0.5 + 1e-3          # |v|_2 <= vmax (`holonomic.py:54-58`)
    t_end = signals['time'][0, -1]
    assert 16.0 < t_end < 30.0                                                   # 8 m at 0.5 m/s: >= 16 s
    for obstacle in obstacles:
        n = min(state.shape[1], obstacle.signals['position'].shape[1])
        dist = np.linalg.norm(state[:, :n] - obstacle.signals['position'][:, :n], axis=0)
        assert dist.min() >= 0.75 + 0.2 - 2e-2


def test_free_T_with_two_norm_limits_and_moving_circles():
    check_balls_run(*_balls_run())


def _interveh_run(N=2):
    """`examples/p2p_holonomic_interveh_avoidance.py:22-48`: vehicles swap places through the centre."""
    from omgtools import Holonomic, Environment, Square, Point2point, Simulator
    vehicles = [Holonomic() for k in range(N)]
    for k, vehicle in enumerate(vehicles):
        vehicle.set_initial_conditions([1.5 * np.cos((k * 2. * np.pi) / N), 1.5 * np.sin((k * 2. * np.pi) / N)])
        vehicle.set_terminal_conditions([-1.5 * np.cos((k * 2. * np.pi) / N), -1.5 * np.sin((k * 2. * np.pi) / N)])
    environment = Environment(room={'shape': Square(5.)})
    problem = Point2point(vehicles, environment, freeT=False)
    problem.set_options({'inter_vehicle_avoidance': True, 'verbose': 0})
    problem.init()
    Simulator(problem).run()
    return vehicles


def check_interveh_run(vehicles):
    N = len(vehicles)
    n = min(v.signals['state'].shape[1] for v in vehicles)
    for k, v in enumerate(vehicles):
        goal = np.array([-1.5 * np.cos((k * 2. * np.pi) / N), -1.5 * np.sin((k * 2. * np.pi) / N)])
        assert np.linalg.norm(v.signals['state'][:2, -1] - goal) < 1e-2
        assert np.abs(v.signals['input']).max() <= 0.5 + 1e-3
    for k in range(N):
        for l in range(k + 1, N):
            d = np.linalg.norm(vehicles[k].signals['state'][:2, :n] - vehicles[l].signals['state'][:2, :n], axis=0)
            assert d.min() >= 0.1 + 0.1 - 2e-2        # two Circle(0.1) vehicles never overlap


def test_intervehicle_avoidance_example():
    """Two vehicles in ONE problem with separating hyperplanes between them (`environment.py:148-176`);
    the NLP is pinned against the reference in tests/test_golden_nlp.py (interveh_holonomic)."""
    check_interveh_run(_interveh_run())


def _free_end_solve():
    """`FreeEndPoint2point` (`point2point.py:376-418`) with a free x target: the plan keeps x and
    moves y to the target (the NLP is pinned in tests/test_golden_nlp.py: freeend_holonomic)."""
    from omgtools import Holonomic, Environment, Obstacle, Circle, Square, FreeEndPoint2point
    vehicle = Holonomic()
    vehicle.set_initial_conditions([-1.5, -1.5])
    vehicle.set_terminal_conditions([2., 2.])
    environment = Environment(room={'shape': Square(5.)})
    environment.add_obstacle(Obstacle({'position': [0.2, -0.4]}, shape=Circle(0.4)))
    problem = FreeEndPoint2point(vehicle, environment, {'verbose': 0}, {vehicle: [0]})
    problem.init()
    problem.reinitialize()
    problem.solve(0., 0.1)
    return problem, vehicle


def check_free_end(problem, vehicle):
    assert problem.problem.stats()['return_status'] == 'Solve_Succeeded'
    tpl = problem.father.template
    x = problem.father.get_variables().cat
    lo, hi = tpl.entry_range(problem.label, 'conT0', 'var')
    conT = x[lo:hi]
    lo, hi = tpl.entry_range(vehicle.label, 'splines_seg0', 'var')
    c = x[lo:hi].reshape(2, -1)
    assert abs(conT[0] + 1.5) < 1e-2                      # the free x target settles at the start
    assert abs(c[0, -1] - conT[0]) < 1e-2 and abs(c[1, -1] - 2.) < 1e-6   # end point: (conT, fixed y target)


def test_free_end_point2point_solves():
    check_free_end(*_free_end_solve())
