"""Drop-in check shaped like the reference's own test (`tests/test_examples.py`
runs every example and asserts it completes): the script of
`examples/p2p_holonomic.py:23-51` runs unchanged against this package -- receding
horizon loop (Simulator -> Deployer -> predict/solve/store/simulate, knot shifts)
with every solve on the HIP path -- and the vehicle reaches its target without
touching the moving obstacle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_p2p_holonomic_example_runs_to_target():
    from omgtools import Holonomic, Environment, Obstacle, Circle, Square, Rectangle, Point2point, Simulator
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
    simulator = Simulator(problem)
    problem.plot('scene')
    vehicle.plot('input', knots=True, prediction=True, labels=['v_x (m/s)', 'v_y (m/s)'])
    trajectories, signals = simulator.run()
    state = signals['state']
    assert np.linalg.norm(state[:, -1] - np.array([2., 2.])) < 1e-2          # reached the target
    assert np.abs(signals['input']).max() <= 0.5 + 1e-3                      # velocity limits respected
    n = min(state.shape[1], obstacle.signals['position'].shape[1])
    dist = np.linalg.norm(state[:, :n] - obstacle.signals['position'][:, :n], axis=0)
    assert dist.min() >= 0.5 + 0.1 - 2e-2                                    # never inside obstacle + vehicle radius
    assert len(problem.update_times) > 50 and signals['time'][0, -1] > 8.


def test_p2p_quadrotor_example_runs_to_target():
    """`examples/p2p_quadrotor.py:22-43` on the HIP path (Quadrotor over a wall, T = 5 s)."""
    from omgtools import Quadrotor, Environment, Obstacle, Rectangle, Square, Point2point, Simulator
    vehicle = Quadrotor()
    vehicle.set_options({'safety_distance': 0.1})
    vehicle.set_initial_conditions([-4., -4., 0., 0., 0.])
    vehicle.set_terminal_conditions([4., 4.])
    environment = Environment(room={'shape': Square(10.)})
    environment.add_obstacle(Obstacle({'position': [-0.6, -5.4]}, shape=Rectangle(width=0.2, height=12.)))
    problem = Point2point(vehicle, environment, {'horizon_time': 5, 'verbose': 0})
    problem.init()
    trajectories, signals = Simulator(problem).run()
    state = signals['state']
    assert np.linalg.norm(state[:2, -1] - np.array([4., 4.])) < 2e-2
    inside = (np.abs(state[0] + 0.6) < 0.1 + 0.2 - 2e-2) & (state[1] < 0.6 + 0.2 - 2e-2)
    assert not inside.any()


def test_free_T_point2point_runs_to_target():
    """Free end time problem (`point2point.py:269-369`) through Simulator on the HIP path."""
    from test_examples_cpu import _free_T_run, check_free_T_run
    check_free_T_run(*_free_T_run())


def test_balls_example_with_quartic_rows_runs_to_target():
    """`examples/p2p_holonomic_balls.py` (free end time, 2-norm limits: rows of degree 4) on the HIP path."""
    from test_examples_cpu import _balls_run, check_balls_run
    check_balls_run(*_balls_run())


def test_intervehicle_avoidance_example_runs_to_target():
    """`examples/p2p_holonomic_interveh_avoidance.py` on the HIP path."""
    from test_examples_cpu import _interveh_run, check_interveh_run
    check_interveh_run(_interveh_run())


def test_free_end_point2point_solves_on_hip():
    """`FreeEndPoint2point` (duplicate terminal equality rows of the reference included) on the HIP path."""
    from test_examples_cpu import _free_end_solve, check_free_end
    check_free_end(*_free_end_solve())
