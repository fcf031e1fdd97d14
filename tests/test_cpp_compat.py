"""The C++ export compatibility layer: `omg::Point2Point`, `omg::Vehicle`, `omg::Holonomic` with the
public interface of the reference's exported classes (`export/point2point/Point2Point.hpp:45-107`,
`export/vehicles/Vehicle.hpp`, `Holonomic.hpp`) over libomgx.so (`omg-tools_amd/compat`).

CPU tier: the library builds, and the horizon-shift matrices it computes for its bases equal the front end's
`shiftoverknot_T` (pinned to the reference's, tests/golden/spline_kats.npz).  GPU tier: the replay test shaped like
`export/tests/point2point/test.cpp:84-141` -- the scenario of `examples/p2p_holonomic_export.py`, 25 updates (two knot
crossings) in C++, state and input trajectories of every update against CSV files written by the Python path
(`Deployer.update` of this package on the HIP solver), two-sided."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMPAT = os.path.join(ROOT, 'omg-tools_amd', 'compat')
CSRC = os.path.join(ROOT, 'omg-tools_amd', 'csrc')


@pytest.fixture(scope='module')
def compat_lib():
    subprocess.check_call(['make', '-C', COMPAT], stdout=subprocess.DEVNULL)
    return os.path.join(COMPAT, 'libomg_compat.so')


def test_shift_matrices_match_the_front_end(compat_lib):
    from omgtools.splines import BSplineBasis, shiftoverknot_T
    lib = C.CDLL(compat_lib)
    lib.omg_compat_shift_matrix.argtypes = [C.c_int, C.c_int, C.c_void_p]
    for degree, K in ((3, 10), (1, 10), (3, 11), (2, 7), (4, 13)):
        knots = np.r_[np.zeros(degree), np.linspace(0, 1, K + 1), np.ones(degree)]
        want = shiftoverknot_T(BSplineBasis(knots, degree))
        L = K + degree
        got = np.zeros((L, L))
        assert lib.omg_compat_shift_matrix(degree, K, got.ctypes.data) == L * L
        assert np.abs(got - want).max() < 1e-9, (degree, K, np.abs(got - want).max())


def _scenario():
    from omgtools import Holonomic, Environment, Obstacle, Point2point, Circle, Square, Rectangle
    vehicle = Holonomic(shapes=Circle(0.1), options={'room_constraint': None})
    vehicle.set_initial_conditions([0.0, 0.0])
    vehicle.set_terminal_conditions([3.5, 3.5])
    environment = Environment(room={'shape': Square(5.), 'position': [1.5, 1.5]})
    rectangle = Rectangle(width=3., height=0.2)
    environment.add_obstacle(Obstacle({'position': [-0.6, 1.0]}, shape=rectangle))
    environment.add_obstacle(Obstacle({'position': [3.2, 1.0]}, shape=rectangle))
    return vehicle, environment, Point2point


@pytest.mark.gpu
def test_replay_of_the_export_example_matches_the_python_path(compat_lib, tmp_path):
    import omgtools.backend as be
    from omgtools import Deployer
    tol_solver, n_iter, tl = 1e-5, 25, 5
    vehicle, environment, Point2point = _scenario()
    problem = Point2point(vehicle, environment, freeT=False,
                          options={'verbose': 0, 'solver_options': {'ipopt': {'ipopt.tol': tol_solver, 'ipopt.max_iter': 500}}})
    problem.init()
    tpl_path = be.save_template(problem.father.template, str(tmp_path / 'p2p.omgx'))
    # the Python path: Deployer.update with ideal prediction (the plan itself at the next update time), like the C++ loop
    vehicle.options['ideal_prediction'] = True
    deployer = Deployer(problem, sample_time=0.01, update_time=0.1)
    deployer.reset()
    rows_state, rows_input = [], []
    t = 0.0
    for i in range(n_iter):
        traj = deployer.update(t)
        assert problem.problem.stats()['return_status'] == 'Solve_Succeeded', i
        for k in range(2):
            rows_state.append(traj['state'][k, :tl])
            rows_input.append(traj['input'][k, :tl])
        t += 0.1
    np.savetxt(str(tmp_path / 'data_state.csv'), np.array(rows_state), delimiter=',', fmt='%.17g')
    np.savetxt(str(tmp_path / 'data_input.csv'), np.array(rows_input), delimiter=',', fmt='%.17g')
    exe = str(tmp_path / 'replay')
    subprocess.check_call(['g++', '-std=c++14', '-O1', os.path.join(ROOT, 'tests', 'cpp', 'replay.cpp'), '-I', COMPAT, '-L', COMPAT,
                           '-lomg_compat', '-L', CSRC, '-lomgx', '-Wl,-rpath,' + COMPAT, '-Wl,-rpath,' + CSRC, '-o', exe])
    env = dict(os.environ, OMG_TEMPLATE=tpl_path, OMG_TOL=str(tol_solver))
    out = subprocess.run([exe, str(tmp_path / 'data_state.csv'), str(tmp_path / 'data_input.csv'), str(n_iter), '1e-4'],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    text = out.stdout.decode()
    assert out.returncode == 0, text[-3000:]
    assert 'replayed %d updates' % n_iter in text
