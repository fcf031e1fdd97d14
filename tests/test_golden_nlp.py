"""The NLP our front end builds == the NLP the reference builds.

tests/golden/nlp_*.npz hold f(x,p), g(x,p), bounds, initial values and the flat
layouts produced by the REFERENCE's own construct code (executed once in the
build container by tests/golden/generate_golden.py under a numeric casadi
stand-in).  Here the same scenarios are built with this repo's front end and
evaluated with the numpy oracle over the flat template arrays -- the arrays the
HIP kernel consumes.  Tolerance: 1e-9 relative (fp64, different summation order).
"""
import os
import re

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(autouse=True)
def no_device(monkeypatch):
    import omgtools.backend as be
    monkeypatch.setattr(be, 'create_nlp', lambda tpl, opt, name='': (None, 0.))


def _norm(name):
    return re.sub(r'(vehicle|obstacle|p2p|environment|problem)\d+', r'\1#', name)


def build(tag):
    from omgtools import (Holonomic, Holonomic3D, Quadrotor, Environment, Obstacle, Point2point,
                          Circle, Square, Rectangle, Sphere, Cube)
    quiet = {'verbose': 0}
    if tag == 'cfg1_p2p_holonomic':
        vehicle = Holonomic()
        vehicle.set_options({'safety_distance': 0.1})
        vehicle.set_options({'ideal_prediction': False})
        vehicle.set_initial_conditions([-1.5, -1.5])
        vehicle.set_terminal_conditions([2., 2.])
        environment = Environment(room={'shape': Square(5.)})
        trajectories = {'velocity': {'time': [0., 40.], 'values': [[-0.35, 0.35], [0., 0.15]]}}
        environment.add_obstacle(Obstacle({'position': [1.5, -1]}, shape=Circle(0.5),
                                          options={'bounce': False},
                                          simulation={'trajectories': trajectories}))
        problem = Point2point(vehicle, environment, options=quiet, freeT=False)
    elif tag == 'cfg2_holonomic_k11_o3':
        vehicle = Holonomic()
        vehicle.define_knots(knot_intervals=11)
        vehicle.set_initial_conditions([-1.5, -1.2])
        vehicle.set_terminal_conditions([1.4, 1.7])
        environment = Environment(room={'shape': Square(5.)})
        for pos, r in (([0.1, -0.3], 0.3), ([-0.5, 0.4], 0.25), ([0.6, 0.5], 0.35)):
            environment.add_obstacle(Obstacle({'position': pos}, shape=Circle(r)))
        problem = Point2point(vehicle, environment, options=quiet, freeT=False)
    elif tag == 'holonomic_rectangles':
        vehicle = Holonomic(shapes=Rectangle(0.3, 0.2))
        vehicle.set_initial_conditions([-1.5, -1.5])
        vehicle.set_terminal_conditions([2., 2.])
        environment = Environment(room={'shape': Square(5.)})
        environment.add_obstacle(Obstacle({'position': [0.2, -0.4]},
                                          shape=Rectangle(width=1., height=0.4)))
        problem = Point2point(vehicle, environment, options=quiet, freeT=False)
    elif tag == 'holonomic3d_spheres':
        vehicle = Holonomic3D(Sphere(0.1))
        vehicle.set_initial_conditions([-1.5, -1.5, -1.])
        vehicle.set_terminal_conditions([1.5, 1.5, 1.])
        environment = Environment(room={'shape': Cube(5.)})
        for pos, r in (([0.1, -0.3, 0.], 0.3), ([-0.5, 0.4, 0.2], 0.25)):
            environment.add_obstacle(Obstacle({'position': pos}, shape=Sphere(r)))
        opts = dict(quiet)
        opts.update({'horizon_time': 12., 'hard_term_con': True})
        problem = Point2point(vehicle, environment, options=opts, freeT=False)
    elif tag == 'quadrotor_k13_o2':
        vehicle = Quadrotor(0.2)
        vehicle.define_knots(knot_intervals=13)
        vehicle.set_initial_conditions([-4., -4.])
        vehicle.set_terminal_conditions([4., 4.])
        environment = Environment(room={'shape': Square(10.)})
        for pos, vel, r in (([0., -1.], [0.1, 0.05], 0.4), ([-1.5, 1.], [-0.1, 0.12], 0.3)):
            environment.add_obstacle(Obstacle({'position': pos, 'velocity': vel}, shape=Circle(r)))
        opts = dict(quiet)
        opts.update({'horizon_time': 5.})
        problem = Point2point(vehicle, environment, options=opts, freeT=False)
    elif tag == 'freeT_holonomic':
        vehicle = Holonomic()
        vehicle.set_initial_conditions([-1.5, -1.5])
        vehicle.set_terminal_conditions([2., 2.])
        environment = Environment(room={'shape': Square(5.)})
        environment.add_obstacle(Obstacle({'position': [0.2, -0.4]}, shape=Circle(0.4)))
        environment.add_obstacle(Obstacle({'position': [1.0, 1.2], 'velocity': [-0.1, 0.05]}, shape=Circle(0.3)))
        problem = Point2point(vehicle, environment, options=quiet, freeT=True)
    elif tag == 'interveh_holonomic':
        vehicles = [Holonomic() for k in range(2)]
        for k, vehicle in enumerate(vehicles):
            vehicle.set_initial_conditions([1.5 * np.cos(k * np.pi), 1.5 * np.sin(k * np.pi) + 0.1 * k])
            vehicle.set_terminal_conditions([-1.5 * np.cos(k * np.pi), -1.5 * np.sin(k * np.pi)])
        environment = Environment(room={'shape': Square(5.)})
        environment.add_obstacle(Obstacle({'position': [0.1, 0.9]}, shape=Circle(0.3)))
        problem = Point2point(vehicles, environment, options=quiet, freeT=False)
        problem.set_options({'inter_vehicle_avoidance': True})
    elif tag == 'freeend_holonomic':
        from omgtools import FreeEndPoint2point
        vehicle = Holonomic()
        vehicle.set_initial_conditions([-1.5, -1.5])
        vehicle.set_terminal_conditions([2., 2.])
        environment = Environment(room={'shape': Square(5.)})
        environment.add_obstacle(Obstacle({'position': [0.2, -0.4]}, shape=Circle(0.4)))
        problem = FreeEndPoint2point(vehicle, environment, quiet, {vehicle: [0]})
    elif tag == 'freeT_balls_norm2':
        # `examples/p2p_holonomic_balls.py`: free end time, 2-norm limits (terms of degree 4 in the variables), moving circles
        vehicle = Holonomic(shapes=Circle(0.2), options={'syslimit': 'norm_2'})
        vehicle.define_knots(knot_intervals=10)
        vehicle.set_initial_conditions([-4., 0])
        vehicle.set_terminal_conditions([4., 0])
        environment = Environment(room={'shape': Square(10.)})
        environment.add_obstacle(Obstacle({'position': [0., -0.5]}, shape=Circle(0.75), simulation={
            'trajectories': {'velocity': {'time': [0, 4.5], 'values': [[0., 0.0], [0., 0.35]]}}}))
        environment.add_obstacle(Obstacle({'position': [2., 0.5]}, shape=Circle(0.75)))
        environment.add_obstacle(Obstacle({'position': [-2., 0.5]}, shape=Circle(0.75)))
        environment.add_obstacle(Obstacle({'position': [0., -2.25]}, shape=Circle(0.75), simulation={
            'trajectories': {'velocity': {'time': [0, 5.], 'values': [[0., 0.0], [0., 0.25]]}}}))
        problem = Point2point(vehicle, environment, options=quiet, freeT=True)
    problem.init()
    return problem


TAGS = ['cfg1_p2p_holonomic', 'cfg2_holonomic_k11_o3', 'holonomic_rectangles',
        'holonomic3d_spheres', 'quadrotor_k13_o2', 'freeT_holonomic', 'interveh_holonomic', 'freeend_holonomic',
        'freeT_balls_norm2']


@pytest.mark.parametrize('tag', TAGS)
def test_nlp_matches_reference(tag):
    from oracle.nlp_numpy import NumpyNLP
    gold = np.load(os.path.join(GOLD, 'nlp_%s.npz' % tag), allow_pickle=True)
    problem = build(tag)
    tpl = problem.father.template
    # flat layouts: same order, offsets and shapes as the reference's structs
    for which, key in (('var', 'var_layout'), ('par', 'par_layout'), ('con', 'con_layout')):
        mine = [(_norm(lab + '/' + name) if which != 'con' else _norm(name), off, r, c)
                for (lab, name, off, r, c) in tpl.block_table(which)]
        ref = [(_norm(str(n)), int(o), int(r), int(c)) for (n, o, r, c) in gold[key]]
        assert mine == ref, which
    assert np.array_equal(tpl.lb, gold['lb']) and np.array_equal(tpl.ub, gold['ub'])
    np.testing.assert_allclose(problem.father.get_variables().cat, gold['x_init'], atol=1e-14)
    problem.reinitialize()
    np.testing.assert_allclose(problem.father.get_variables().cat, gold['x_reinit'], atol=1e-14)
    np.testing.assert_allclose(problem.father.set_parameters(0.).cat, gold['p0'], atol=1e-14)
    nlp = NumpyNLP(tpl)
    for x, p, f_ref, g_ref in zip(gold['X'], gold['P'], gold['F'], gold['G']):
        f, g = nlp.fg(x, nlp.term_coefs(p))
        scale = 1.0 + np.abs(g_ref)
        assert np.abs(f - f_ref) <= 1e-9 * (1 + abs(f_ref))
        assert (np.abs(g - g_ref) / scale).max() <= 1e-9
