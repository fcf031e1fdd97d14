"""Seeded synthetic workloads of BASELINE.json's configs, built through the front end (any knot count,
any number of obstacles; `omgtools.workloads` loads the configurations of bench.py from committed bundles instead and
shares the parameter generators with this module).

Each builder returns (problem, P) where `problem` is an initialised single-agent
`Point2point` (its template is shared by the whole batch) and P is a dict of
per-agent arrays: 'p' [B, n_par] parameter vectors in the template's layout,
'x0' [B, n_var] initial guesses (`get_init_spline_value`, hyperplanes zero), and for
the Quadrotor class 'solver_options' (settings of the HIP solver that suit the
class, like the per-problem `solver_options` of the reference's examples).
"""
import numpy as np

from .shapes import Circle, Square, Sphere, Cube
from .vehicles import Holonomic, Holonomic3D, Quadrotor
from .environment import Environment, Obstacle
from .problems import Point2point


from . import workloads as wl

_place_obstacles = wl.place_obstacles


def holonomic_p2p(n_agents, knot_intervals=11, n_obs=3, seed=20240807 + 2,
                  safety_distance=0., horizon_time=10., options=None, gap=0.25):
    """Config 2: batch of independent Holonomic point-to-point problems with
    `n_obs` static circular obstacles each."""
    vehicle = Holonomic(options={'safety_distance': safety_distance})
    vehicle.define_knots(knot_intervals=knot_intervals)
    vehicle.set_initial_conditions([-1.5, -1.5])
    vehicle.set_terminal_conditions([1.5, 1.5])
    environment = Environment(room={'shape': Square(5.)})
    for l in range(n_obs):
        environment.add_obstacle(Obstacle({'position': [0., 0.]}, shape=Circle(0.3)))
    opts = {'horizon_time': horizon_time, 'verbose': 0}
    opts.update(options or {})
    problem = Point2point(vehicle, environment, options=opts)
    problem.init()
    tpl = problem.father.template
    return problem, wl.fill_holonomic_p2p(tpl, vehicle.label, problem.label, [o.label for o in environment.obstacles],
                                          len(vehicle.basis), n_agents, seed, horizon_time, gap=gap)


def quadrotor_p2p(n_agents, knot_intervals=13, n_obs=5, seed=20240807 + 3, horizon_time=5.,
                  options=None):
    """Config 3: batch of independent 2-D Quadrotor point-to-point problems (flat outputs y, z of
    degree 4; thrust / pitch-rate rows cubic in the coefficients, `vehicles/quadrotor.py:48-62`)
    with `n_obs` circular obstacles moving at constant velocity."""
    vehicle = Quadrotor(0.2)
    vehicle.define_knots(knot_intervals=knot_intervals)
    vehicle.set_initial_conditions([-4., -4.])
    vehicle.set_terminal_conditions([4., 4.])
    environment = Environment(room={'shape': Square(10.)})
    for l in range(n_obs):
        environment.add_obstacle(Obstacle({'position': [0., 0.], 'velocity': [0., 0.]}, shape=Circle(0.3)))
    opts = {'horizon_time': horizon_time, 'verbose': 0}
    opts.update(options or {})
    problem = Point2point(vehicle, environment, options=opts)
    problem.init()
    tpl = problem.father.template
    return problem, wl.fill_quadrotor_p2p(tpl, vehicle.label, problem.label, [o.label for o in environment.obstacles],
                                          len(vehicle.basis), vehicle.degree, n_agents, seed, horizon_time)


def holonomic3d_p2p(n_agents, knot_intervals=15, n_obs=10, seed=20240807 + 5, horizon_time=12.,
                    options=None):
    """Config 5: batch of independent Holonomic3D(Sphere(0.1)) problems with `n_obs` static
    spheres, `hard_term_con=True`."""
    vehicle = Holonomic3D(Sphere(0.1))
    vehicle.define_knots(knot_intervals=knot_intervals)
    vehicle.set_initial_conditions([-1.5, -1.5, -1.5])
    vehicle.set_terminal_conditions([1.5, 1.5, 1.5])
    environment = Environment(room={'shape': Cube(5.)})
    for l in range(n_obs):
        environment.add_obstacle(Obstacle({'position': [0., 0., 0.]}, shape=Sphere(0.2)))
    opts = {'horizon_time': horizon_time, 'hard_term_con': True, 'verbose': 0}
    opts.update(options or {})
    problem = Point2point(vehicle, environment, options=opts)
    problem.init()
    tpl = problem.father.template
    return problem, wl.fill_holonomic3d_p2p(tpl, vehicle.label, problem.label, [o.label for o in environment.obstacles],
                                            len(vehicle.basis), n_agents, seed, horizon_time)


def _obstacle_facts(environment, with_velocity=True):
    """What a fleet's x-update template reads per obstacle (`workloads.fill_formation`)."""
    out = []
    for obs in environment.obstacles:
        chk, rad = obs.shape.get_checkpoints()
        out.append(dict(label=obs.label, position=np.asarray(obs.signals['position'][:, -1], float).tolist(),
                        velocity=np.asarray(obs.signals['velocity'][:, -1], float).tolist(),
                        checkpoints=np.asarray(chk, float).tolist(), rad=np.asarray(rad, float).tolist()))
    return out


def formation_holonomic(n_agents, knot_intervals=10, seed=20240807 + 4, rho=1.0, horizon_time=10.,
                        with_obstacles=True, obstacles=None, interconnection='circular'):
    """Config 4: `n_agents` Holonomic vehicles keeping a regular-polygon formation
    (circular interconnection), shape of `examples/formation_holonomic.py:22-57`
    scaled to the fleet size.  Returns (problem, updater, father,
    layout, P) with P = {'p' [B,n_par], 'x0' [B,n_var], 'nbr' [B,2]}."""
    from .shapes import Rectangle, Circle
    from .formation import build_updx_template, FormationLayout, circular_neighbors
    vehicle = Holonomic()
    vehicle.define_knots(knot_intervals=knot_intervals)
    vehicle.set_initial_conditions([0., 0.])
    vehicle.set_terminal_conditions([1., 1.])
    radius = wl.formation_radius(n_agents)
    environment = Environment(room={'shape': Square(2. * radius + 12.)})
    if obstacles is not None:            # [(position, velocity, shape)]: a caller's own environment
        for pos, vel, shape in obstacles:
            environment.add_obstacle(Obstacle({'position': list(pos), 'velocity': list(vel)}, shape=shape))
    elif with_obstacles:
        rect = Rectangle(width=3., height=0.2)
        environment.add_obstacle(Obstacle({'position': [-2.6, -1.0]}, shape=rect))
        environment.add_obstacle(Obstacle({'position': [2.6, -1.0]}, shape=rect))
        # the moving circle of `examples/formation_holonomic.py:41-44` (there it is pushed at t = 3 s by a
        # simulation trajectory; the synthetic workload gives it that velocity from the start)
        environment.add_obstacle(Obstacle({'position': [1.5, 0.5], 'velocity': [-0.15, 0.0]}, shape=Circle(0.4)))
    # (`vehicles/fleet.py:49-60`: 'circular' = next and previous vehicle, 'full' = every other vehicle)
    n_nghb = 2 if interconnection == 'circular' else n_agents - 1
    problem, updater, father = build_updx_template(
        vehicle, environment, n_nghb, {'horizon_time': horizon_time})
    tpl = father.template
    lay = FormationLayout(tpl, vehicle, problem, updater, n_nghb)
    P = wl.fill_formation(tpl, lay, _obstacle_facts(environment), n_agents, horizon_time, rho)
    P['nbr'] = circular_neighbors(n_agents) if interconnection == 'circular' else \
        np.array([[j for j in range(n_agents) if j != i] for i in range(n_agents)], dtype=np.int32)
    return problem, updater, father, lay, P


def rendezvous_holonomic(n_agents, knot_intervals=10, seed=20240807 + 6, rho=2.0, horizon_time=10.):
    """RendezVous ADMM workload in the shape of `examples/rendezvous_holonomic_export.py:31-53` scaled to the fleet
    size: `n_agents` Holonomic vehicles start on a large circle and agree on a meeting point in a regular-polygon
    configuration (circular interconnection), one rectangular obstacle.  Returns (problem, updater, father,
    layout, P) like `formation_holonomic`."""
    from .shapes import Rectangle
    from .rendezvous import build_rendezvous_template, RendezVousLayout
    from .formation import circular_neighbors
    vehicle = Holonomic(shapes=Circle(0.1), options={'room_constraints': None})
    vehicle.define_knots(knot_intervals=knot_intervals)
    vehicle.set_initial_conditions([0., 0.])
    vehicle.set_terminal_conditions([0., 0.])
    radius = wl.formation_radius(n_agents)
    span = 3. + 2. * radius
    environment = Environment(room={'shape': Square(2. * span + 4.)})
    environment.add_obstacle(Obstacle({'position': [0.3 * span, -0.2 * span]}, shape=Rectangle(width=1.5, height=0.2)))
    problem, updater, father = build_rendezvous_template(vehicle, environment, 2, {'horizon_time': horizon_time})
    tpl = father.template
    lay = RendezVousLayout(tpl, vehicle, problem, updater, 2)
    P = wl.fill_rendezvous(tpl, lay, _obstacle_facts(environment), n_agents, seed, horizon_time, rho)
    P['nbr'] = circular_neighbors(n_agents)
    return problem, updater, father, lay, P
