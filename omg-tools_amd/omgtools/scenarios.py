"""Seeded synthetic workloads of BASELINE.json's configs.

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


def _place_obstacles(rng, n_obs, start, goal, r_lo, r_hi, box, clearance=0.3, gap=0.25):
    """Rejection sampling of circular obstacles: not within `clearance` of start/goal, and any
    two discs leave a passable gap (`gap` >= vehicle diameter 0.2 m + 5 cm).  Two discs with an
    impassable gap across the straight start-goal line put the reference's straight-line initial
    guess in a homotopy class no local NLP method (IPOPT included) can leave."""
    centres, radii, tries = [], [], 0
    while len(centres) < n_obs:
        tries += 1
        if tries > 2000:                        # restart an unlucky draw
            centres, radii, tries = [], [], 0
        r = rng.uniform(r_lo, r_hi)
        c = rng.uniform(-box, box, size=2)
        if np.linalg.norm(c - start) < r + clearance or np.linalg.norm(c - goal) < r + clearance:
            continue
        if any(np.linalg.norm(c - c2) < r + r2 + gap for c2, r2 in zip(centres, radii)):
            continue
        centres.append(c)
        radii.append(r)
    return np.array(centres), np.array(radii)


def holonomic_p2p(n_agents, knot_intervals=11, n_obs=3, seed=20240807 + 2,
                  safety_distance=0., horizon_time=10., options=None, gap=0.25):
    """Config 2: batch of independent Holonomic point-to-point problems with
    `n_obs` static circular obstacles each."""
    rng = np.random.default_rng(seed)
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
    L = len(vehicle.basis)
    p = np.zeros((n_agents, tpl.n_par))
    x0 = np.zeros((n_agents, tpl.n_var))
    rng_pos = rng
    for b in range(n_agents):
        start = rng_pos.uniform(-2., -1., size=2)
        goal = rng_pos.uniform(1., 2., size=2)
        centres, radii = _place_obstacles(rng_pos, n_obs, start, goal, 0.2, 0.4, 0.8, gap=gap)
        lo, hi = tpl.entry_range(vehicle.label, 'state0', 'par'); p[b, lo:hi] = start
        lo, hi = tpl.entry_range(vehicle.label, 'poseT', 'par'); p[b, lo:hi] = goal
        for l, obs in enumerate(environment.obstacles):
            lo, hi = tpl.entry_range(obs.label, 'x', 'par'); p[b, lo:hi] = centres[l]
            lo, hi = tpl.entry_range(obs.label, 'rad', 'par'); p[b, lo:hi] = radii[l]
        lo, hi = tpl.entry_range(problem.label, 'T', 'par'); p[b, lo:hi] = horizon_time
        lo, hi = tpl.entry_range(vehicle.label, 'splines_seg0', 'var')
        x0[b, lo:hi] = np.c_[np.linspace(start[0], goal[0], L),
                             np.linspace(start[1], goal[1], L)].reshape(-1, order='F')
    return problem, {'p': p, 'x0': x0}


def _set(tpl, arr, b, label, name, value, kind='par'):
    lo, hi = tpl.entry_range(label, name, kind)
    arr[b, lo:hi] = value


def _straight_line(tpl, x0, b, vehicle, start, goal, clamp=0):
    """Initial guess of the reference's `get_init_spline_value`: coefficients on the straight line
    (`holonomic.py:107-114`); `clamp` = degree repeats the end points so that the guess starts and
    ends at rest (`quadrotor.py:94-102`)."""
    L = len(vehicle.basis)
    lo, hi = tpl.entry_range(vehicle.label, 'splines_seg0', 'var')
    x0[b, lo:hi] = np.stack([np.r_[s * np.ones(clamp), np.linspace(s, g, L - 2 * clamp), g * np.ones(clamp)]
                             for s, g in zip(start, goal)]).reshape(-1)


def quadrotor_p2p(n_agents, knot_intervals=13, n_obs=5, seed=20240807 + 3, horizon_time=5.,
                  options=None):
    """Config 3: batch of independent 2-D Quadrotor point-to-point problems (flat outputs y, z of
    degree 4; thrust / pitch-rate rows cubic in the coefficients, `vehicles/quadrotor.py:48-62`)
    with `n_obs` circular obstacles moving at constant velocity."""
    rng = np.random.default_rng(seed)
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
    p = np.zeros((n_agents, tpl.n_par))
    x0 = np.zeros((n_agents, tpl.n_var))
    for b in range(n_agents):
        start = rng.uniform(-4.5, -3.5, size=2)
        goal = rng.uniform(3.5, 4.5, size=2)
        centres, radii = _place_obstacles(rng, n_obs, start, goal, 0.2, 0.5, 2.5, gap=0.45)
        vel = rng.uniform(-0.15, 0.15, size=(n_obs, 2))
        _set(tpl, p, b, vehicle.label, 'spl0', start)
        _set(tpl, p, b, vehicle.label, 'poseT', goal)
        for l, obs in enumerate(environment.obstacles):
            _set(tpl, p, b, obs.label, 'x', centres[l])
            _set(tpl, p, b, obs.label, 'v', vel[l])
            _set(tpl, p, b, obs.label, 'rad', radii[l])
        _set(tpl, p, b, problem.label, 'T', horizon_time)
        _straight_line(tpl, x0, b, vehicle, start, goal, clamp=vehicle.degree)
    # cold starts of this class: barrier parameter from 1 instead of 0.1 (82 -> 57 iterations on average, the
    # same agents converge)
    return problem, {'p': p, 'x0': x0, 'solver_options': {'mu_init': 1.0}}


def holonomic3d_p2p(n_agents, knot_intervals=15, n_obs=10, seed=20240807 + 5, horizon_time=12.,
                    options=None):
    """Config 5: batch of independent Holonomic3D(Sphere(0.1)) problems with `n_obs` static
    spheres, `hard_term_con=True`."""
    rng = np.random.default_rng(seed)
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
    p = np.zeros((n_agents, tpl.n_par))
    x0 = np.zeros((n_agents, tpl.n_var))
    for b in range(n_agents):
        start = rng.uniform(-2., -1., size=3)
        goal = rng.uniform(1., 2., size=3)
        centres, radii = [], []
        while len(centres) < n_obs:                 # spheres may overlap; they only keep clear of start/goal
            r, c = rng.uniform(0.15, 0.3), rng.uniform(-1.5, 1.5, size=3)
            if min(np.linalg.norm(c - start), np.linalg.norm(c - goal)) < r + 0.3:
                continue
            centres.append(c); radii.append(r)
        _set(tpl, p, b, vehicle.label, 'state0', start)
        _set(tpl, p, b, vehicle.label, 'poseT', goal)
        for l, obs in enumerate(environment.obstacles):
            _set(tpl, p, b, obs.label, 'x', centres[l])
            _set(tpl, p, b, obs.label, 'rad', radii[l])
        _set(tpl, p, b, problem.label, 'T', horizon_time)
        _straight_line(tpl, x0, b, vehicle, start, goal)
    return problem, {'p': p, 'x0': x0, 'solver_options': {}}


def formation_holonomic(n_agents, knot_intervals=10, seed=20240807 + 4, rho=1.0, horizon_time=10.,
                        with_obstacles=True, obstacles=None):
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
    radius = max(0.2, 0.2 * n_agents / (2 * np.pi))
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
    problem, updater, father = build_updx_template(
        vehicle, environment, 2, {'horizon_time': horizon_time})
    tpl = father.template
    lay = FormationLayout(tpl, vehicle, problem, updater, 2)
    ang = 2 * np.pi * np.arange(n_agents) / n_agents
    config = radius * np.c_[np.cos(ang), np.sin(ang)]          # position w.r.t. the fleet centre
    start_c, goal_c = np.array([0., -3.5]), np.array([0., 3.0])
    L = lay.L
    p = np.zeros((n_agents, tpl.n_par))
    x0 = np.zeros((n_agents, tpl.n_var))
    for l, obs in enumerate(environment.obstacles):
        chk, rad = obs.shape.get_checkpoints()
        lo = tpl.entry_range(obs.label, 'x', 'par')[0]; p[:, lo:lo + 2] = obs.signals['position'][:, -1]
        lo = tpl.entry_range(obs.label, 'v', 'par')[0]; p[:, lo:lo + 2] = obs.signals['velocity'][:, -1]
        lo = tpl.entry_range(obs.label, 'checkpoints', 'par')[0]; p[:, lo:lo + 2 * len(chk)] = np.reshape(chk, -1)
        lo = tpl.entry_range(obs.label, 'rad', 'par')[0]; p[:, lo:lo + len(rad)] = rad
    for b in range(n_agents):
        start, goal = start_c + config[b], goal_c + config[b]
        p[b, lay.p_rel:lay.p_rel + 2] = -config[b]                 # rel_pos_c = -configuration (fleet.py:93-98)
        p[b, lay.p_state0:lay.p_state0 + 2] = start
        p[b, lay.p_poseT:lay.p_poseT + 2] = goal
        p[b, lay.p_T] = horizon_time
        p[b, lay.p_rho] = rho
        x0[b, lay.x_spl:lay.x_spl + 2 * L] = np.c_[np.linspace(start[0], goal[0], L),
                                                   np.linspace(start[1], goal[1], L)].reshape(-1, order='F')
    return problem, updater, father, lay, {'p': p, 'x0': x0, 'nbr': circular_neighbors(n_agents)}


def rendezvous_holonomic(n_agents, knot_intervals=10, seed=20240807 + 6, rho=2.0, horizon_time=10.):
    """RendezVous ADMM workload in the shape of `examples/rendezvous_holonomic_export.py:31-53` scaled to the fleet
    size: `n_agents` Holonomic vehicles start on a large circle and agree on a meeting point in a regular-polygon
    configuration (circular interconnection), one rectangular obstacle.  Returns (problem, updater, father,
    layout, P) like `formation_holonomic`."""
    from .shapes import Rectangle
    from .rendezvous import build_rendezvous_template, RendezVousLayout
    from .formation import circular_neighbors
    rng = np.random.default_rng(seed)
    vehicle = Holonomic(shapes=Circle(0.1), options={'room_constraints': None})
    vehicle.define_knots(knot_intervals=knot_intervals)
    vehicle.set_initial_conditions([0., 0.])
    vehicle.set_terminal_conditions([0., 0.])
    radius = max(0.2, 0.2 * n_agents / (2 * np.pi))
    span = 3. + 2. * radius
    environment = Environment(room={'shape': Square(2. * span + 4.)})
    environment.add_obstacle(Obstacle({'position': [0.3 * span, -0.2 * span]}, shape=Rectangle(width=1.5, height=0.2)))
    problem, updater, father = build_rendezvous_template(vehicle, environment, 2, {'horizon_time': horizon_time})
    tpl = father.template
    lay = RendezVousLayout(tpl, vehicle, problem, updater, 2)
    ang = 2 * np.pi * np.arange(n_agents) / n_agents
    config = radius * np.c_[np.cos(ang), np.sin(ang)]
    starts = span * np.c_[np.cos(ang + 0.4), np.sin(ang + 0.4)] * (0.7 + 0.3 * rng.random((n_agents, 1)))
    L = len(vehicle.basis)
    p = np.zeros((n_agents, tpl.n_par))
    x0 = np.zeros((n_agents, tpl.n_var))
    for obs in environment.obstacles:
        chk, rad = obs.shape.get_checkpoints()
        lo = tpl.entry_range(obs.label, 'x', 'par')[0]; p[:, lo:lo + 2] = obs.signals['position'][:, -1]
        lo = tpl.entry_range(obs.label, 'checkpoints', 'par')[0]; p[:, lo:lo + 2 * len(chk)] = np.reshape(chk, -1)
        lo = tpl.entry_range(obs.label, 'rad', 'par')[0]; p[:, lo:lo + len(rad)] = rad
    for b in range(n_agents):
        goal = config[b]                                                # first guess: meet at the origin
        p[b, lay.p_rel:lay.p_rel + 2] = -config[b]                      # rel_pos_c = -configuration (fleet.py:93-98)
        p[b, lay.p_state0:lay.p_state0 + 2] = starts[b]
        p[b, lay.p_poseT:lay.p_poseT + 2] = goal
        p[b, lay.p_T] = horizon_time
        p[b, lay.p_rho] = rho
        x0[b, lay.x_traj:lay.x_traj + 2 * L] = np.c_[np.linspace(starts[b, 0], goal[0], L),
                                                     np.linspace(starts[b, 1], goal[1], L)].reshape(-1, order='F')
        x0[b, lay.x_spl:lay.x_spl + 2] = goal                           # conT0 (`point2point.py:391-399`)
    return problem, updater, father, lay, {'p': p, 'x0': x0, 'nbr': circular_neighbors(n_agents)}
