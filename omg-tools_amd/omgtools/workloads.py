"""Seeded synthetic workloads of BASELINE.json's configs from committed problem bundles.

A bundle (`omgtools/data/<name>.npz`, written by `tools/generate_workload_bundles.py`) is what the device needs of one
problem class and nothing else: the flat NLP template with its block table (`NLPTemplate.to_npz`: the role of the
generated nlp.so of the reference's export, `export/export.py:236-262`) and the few facts the receding-horizon loop reads
off the reference's objects -- labels, the trajectory basis, horizon and knot time, which spline variables a knot
crossing shifts (`basics/optilayer.py:470-490`) and where their multipliers move (`batch.dual_shift_perm`).  `bench.py`
and the GPU tier load these; none of the front-end modules (vehicles / environment / problems / shapes / execution) is
imported on that path.  `omgtools.scenarios` builds the same workloads through the front end (any knot count, any number of
obstacles) with the parameter generators below; `tests/test_workload_bundles.py` checks bundle == front end, array for
array.
"""
import json
import os

import numpy as np

from .splines import BSplineBasis
from .template import NLPTemplate

DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data')


# ---- what the loops read off a problem: plain views ---------------------------------------------------------------
class _Labelled(object):
    def __init__(self, label):
        self.label = label


class VehicleView(_Labelled):
    def __init__(self, label, knots, degree, n_dim, n_spl):
        _Labelled.__init__(self, label)
        self.basis = BSplineBasis(np.asarray(knots, float), int(degree))
        self.degree, self.n_dim, self.n_spl = int(degree), int(n_dim), int(n_spl)


class FatherView(object):
    """`father.template`, `father.shifted_entries()` and the multiplier map of a knot crossing, from a bundle."""

    def __init__(self, template, shifted, perm):
        self.template = template
        self._shifted = shifted          # {rule: [(label, name, knots, degree)]}, rule = 'seg' / 'every'
        self.dual_perm = np.asarray(perm, dtype=np.int64)

    def shifted_entries(self, seg_shift=None, every_spline=False):
        return [(label, name, {'basis': BSplineBasis(np.asarray(k, float), int(d)), 'init': None})
                for label, name, k, d in self._shifted['every' if every_spline else 'seg']]


class _Env(object):
    def __init__(self, labels):
        self.obstacles = [_Labelled(l) for l in labels]


class ProblemView(_Labelled):
    def __init__(self, meta, template):
        _Labelled.__init__(self, meta['problem_label'])
        v = meta['vehicle']
        self.vehicles = [VehicleView(v['label'], v['knots'], v['degree'], v['n_dim'], v['n_spl'])]
        self.options = {'horizon_time': float(meta['horizon_time'])}
        self.knot_time = float(meta['knot_time'])
        self.environment = _Env(meta['obstacle_labels'])
        self.father = FatherView(template, meta['shifted'], meta['dual_perm'])
        self.meta = meta


def describe(problem, father=None, extra=None):
    """The bundle's description of an initialised front-end problem (used by the generator and by the equality test)."""
    from .batch import dual_shift_perm
    father = father if father is not None else problem.father
    veh = problem.vehicles[0]
    shifted = {}
    for rule, every in (('seg', False), ('every', True)):
        shifted[rule] = [[label, name, np.asarray(spl['basis'].knots, float).tolist(), int(spl['basis'].degree)]
                         for label, name, spl in father.shifted_entries(every_spline=every)]
    meta = dict(problem_label=problem.label, horizon_time=float(problem.options['horizon_time']), knot_time=float(problem.knot_time),
                vehicle=dict(label=veh.label, knots=np.asarray(veh.basis.knots, float).tolist(), degree=int(veh.basis.degree),
                             n_dim=int(veh.n_dim), n_spl=int(veh.n_spl)),
                obstacle_labels=[o.label for o in problem.environment.obstacles], shifted=shifted,
                dual_perm=dual_shift_perm(father).tolist())
    meta.update(extra or {})
    return meta


def save_bundle(path, template, meta):
    template.to_npz(path, meta=np.array(json.dumps(meta)))
    return path


def load_bundle(name):
    path = name if os.path.isabs(name) else os.path.join(DATA_DIR, name + '.npz')
    if not os.path.exists(path):
        raise IOError('no workload bundle %s (tools/generate_workload_bundles.py writes them)' % path)
    tpl = NLPTemplate.from_npz(path)
    meta = json.loads(str(np.load(path)['meta']))
    return tpl, meta


# ---- parameter generators (shared with omgtools.scenarios: the same numbers from either side) -------------------
def place_obstacles(rng, n_obs, start, goal, r_lo, r_hi, box, clearance=0.3, gap=0.25):
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


def _set(tpl, arr, b, label, name, value, kind='par'):
    lo, hi = tpl.entry_range(label, name, kind)
    arr[b, lo:hi] = value


def straight_line(tpl, x0, b, veh_label, L, start, goal, clamp=0):
    """Initial guess of the reference's `get_init_spline_value`: coefficients on the straight line
    (`holonomic.py:107-114`); `clamp` = degree repeats the end points so that the guess starts and
    ends at rest (`quadrotor.py:94-102`)."""
    lo, hi = tpl.entry_range(veh_label, 'splines_seg0', 'var')
    x0[b, lo:hi] = np.stack([np.r_[s * np.ones(clamp), np.linspace(s, g, L - 2 * clamp), g * np.ones(clamp)]
                             for s, g in zip(start, goal)]).reshape(-1)


def fill_holonomic_p2p(tpl, veh_label, problem_label, obstacle_labels, L, n_agents, seed, horizon_time, gap=0.25):
    """Config 2: start ~U([-2,-1]^2), goal ~U([1,2]^2), static circles r ~U(0.2,0.4) around the origin."""
    rng = np.random.default_rng(seed)
    n_obs = len(obstacle_labels)
    p = np.zeros((n_agents, tpl.n_par))
    x0 = np.zeros((n_agents, tpl.n_var))
    for b in range(n_agents):
        start = rng.uniform(-2., -1., size=2)
        goal = rng.uniform(1., 2., size=2)
        centres, radii = place_obstacles(rng, n_obs, start, goal, 0.2, 0.4, 0.8, gap=gap)
        _set(tpl, p, b, veh_label, 'state0', start)
        _set(tpl, p, b, veh_label, 'poseT', goal)
        for l, ol in enumerate(obstacle_labels):
            _set(tpl, p, b, ol, 'x', centres[l])
            _set(tpl, p, b, ol, 'rad', radii[l])
        _set(tpl, p, b, problem_label, 'T', horizon_time)
        lo, hi = tpl.entry_range(veh_label, 'splines_seg0', 'var')
        x0[b, lo:hi] = np.c_[np.linspace(start[0], goal[0], L),
                             np.linspace(start[1], goal[1], L)].reshape(-1, order='F')
    return {'p': p, 'x0': x0}


def fill_quadrotor_p2p(tpl, veh_label, problem_label, obstacle_labels, L, degree, n_agents, seed, horizon_time):
    """Config 3: start ~U([-4.5,-3.5]^2), goal ~U([3.5,4.5]^2), circles r ~U(0.2,0.5) moving at ~U([-0.15,0.15]^2)."""
    rng = np.random.default_rng(seed)
    n_obs = len(obstacle_labels)
    p = np.zeros((n_agents, tpl.n_par))
    x0 = np.zeros((n_agents, tpl.n_var))
    for b in range(n_agents):
        start = rng.uniform(-4.5, -3.5, size=2)
        goal = rng.uniform(3.5, 4.5, size=2)
        centres, radii = place_obstacles(rng, n_obs, start, goal, 0.2, 0.5, 2.5, gap=0.45)
        vel = rng.uniform(-0.15, 0.15, size=(n_obs, 2))
        _set(tpl, p, b, veh_label, 'spl0', start)
        _set(tpl, p, b, veh_label, 'poseT', goal)
        for l, ol in enumerate(obstacle_labels):
            _set(tpl, p, b, ol, 'x', centres[l])
            _set(tpl, p, b, ol, 'v', vel[l])
            _set(tpl, p, b, ol, 'rad', radii[l])
        _set(tpl, p, b, problem_label, 'T', horizon_time)
        straight_line(tpl, x0, b, veh_label, L, start, goal, clamp=degree)
    # cold starts of this class: barrier parameter from 1 instead of 0.1 (82 -> 57 iterations on average, the
    # same agents converge); no second-order correction: with the KKT store in the slab (mode 1) the extra forward
    # substitution costs more than the 4 % of the iterations it saves (4096 agents: 17.2 k cold solves/s without, 14.0 k with;
    # the 3-D class, mode 3, gains 15 % of its iterations and keeps it)
    return {'p': p, 'x0': x0, 'solver_options': {'mu_init': 1.0, 'max_soc': 0}}


def fill_holonomic3d_p2p(tpl, veh_label, problem_label, obstacle_labels, L, n_agents, seed, horizon_time):
    """Config 5: start ~U([-2,-1]^3), goal ~U([1,2]^3), static spheres r ~U(0.15,0.3) in [-1.5,1.5]^3 (they may overlap)."""
    rng = np.random.default_rng(seed)
    n_obs = len(obstacle_labels)
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
        _set(tpl, p, b, veh_label, 'state0', start)
        _set(tpl, p, b, veh_label, 'poseT', goal)
        for l, ol in enumerate(obstacle_labels):
            _set(tpl, p, b, ol, 'x', centres[l])
            _set(tpl, p, b, ol, 'rad', radii[l])
        _set(tpl, p, b, problem_label, 'T', horizon_time)
        straight_line(tpl, x0, b, veh_label, L, start, goal)
    # (max_soc 4 -- IPOPT's default count -- was measured on this class in round 6: 36.3 -> 32.0 cold iterations on the host build, but on
    # the device the further corrections cost more than the iterations they save -- a blocked second solve out of the slab each:
    # 7.6 k -> 7.2 k cold solves/s at 8192 agents, 3.9 k -> 2.7 k at 1024: the default of one correction stays)
    return {'p': p, 'x0': x0, 'solver_options': {}}


# ---- the workloads of bench.py ------------------------------------------------------------------------------------
def _p2p(name, n_agents, seed, fill, **kw):
    tpl, meta = load_bundle(name)
    problem = ProblemView(meta, tpl)
    veh = problem.vehicles[0]
    args = dict(veh_label=veh.label, problem_label=problem.label, obstacle_labels=meta['obstacle_labels'], L=len(veh.basis),
                n_agents=n_agents, seed=seed, horizon_time=problem.options['horizon_time'])
    args.update(kw)
    return problem, fill(tpl, **args)


def holonomic_p2p(n_agents, seed=20240807 + 2, gap=0.25):
    """Config 2 from its bundle: (problem view, {'p', 'x0'})."""
    return _p2p('holonomic_p2p_k11_o3', n_agents, seed, fill_holonomic_p2p, gap=gap)


def quadrotor_p2p(n_agents, seed=20240807 + 3):
    tpl, meta = load_bundle('quadrotor_p2p_k13_o5')
    return _p2p('quadrotor_p2p_k13_o5', n_agents, seed, fill_quadrotor_p2p, degree=int(meta['vehicle']['degree']))


def holonomic3d_p2p(n_agents, seed=20240807 + 5):
    return _p2p('holonomic3d_p2p_k15_o10', n_agents, seed, fill_holonomic3d_p2p)


def _fleet(name):
    from .consensus import FormationLayout, RendezVousLayout
    tpl, meta = load_bundle(name)
    problem = ProblemView(meta, tpl)
    updater = _Labelled(meta['updater_label'])
    cls = RendezVousLayout if meta.get('layout_class') == 'rendezvous' else FormationLayout
    lay = cls(tpl, problem.vehicles[0], problem, updater, 2)
    return tpl, meta, problem, updater, lay


def formation_holonomic(n_agents, rho=1.0):
    """Config 4 from its bundle: (problem view, updater view, father view, layout, {'p', 'x0', 'nbr'}).  (The room of
    the x-update template grows with the fleet: one bundle per fleet size; a fleet size without a committed bundle is built
    by the front end -- `omgtools.scenarios`, the same numbers, tests/test_workload_bundles.py.)"""
    from .consensus import circular_neighbors
    if not have('formation_holonomic_k10_%d' % n_agents):
        from . import scenarios
        return scenarios.formation_holonomic(n_agents, rho=rho)
    tpl, meta, problem, updater, lay = _fleet('formation_holonomic_k10_%d' % n_agents)
    P = fill_formation(tpl, lay, meta['obstacles'], n_agents, float(problem.options['horizon_time']), rho)
    P['nbr'] = circular_neighbors(n_agents)
    return problem, updater, problem.father, lay, P


def rendezvous_holonomic(n_agents, seed=20240807 + 6, rho=2.0):
    from .consensus import circular_neighbors
    if not have('rendezvous_holonomic_k10_%d' % n_agents):
        from . import scenarios
        return scenarios.rendezvous_holonomic(n_agents, seed=seed, rho=rho)
    tpl, meta, problem, updater, lay = _fleet('rendezvous_holonomic_k10_%d' % n_agents)
    P = fill_rendezvous(tpl, lay, meta['obstacles'], n_agents, seed, float(problem.options['horizon_time']), rho)
    P['nbr'] = circular_neighbors(n_agents)
    return problem, updater, problem.father, lay, P


def agv_loop(n_agents):
    """A class with lifted auxiliaries (template.py `_append_lifted`): the body of the reference's `examples/p2p_agv.py` with
    a fixed horizon -- 381 variables (278 of them lifted), 2234 rows -- and the thirteen solves of its closed loop (the solve before the
    loop and twelve updates of the reference's Simulator, each from the reference's warm start) tiled to `n_agents`:
    (template, {'p', 'x0', 'lbg', 'ubg', 'iters_host'}).  The bundle is written from the fixtures the reference's own classes
    produced on `omgx_shim` (tests/golden/generate_shim_fixtures.py; tests/test_workload_bundles.py compares them)."""
    path = os.path.join(DATA_DIR, 'agv_fixedT_k5.npz')
    tpl = NLPTemplate.from_npz(path)
    d = np.load(path)
    idx = np.arange(n_agents) % len(d['loop_p'])
    return tpl, {'p': d['loop_p'][idx], 'x0': d['loop_x0'][idx], 'lbg': d['loop_lbg'], 'ubg': d['loop_ubg'],
                 'iters_host': d['loop_iters'][idx]}


def have(name):
    return os.path.exists(os.path.join(DATA_DIR, name + '.npz'))


def formation_radius(n_agents):
    return max(0.2, 0.2 * n_agents / (2 * np.pi))


def fill_formation(tpl, lay, obstacles, n_agents, horizon_time, rho):
    """Parameters of the formation fleet (`examples/formation_holonomic.py:22-57` scaled to the fleet size).  obstacles:
    [{'label', 'position', 'velocity', 'checkpoints', 'rad'}] -- what the x-update template reads per obstacle."""
    radius = formation_radius(n_agents)
    ang = 2 * np.pi * np.arange(n_agents) / n_agents
    config = radius * np.c_[np.cos(ang), np.sin(ang)]          # position w.r.t. the fleet centre
    start_c, goal_c = np.array([0., -3.5]), np.array([0., 3.0])
    L = lay.L
    p = np.zeros((n_agents, tpl.n_par))
    x0 = np.zeros((n_agents, tpl.n_var))
    for obs in obstacles:
        chk, rad = np.asarray(obs['checkpoints'], float), np.asarray(obs['rad'], float)
        lo = tpl.entry_range(obs['label'], 'x', 'par')[0]; p[:, lo:lo + 2] = obs['position']
        lo = tpl.entry_range(obs['label'], 'v', 'par')[0]; p[:, lo:lo + 2] = obs['velocity']
        lo = tpl.entry_range(obs['label'], 'checkpoints', 'par')[0]; p[:, lo:lo + 2 * len(chk)] = np.reshape(chk, -1)
        lo = tpl.entry_range(obs['label'], 'rad', 'par')[0]; p[:, lo:lo + len(rad)] = rad
    for b in range(n_agents):
        start, goal = start_c + config[b], goal_c + config[b]
        p[b, lay.p_rel:lay.p_rel + 2] = -config[b]                 # rel_pos_c = -configuration (fleet.py:93-98)
        p[b, lay.p_state0:lay.p_state0 + 2] = start
        p[b, lay.p_poseT:lay.p_poseT + 2] = goal
        p[b, lay.p_T] = horizon_time
        p[b, lay.p_rho] = rho
        x0[b, lay.x_spl:lay.x_spl + 2 * L] = np.c_[np.linspace(start[0], goal[0], L),
                                                   np.linspace(start[1], goal[1], L)].reshape(-1, order='F')
    return {'p': p, 'x0': x0}


def fill_rendezvous(tpl, lay, obstacles, n_agents, seed, horizon_time, rho):
    """Parameters of the rendez-vous fleet (`examples/rendezvous_holonomic_export.py:31-53` scaled to the fleet size)."""
    rng = np.random.default_rng(seed)
    radius = formation_radius(n_agents)
    span = 3. + 2. * radius
    ang = 2 * np.pi * np.arange(n_agents) / n_agents
    config = radius * np.c_[np.cos(ang), np.sin(ang)]
    starts = span * np.c_[np.cos(ang + 0.4), np.sin(ang + 0.4)] * (0.7 + 0.3 * rng.random((n_agents, 1)))
    L = len(lay.basis)
    p = np.zeros((n_agents, tpl.n_par))
    x0 = np.zeros((n_agents, tpl.n_var))
    for obs in obstacles:
        chk, rad = np.asarray(obs['checkpoints'], float), np.asarray(obs['rad'], float)
        lo = tpl.entry_range(obs['label'], 'x', 'par')[0]; p[:, lo:lo + 2] = obs['position']
        lo = tpl.entry_range(obs['label'], 'checkpoints', 'par')[0]; p[:, lo:lo + 2 * len(chk)] = np.reshape(chk, -1)
        lo = tpl.entry_range(obs['label'], 'rad', 'par')[0]; p[:, lo:lo + len(rad)] = rad
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
    return {'p': p, 'x0': x0}
