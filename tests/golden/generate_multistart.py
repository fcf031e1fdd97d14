"""Generates tests/golden/sol_cfg2_ms.npz and tests/golden/sol_mpc_cfg2.npz: parity fixtures whose every number
comes from scipy SLSQP (dense SQP with its own QP solver and line search) on the restated NLP
(oracle/nlp_numpy.py, pinned to the reference's construct code by tests/golden/nlp_*.npz).  Neither the product
kernel nor its numpy mirror (oracle/ipm_numpy.py) chooses a basin or a starting point here.

  sol_cfg2_ms.npz   64 seeded agents of config 2 (Holonomic, K = 11, 3 circles), multi-start: SLSQP from the
                    reference's initial guess (`get_init_spline_value`, hyperplanes zero), from that guess bent
                    sideways by +-0.5, +-1, +-1.5 and +-2.5 m at mid-course (the other sides of the obstacles) and from twelve
                    seeded perturbations of the guess (hyperplane normals drawn at random instead of zero: from the
                    degenerate all-zero normals the first step of any method decides the side).  Every converged
                    start is stored (`x_all`, `f_all`, `ok_all`): the product, started from the reference's guess,
                    must land in one of these minima.
  sol_cfg3_ms.npz   8 agents of the Quadrotor class (K = 13, 5 moving circles), 9 starts each (guess, four bends, four
                    random hyperplane directions); SLSQP stops on these sizes with exit code 8 ('positive directional
                    derivative': no step improves the objective any more), accepted with the feasibility bound 1e-7.
  sol_cfg5_ms.npz   8 agents of the Holonomic3D class (K = 15, 10 spheres), 25 starts each + twelve more bent along both
                    perpendiculars of start -> goal (BENDS_3D, `extend_cfg5`: round 4).
  sol_mpc_cfg2.npz  the warm-started path: 8 agents, 12 receding-horizon steps (one knot crossing) of the protocol of
                    bench.py run on the host (oracle port as the solver object); the inputs of every step -- p_k, the
                    shifted plan x0_k, the shifted multipliers lam_k -- are dumped, and the NLP of every step is
                    solved by SLSQP from x0_k.  A warm-started product solve from the dumped inputs must return
                    SLSQP's solution.

Run from the repository root:  python tests/golden/generate_multistart.py [cfg2] [cfg3] [cfg5] [mpc]   (cfg2 5 min, cfg3 7 min, cfg5 30 min on 7 cores)"""
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'omg-tools_amd'))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
HERE = os.path.dirname(os.path.abspath(__file__))
BENDS = (1.0, -1.0, 2.5, -2.5, 0.5, -0.5, 1.5, -1.5)
N_RANDOM = 12
_STATE = {}


def _scenario(n, name='holonomic_p2p'):
    import omgtools.backend as be
    from omgtools import scenarios
    be.create_nlp = lambda tpl, opt, name='': (None, 0.)
    return getattr(scenarios, name)(n)


def _init(n, name='holonomic_p2p', bends=None, n_random=None):
    from oracle.nlp_numpy import NumpyNLP
    problem, P = _scenario(n, name)
    _STATE.update(bends=BENDS if bends is None else bends, n_random=N_RANDOM if n_random is None else n_random, name=name)
    tpl = problem.father.template
    _STATE.update(tpl=tpl, P=P, nlp=NumpyNLP(tpl), problem=problem)


def bent(tpl, problem, x0, s):
    """BatchP2P._bent on one agent: the straight-line guess moved sideways by s sin^2(pi k / (L - 1)) metres."""
    veh = problem.vehicles[0]
    L, ns = len(veh.basis), veh.n_spl
    lo = tpl.entry_range(veh.label, 'splines_seg0', 'var')[0]
    c = x0[lo:lo + ns * L].reshape(ns, L)
    d = c[:, -1] - c[:, 0]
    d = d / max(np.linalg.norm(d), 1e-12)
    nrm = np.zeros(ns)
    nrm[0], nrm[1] = -d[1], d[0]                     # perpendicular to start -> goal in the x-y plane
    nrm /= max(np.linalg.norm(nrm), 1e-12)
    out = x0.copy()
    out[lo:lo + ns * L] += (s * nrm[:, None] * (np.sin(np.linspace(0., 1., L) * np.pi) ** 2)[None, :]).reshape(-1)
    return out


# 3-D classes: the straight-line guess bent along both perpendiculars of start -> goal (s_y in the x-y plane as `bent`, s_z
# along the binormal): the sides of a sphere are not only left and right
BENDS_3D = ((0, 1.0), (0, -1.0), (0, 2.5), (0, -2.5), (1, 1), (1, -1), (-1, 1), (-1, -1), (2.5, 2.5), (2.5, -2.5), (-2.5, 2.5), (-2.5, -2.5))


def bent3(tpl, problem, x0, s_y, s_z):
    veh = problem.vehicles[0]
    L, ns = len(veh.basis), veh.n_spl
    lo = tpl.entry_range(veh.label, 'splines_seg0', 'var')[0]
    c = x0[lo:lo + ns * L].reshape(ns, L)
    d = c[:, -1] - c[:, 0]
    d = d / max(np.linalg.norm(d), 1e-12)
    n1 = np.zeros(ns)
    n1[0], n1[1] = -d[1], d[0]
    n1 /= max(np.linalg.norm(n1), 1e-12)
    n2 = np.cross(d, n1)
    n2 /= max(np.linalg.norm(n2), 1e-12)
    out = x0.copy()
    out[lo:lo + ns * L] += ((s_y * n1 + s_z * n2)[:, None] * (np.sin(np.linspace(0., 1., L) * np.pi) ** 2)[None, :]).reshape(-1)
    return out


def _solve_start3(job):
    from slsqp_reference import solve_slsqp
    b, k = job
    tpl, P, nlp, problem = _STATE['tpl'], _STATE['P'], _STATE['nlp'], _STATE['problem']
    t0 = time.time()
    x, f, ok = solve_slsqp(nlp, tpl, bent3(tpl, problem, P['x0'][b], *BENDS_3D[k]), P['p'][b], maxiter=1500, accept=(0, 8), viol_tol=1e-7)
    return b, k, x, f, ok, time.time() - t0


def extend_cfg5(n, workers, out='sol_cfg5_ms.npz'):
    """Twelve more starts per agent for the 3-D class (BENDS_3D), merged into the existing fixture: every distinct new
    minimum is appended to x_min / f_min (first_start = 1000 + index of the bend)."""
    t0 = time.time()
    d = dict(np.load(os.path.join(HERE, out)))
    jobs = [(b, k) for b in range(n) for k in range(len(BENDS_3D))]
    with ProcessPoolExecutor(workers, initializer=_init, initargs=(n, 'holonomic3d_p2p', (1.0, -1.0, 2.5, -2.5), 20)) as ex:
        res = list(ex.map(_solve_start3, jobs, chunksize=1))
    lo, hi = [int(v) for v in d['spl']]
    mins = [[(d['x_min'][b, i], float(d['f_min'][b, i]), int(d['first_start'][b, i])) for i in range(int(d['n_min'][b]))] for b in range(n)]
    added = 0
    for b, k, x, f, ok, s in res:
        if not ok:
            continue
        if any(abs(f - fj) < 1e-6 * (1 + abs(fj)) and np.abs(x[lo:hi] - xj[lo:hi]).max() < 1e-3 for xj, fj, _ in mins[b]):
            continue
        mins[b].append((x, f, 1000 + k)); added += 1
    K = max(len(m) for m in mins)
    nv = d['x_min'].shape[2]
    x_min = np.full((n, K, nv), np.nan); f_min = np.full((n, K), np.nan); first = np.full((n, K), -1, dtype=np.int32)
    for b, m in enumerate(mins):
        for i, (x, f, k) in enumerate(m):
            x_min[b, i], f_min[b, i], first[b, i] = x, f, k
    d.update(x_min=x_min, f_min=f_min, first_start=first, n_min=np.array([len(m) for m in mins], dtype=np.int32), bends_3d=np.array(BENDS_3D))
    np.savez_compressed(os.path.join(HERE, out), **d)
    print('%s: %d more starts per agent, %d new minima, %d..%d distinct minima per agent now, %.0f s'
          % (out, len(BENDS_3D), added, d['n_min'].min(), d['n_min'].max(), time.time() - t0))


def starts_of(b):
    tpl, P, problem = _STATE['tpl'], _STATE['P'], _STATE['problem']
    x0 = P['x0'][b]
    out = [x0] + [bent(tpl, problem, x0, s) for s in _STATE['bends']]
    rng = np.random.default_rng(9000 + b)
    hyp = [k for k in tpl.var_layout if k[1].startswith('a_')]
    for _ in range(_STATE['n_random']):
        x = x0.copy()
        for key in hyp:
            lo, rows, cols = tpl.var_layout[key]
            nrm = rng.normal(size=cols)
            nrm /= np.linalg.norm(nrm)
            x[lo:lo + rows * cols] = 0.5 * np.repeat(nrm, rows)         # a constant unit direction, half length
        out.append(x)
    return out


def _solve_start(job):
    from slsqp_reference import solve_slsqp
    b, k = job
    tpl, P, nlp = _STATE['tpl'], _STATE['P'], _STATE['nlp']
    t0 = time.time()
    big = _STATE['name'] != 'holonomic_p2p'      # (the larger classes: SLSQP ends with 'positive directional derivative')
    x, f, ok = solve_slsqp(nlp, tpl, starts_of(b)[k], P['p'][b], maxiter=1500 if big else 800,
                           accept=(0, 8) if big else (0,), viol_tol=1e-7 if big else 1e-8)
    return b, k, x, f, ok, time.time() - t0


def distinct_minima(x_all, f_all, ok_all, lo, hi):
    """The converged starts of every agent with duplicates removed (same objective to 1e-6 relative and the same
    trajectory coefficients to 1e-3): x_min [n, K, n_var] (NaN padded), f_min [n, K], n_min [n], and the index of the
    start that found each one first."""
    n, ns, nv = x_all.shape
    keep = []
    for b in range(n):
        mine = []
        for k in range(ns):
            if not ok_all[b, k]:
                continue
            if any(abs(f_all[b, k] - f_all[b, j]) < 1e-6 * (1 + abs(f_all[b, j])) and
                   np.abs(x_all[b, k, lo:hi] - x_all[b, j, lo:hi]).max() < 1e-3 for j in mine):
                continue
            mine.append(k)
        keep.append(mine)
    K = max(len(m) for m in keep)
    x_min = np.full((n, K, nv), np.nan); f_min = np.full((n, K), np.nan); first = np.full((n, K), -1, dtype=np.int32)
    for b, mine in enumerate(keep):
        for i, k in enumerate(mine):
            x_min[b, i], f_min[b, i], first[b, i] = x_all[b, k], f_all[b, k], k
    return x_min, f_min, np.array([len(m) for m in keep], dtype=np.int32), first


def run_cfg2(n, workers, name='holonomic_p2p', out='sol_cfg2_ms.npz', bends=BENDS, n_random=N_RANDOM):
    t0 = time.time()
    n_start = 1 + len(bends) + n_random
    jobs = [(b, k) for b in range(n) for k in range(n_start)]
    with ProcessPoolExecutor(workers, initializer=_init, initargs=(n, name, bends, n_random)) as ex:
        res = list(ex.map(_solve_start, jobs, chunksize=1))
    _init(n, name, bends, n_random)
    tpl, P, problem = _STATE['tpl'], _STATE['P'], _STATE['problem']
    lo, hi = tpl.entry_range(problem.vehicles[0].label, 'splines_seg0', 'var')
    x_all = np.zeros((n, n_start, tpl.n_var)); f_all = np.full((n, n_start), np.nan); ok_all = np.zeros((n, n_start), dtype=bool)
    secs = np.zeros((n, n_start))
    for b, k, x, f, ok, s in res:
        x_all[b, k], f_all[b, k], ok_all[b, k], secs[b, k] = x, f, ok, s
    x_min, f_min, n_min, first = distinct_minima(x_all, f_all, ok_all, lo, hi)
    np.savez_compressed(os.path.join(HERE, out), p=P['p'][:n], x0=P['x0'][:n], x_min=x_min, f_min=f_min,
                        n_min=n_min, first_start=first, f_all=f_all, ok_all=ok_all, spl=np.array([lo, hi]), n_var=tpl.n_var,
                        n_con=tpl.n_con, n_par=tpl.n_par, seconds=secs, bends=np.array(bends), n_random=n_random)
    print('%s: %d agents x %d starts, %d converged, %d..%d distinct minima per agent, %.0f s'
          % (out, n, n_start, int(ok_all.sum()), n_min.min(), n_min.max(), time.time() - t0))


# ---- warm-started path ------------------------------------------------------------------------------------------
def _solve_step(job):
    from slsqp_reference import solve_slsqp
    tpl, nlp = _STATE['tpl'], _STATE['nlp']
    k, b, x0, p = job
    t0 = time.time()
    x, f, ok = solve_slsqp(nlp, tpl, x0, p, maxiter=800)
    return k, b, x, f, ok, time.time() - t0


def run_mpc(n, steps, workers):
    from omgtools.batch import BatchP2P
    from oracle import port_binding
    t0 = time.time()
    _init(n)
    tpl, P, problem = _STATE['tpl'], _STATE['P'], _STATE['problem']
    mpc = BatchP2P(problem, P, ops=port_binding, options=dict(tol=1e-6, max_iter=500))
    mpc.pool = None                                   # the step glue in numpy (so that the inputs of the solve can be taken)
    mpc.solve_cold(bends=())
    assert (np.asarray(mpc.status) == 0).all()
    dump = dict(p=[], x0=[], lam=[], crossed=[], x_port=[])
    solve = mpc._solve

    def tap(warm, *a, **kw):
        dump['p'].append(np.array(mpc.p)); dump['x0'].append(np.array(mpc.x)); dump['lam'].append(np.array(mpc.lam))
        solve(warm, *a, **kw)
        dump['x_port'].append(np.array(mpc.x))
    mpc._solve = tap
    for k in range(steps):
        dump['crossed'].append(bool(mpc.step()))
        assert (np.asarray(mpc.status) == 0).all(), k
    jobs = [(k, b, dump['x0'][k][b], dump['p'][k][b]) for k in range(steps) for b in range(n)]
    with ProcessPoolExecutor(workers, initializer=_init, initargs=(n,)) as ex:
        res = list(ex.map(_solve_step, jobs, chunksize=1))
    x = np.zeros((steps, n, tpl.n_var)); f = np.zeros((steps, n)); ok = np.zeros((steps, n), dtype=bool)
    for k, b, xs, fs, oks, s in res:
        x[k, b], f[k, b], ok[k, b] = xs, fs, oks
    lo, hi = tpl.entry_range(problem.vehicles[0].label, 'splines_seg0', 'var')
    np.savez_compressed(os.path.join(HERE, 'sol_mpc_cfg2.npz'), p=np.array(dump['p']), x0=np.array(dump['x0']),
                        lam=np.array(dump['lam']), crossed=np.array(dump['crossed']), x=x, f=f, ok=ok, x_port=np.array(dump['x_port']),
                        spl=np.array([lo, hi]), n_var=tpl.n_var, n_con=tpl.n_con, n_par=tpl.n_par)
    print('sol_mpc_cfg2.npz: %d steps x %d agents, %d converged, crossings at %s, %.0f s'
          % (steps, n, int(ok.sum()), np.nonzero(dump['crossed'])[0].tolist(), time.time() - t0))


def run_mpc_class(n, steps, workers, name, out):
    """`run_mpc` for another vehicle class (round 4: sol_mpc_cfg3.npz -- 4 Quadrotor agents, 6 steps with one knot crossing,
    five moving circles): the step inputs dumped from a host run of the bench protocol, every step's NLP solved by SLSQP
    from the shifted plan (SLSQP's exit code 8 accepted with the feasibility bound, as for sol_cfg3_ms.npz)."""
    from omgtools.batch import BatchP2P
    from oracle import port_binding
    t0 = time.time()
    _init(n, name)
    tpl, P, problem = _STATE['tpl'], _STATE['P'], _STATE['problem']
    mpc = BatchP2P(problem, P, ops=port_binding, options=dict(P.get('solver_options', {}), tol=1e-6, max_iter=500))
    mpc.pool = None
    mpc.solve_cold(bends=())
    assert (np.asarray(mpc.status) == 0).all(), mpc.status
    dump = dict(p=[], x0=[], lam=[], crossed=[], x_port=[])
    solve = mpc._solve

    def tap(warm, *a, **kw):
        dump['p'].append(np.array(mpc.p)); dump['x0'].append(np.array(mpc.x)); dump['lam'].append(np.array(mpc.lam))
        solve(warm, *a, **kw)
        dump['x_port'].append(np.array(mpc.x))
    mpc._solve = tap
    for k in range(steps):
        dump['crossed'].append(bool(mpc.step()))
        assert (np.asarray(mpc.status) == 0).all(), k
    jobs = [(k, b, dump['x0'][k][b], dump['p'][k][b]) for k in range(steps) for b in range(n)]
    with ProcessPoolExecutor(workers, initializer=_init, initargs=(n, name)) as ex:
        res = list(ex.map(_solve_step_big, jobs, chunksize=1))
    x = np.zeros((steps, n, tpl.n_var)); f = np.zeros((steps, n)); ok = np.zeros((steps, n), dtype=bool)
    for k, b, xs, fs, oks, s in res:
        x[k, b], f[k, b], ok[k, b] = xs, fs, oks
    lo, hi = tpl.entry_range(problem.vehicles[0].label, 'splines_seg0', 'var')
    np.savez_compressed(os.path.join(HERE, out), p=np.array(dump['p']), x0=np.array(dump['x0']),
                        lam=np.array(dump['lam']), crossed=np.array(dump['crossed']), x=x, f=f, ok=ok, x_port=np.array(dump['x_port']),
                        spl=np.array([lo, hi]), n_var=tpl.n_var, n_con=tpl.n_con, n_par=tpl.n_par)
    print('%s: %d steps x %d agents, %d converged, crossings at %s, %.0f s'
          % (out, steps, n, int(ok.sum()), np.nonzero(dump['crossed'])[0].tolist(), time.time() - t0))


def _solve_step_big(job):
    from slsqp_reference import solve_slsqp
    tpl, nlp = _STATE['tpl'], _STATE['nlp']
    k, b, x0, p = job
    t0 = time.time()
    x, f, ok = solve_slsqp(nlp, tpl, x0, p, maxiter=1500, accept=(0, 8), viol_tol=1e-7)
    return k, b, x, f, ok, time.time() - t0


# ---- formation: a sequence of ADMM x-updates ----------------------------------------------------------------------
def _init_formation(n):
    import omgtools.backend as be
    from omgtools import scenarios
    from oracle.nlp_numpy import NumpyNLP
    be.create_nlp = lambda tpl, opt, name='': (None, 0.)
    problem, updater, father, lay, P = scenarios.formation_holonomic(n)
    tpl = father.template
    _STATE.update(tpl=tpl, nlp=NumpyNLP(tpl), problem=problem, father=father, lay=lay, P=P, name='formation')


def run_formation(n, updates, workers, out='sol_admm_xupdate.npz'):
    """The x-update NLPs of a formation run (`problems/admm.py:390`; 151 variables / 671 rows for the fleet of
    `examples/formation_holonomic.py`): the receding-horizon protocol of bench.py --workload formation on the host
    (numpy ADMM ops of the tests, oracle port as the x-update solver), the inputs of every x-update after the start-up
    iterations dumped -- parameters incl. the consensus state z, l and rho, the warm-start plan, the multipliers -- and every
    one of these NLPs solved by SLSQP from the warm-start plan."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from omgtools.admm import BatchADMM, FormationMPC
    from admm_numpy_ops import NumpyAdmmOps
    from oracle import port_binding
    t0 = time.time()
    _init_formation(n)
    tpl, problem, father, lay, P = _STATE['tpl'], _STATE['problem'], _STATE['father'], _STATE['lay'], _STATE['P']
    ops = NumpyAdmmOps(tpl, lay, P['p'], P['x0'], tol=1e-6)
    admm = BatchADMM(lay, P['nbr'], ops, rho=1.0)
    moving = []
    for obs in problem.environment.obstacles:
        ox, ov, oa = (tpl.entry_range(obs.label, nm, 'par') for nm in ('x', 'v', 'a'))
        if np.any(P['p'][:, ov[0]:ov[1]] != 0.):
            moving.append((ox[0], ov[0], oa[0], ox[1] - ox[0]))
    mpc = FormationMPC(admm, father, tpl, lay, problem.vehicles[0], obstacles=moving, update_time=0.1, init_iter=5, knot_time=problem.knot_time)
    mpc.initialize()
    dump = dict(p=[], x0=[], lam=[], crossed=[], x_port=[])
    solve = ops.solve

    def tap():
        dump['p'].append(ops.p.copy()); dump['x0'].append(ops.x.copy()); dump['lam'].append(ops.lam.copy())
        st = solve()
        dump['x_port'].append(ops.x.copy())
        return st
    ops.solve = tap
    for k in range(updates):
        status, crossed = mpc.step()
        dump['crossed'].append(bool(crossed))
        assert np.all(np.asarray(status) == 0), k
    jobs = [(k, b, dump['x0'][k][b], dump['p'][k][b]) for k in range(updates) for b in range(n)]
    with ProcessPoolExecutor(workers, initializer=_init_formation, initargs=(n,)) as ex:
        res = list(ex.map(_solve_step_big, jobs, chunksize=1))
    x = np.zeros((updates, n, tpl.n_var)); f = np.zeros((updates, n)); ok = np.zeros((updates, n), dtype=bool)
    for k, b, xs, fs, oks, s in res:
        x[k, b], f[k, b], ok[k, b] = xs, fs, oks
    np.savez_compressed(os.path.join(HERE, out), p=np.array(dump['p']), x0=np.array(dump['x0']), lam=np.array(dump['lam']),
                        crossed=np.array(dump['crossed']), x=x, f=f, ok=ok, x_port=np.array(dump['x_port']),
                        spl=np.array([lay.x_spl, lay.x_spl + lay.ns]), n_var=tpl.n_var, n_con=tpl.n_con, n_par=tpl.n_par)
    print('%s: %d updates x %d agents, %d converged, crossings at %s, %.0f s'
          % (out, updates, n, int(ok.sum()), np.nonzero(dump['crossed'])[0].tolist(), time.time() - t0))


if __name__ == '__main__':
    workers = int(os.environ.get('WORKERS', '8'))
    which = sys.argv[1:] or ['cfg2', 'mpc']
    if 'mpc' in which:
        run_mpc(8, 12, workers)
    if 'cfg2' in which:
        run_cfg2(64, workers)
    if 'cfg3' in which:
        run_cfg2(8, workers, 'quadrotor_p2p', 'sol_cfg3_ms.npz', bends=(1.0, -1.0, 2.5, -2.5), n_random=4)
    if 'cfg5' in which:
        run_cfg2(8, workers, 'holonomic3d_p2p', 'sol_cfg5_ms.npz', bends=(1.0, -1.0, 2.5, -2.5), n_random=20)
    if 'cfg5' in which or 'cfg5_3d' in which:
        extend_cfg5(8, workers)
    if 'mpc3' in which:
        run_mpc_class(4, 6, workers, 'quadrotor_p2p', 'sol_mpc_cfg3.npz')
    if 'admm' in which:
        run_formation(6, 12, workers)
