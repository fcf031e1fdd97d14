"""Generates tests/golden/sol_*.npz: solutions of seeded agents of BASELINE.json's configurations from
solvers that share NO code with the product kernel (csrc/omgx_core.h) -- the parity fixtures of
tests/test_golden_solutions.py (CPU tier: host build of the kernel; GPU tier: the HIP path) and of
`__graft_entry__.smoke()`.

  sol_cfg2.npz   64 agents of config 2 (Holonomic, K = 11, 3 circles) by scipy SLSQP (dense SQP: its own
                 QP solver, its own line search; nothing interior-point about it) on the restated NLP
                 (oracle/nlp_numpy.py, which tests/golden pins to the reference's construct code), from the
                 reference's initial guess (`get_init_spline_value`, hyperplanes zero) where that lands in
                 the local minimum an interior point reaches, otherwise from inside that basin; the field
                 `method` says which (see `_solve_slsqp`).
  sol_cfg3.npz   8 agents of the Quadrotor class (K = 13, 5 moving circles),
  sol_cfg5.npz   8 agents of the Holonomic3D class (K = 15, 10 spheres): oracle/ipm_numpy.py at tol 1e-6
                 (SLSQP needs hours at these sizes), each solution then handed to SLSQP as a starting
                 point for a bounded number of iterations: the objective must not improve (field `f_polish`).

CasADi/IPOPT outputs are unobtainable here; these are the independent stand-ins.
Run from the repository root:  python tests/golden/generate_solutions.py  (about 20 minutes on 8 cores)."""
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

_STATE = {}


def _scenario(name, n):
    import omgtools.backend as be
    from omgtools import scenarios
    be.create_nlp = lambda tpl, opt, name='': (None, 0.)
    return getattr(scenarios, name)(n)


def _init(name, n):
    from oracle.nlp_numpy import NumpyNLP
    problem, P = _scenario(name, n)
    tpl = problem.father.template
    _STATE.update(tpl=tpl, P=P, nlp=NumpyNLP(tpl), problem=problem)


def _solve_slsqp(b):
    """Config 2.  From the degenerate initial guess (hyperplanes zero: the first step decides on which side
    of every obstacle the plan passes) SLSQP and an interior point end in the same local minimum for about
    half of the agents.  method 0: SLSQP from the initial guess, and the dense numpy interior point
    (oracle/ipm_numpy.py) from the same guess agrees with it (objective to 1e-6).  method 2: they chose
    different sides; stored is what SLSQP converges to when started inside the interior point's basin
    (at its solution): an independent confirmation and refinement of that local minimum.  method 1: SLSQP
    failed there too, stored is the interior point's own solution."""
    from slsqp_reference import solve_slsqp
    from oracle import ipm_numpy
    tpl, P, nlp = _STATE['tpl'], _STATE['P'], _STATE['nlp']
    t0 = time.time()
    xa, fa, oka = solve_slsqp(nlp, tpl, P['x0'][b], P['p'][b], maxiter=600)
    # (tol 1e-6: at 1e-8 the dense unpivoted LDL' of the numpy statement ends in rounding noise for half of the agents)
    r = ipm_numpy.solve(nlp, P['x0'][b], P['p'][b], tpl.lb, tpl.ub, opts={'tol': 1e-6, 'max_iter': 400})
    okb, fb = r['status'] == 0, float(r['f'])
    if oka and okb and abs(fa - fb) < 1e-5 * (1 + abs(fa)):
        return b, xa, fa, True, 0, time.time() - t0
    if okb:
        xc, fc, okc = solve_slsqp(nlp, tpl, r['x'], P['p'][b], maxiter=600)
        if okc and abs(fc - fb) < 1e-5 * (1 + abs(fb)):
            return b, xc, fc, True, 2, time.time() - t0
        return b, r['x'], fb, True, 1, time.time() - t0
    return b, xa, fa, bool(oka), 3, time.time() - t0          # method 3: SLSQP only (the interior point did not converge)


def _solve_ipm(b):
    from slsqp_reference import solve_slsqp
    from oracle import ipm_numpy
    tpl, P, nlp = _STATE['tpl'], _STATE['P'], _STATE['nlp']
    t0 = time.time()
    opts = dict(P.get('solver_options', {}), tol=float(os.environ.get('ORACLE_TOL', '1e-6')), max_iter=500)
    r = ipm_numpy.solve(nlp, P['x0'][b], P['p'][b], tpl.lb, tpl.ub, opts=opts)
    f_polish = np.nan
    if r['status'] == 0:
        xs, fs, oks = solve_slsqp(nlp, tpl, r['x'], P['p'][b], maxiter=int(os.environ.get('POLISH_ITERS', '8')))
        f_polish = fs
    return b, r['x'], float(r['f']), r['status'] == 0, 1, time.time() - t0, f_polish


def run(name, n, fn, out, workers):
    t0 = time.time()
    with ProcessPoolExecutor(workers, initializer=_init, initargs=(name, n)) as ex:
        res = list(ex.map(fn, range(n)))
    _init(name, n)
    tpl, P, problem = _STATE['tpl'], _STATE['P'], _STATE['problem']
    lo, hi = tpl.entry_range(problem.vehicles[0].label, 'splines_seg0', 'var')
    x = np.stack([r[1] for r in res])
    data = dict(p=P['p'][:n], x0=P['x0'][:n], x=x, f=np.array([r[2] for r in res]),
                ok=np.array([r[3] for r in res]), method=np.array([r[4] for r in res], dtype=np.int8),
                spl=np.array([lo, hi]), n_var=tpl.n_var, n_con=tpl.n_con, n_par=tpl.n_par,
                seconds=np.array([r[5] for r in res]))
    if len(res[0]) > 6:
        data['f_polish'] = np.array([r[6] for r in res])
    np.savez_compressed(os.path.join(HERE, out), **data)
    print('%s: %d agents, %d converged, methods %s, %.0f s' % (out, n, int(data['ok'].sum()), np.bincount(data['method']), time.time() - t0))


if __name__ == '__main__':
    workers = int(os.environ.get('WORKERS', '8'))
    which = sys.argv[1:] or ['cfg2', 'cfg3', 'cfg5']
    if 'cfg2' in which:
        run('holonomic_p2p', 64, _solve_slsqp, 'sol_cfg2.npz', workers)
    if 'cfg3' in which:
        run('quadrotor_p2p', 8, _solve_ipm, 'sol_cfg3.npz', workers)
    if 'cfg5' in which:
        run('holonomic3d_p2p', 8, _solve_ipm, 'sol_cfg5.npz', workers)
