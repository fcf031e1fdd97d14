#!/usr/bin/env python
"""Fixtures of problem classes only the reference's front end has, as flat templates.

tests/golden/revolving_door.npz: `examples/revolving_door.py` (two rotating beams, `environment/obstacle.py:299-332`: the
orientation enters through cos / sin of a parameter expression -- COS / SIN atoms of the template's parameter program).

tests/golden/dubins_fixedT.npz: the NLP of the reference's OWN Dubins class (`vehicles/dubins.py:47`, tangent-half-angle
model; `examples/p2p_dubins.py` with a fixed horizon and without the substituted velocity splines, so that the hyperplane
rows a . integral(v_til (1 - tg_ha^2)) are of degree 4 in the variables) as a flat template, produced by executing the
reference's modules on `omgx_shim` (tests/helpers/run_reference_on_shim.py, this container only), together with

  p0, x0         parameters and initial guess of the first solve (`Problem.reinitialize`)
  xs, ps, fs, gs values of the reference's own f / g graphs at three random points (the shim evaluates the closures the
                 reference built; graph vs template agree to 1e-15 in the helper)
  x_slsqp, f_slsqp   scipy SLSQP on the restated NLP from x0 (the independent solver of tests/slsqp_reference.py)

tests/golden/dubins_subst.npz (round 4): the same class with `options['substitution']` (`vehicles/dubins.py:92-115`): the position is
the integral of separate velocity splines, tied to the tangent-half-angle expressions by TWO-SIDED rows  -1e-3 <= x - int(v_til (1 -
tg_ha^2)) <= 1e-3  (`basics/optilayer.py:634-666`) -- 118 range rows; what the library could not take before ABI 5.

tests/golden/bicycle_fixedT.npz, agv_fixedT.npz, dubins_freeT.npz, trailer_freeT.npz (round 5): the classes whose rows are products of
more than four variable factors or quotients by a variable -- `vehicles/bicycle.py:53`, `vehicles/agv.py:50`, `vehicles/trailer.py:28`
(the bodies of `examples/p2p_bicycle.py`, `p2p_agv.py`, `p2p_trailer.py`) and `examples/p2p_dubins.py` as shipped (free end time: the
velocity divides by T).  `omgx_shim` writes such an expression with auxiliary variables (omgtools/symbolic.py `LIFT_CAP`, template.py
`_append_lifted`): the template has the caller's variables and rows first and the auxiliaries / their defining rows behind them.
xs carries the caller's variables extended by the auxiliaries, gs the REFERENCE's g at the caller's variables (defining rows: 0).
x_slsqp: SLSQP on the caller's own problem (oracle/slsqp_numpy.py `solve_slsqp_reduced`) where it converges.

Run in the build container:  python tests/golden/generate_shim_fixtures.py [dubins | dubins_subst | revolving_door | bicycle | agv | agv_loop | dubins_freeT | dubins_shipped | trailer]"""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, 'omg-tools_amd'))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def make(case, out_name, subst='0', freeT='0', slsqp=True):
    tmp = os.path.join(HERE, '_shim_tmp.npz')
    lifted = out_name.split('_')[0] in ('bicycle', 'agv', 'trailer') or freeT == '1'
    env = dict(os.environ, SHIM_TEMPLATE=tmp, DUBINS_SUBST=subst, DUBINS_FREET=freeT, FREET=freeT, KNOTS='5',
               SHIM_NO_SIM='1' if (subst == '1' or lifted) else '0', SHIM_DUMP=os.path.join(HERE, '_shim_dump.npz'))
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'helpers', 'run_reference_on_shim.py'), case],
                       capture_output=True, text=True, env=env)
    print(r.stdout[-600:])
    assert 'SHIM_RESULT' in r.stdout, r.stderr[-3000:]
    from omgtools.template import NLPTemplate
    from oracle.nlp_numpy import NumpyNLP
    from slsqp_reference import solve_slsqp
    tpl = NLPTemplate.from_npz(tmp)
    d = dict(np.load(tmp))
    nlp = NumpyNLP(tpl)
    if getattr(tpl, 'n_lift', 0):
        from oracle.slsqp_numpy import solve_slsqp_reduced
        xs, fs, ok = (solve_slsqp_reduced(nlp, tpl, d['x0'], d['p0'], accept=(0, 8), viol_tol=1e-7) if slsqp
                      else (np.zeros(tpl.n_var), float('nan'), False))
    else:
        xs, fs, ok = solve_slsqp(nlp, tpl, d['x0'], d['p0'])
    print('SLSQP f', fs, 'converged', ok)        # (slsqp_ok = 0: SLSQP gave up; the tests then rely on the KKT conditions)
    np.savez_compressed(os.path.join(HERE, out_name), x_slsqp=xs, f_slsqp=fs, slsqp_ok=int(ok), **d)
    os.remove(tmp)
    os.remove(env['SHIM_DUMP'])


def make_loop(case, out_name, updates):
    """The first `updates` updates of the reference's Simulator on `case` (execution/simulator.py:39-52, its own classes on the shim,
    the host build of the solver behind them): parameters, initial guess (the reference's warm start: the shifted previous plan),
    bounds and the result of every solve -- the device re-solves them as one batch (tests/test_lifted.py)."""
    import glob
    import tempfile
    tmp = tempfile.mkdtemp()
    env = dict(os.environ, PORT_SOLVER_DUMP=tmp, SHIM_MAX_UPDATES=str(updates), FREET='0', KNOTS='5')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'helpers', 'run_reference_on_shim.py'), case],
                       capture_output=True, text=True, env=env)
    print(r.stdout[-400:])
    assert 'SHIM_RESULT' in r.stdout, r.stderr[-3000:]
    files = sorted(glob.glob(os.path.join(tmp, 'solve_*.npz')))
    ds = [np.load(f) for f in files]
    assert all(np.array_equal(d['lbg'], ds[0]['lbg']) and np.array_equal(d['ubg'], ds[0]['ubg']) for d in ds)
    np.savez_compressed(os.path.join(HERE, out_name), p=np.array([d['p'] for d in ds]), x0=np.array([d['x0'] for d in ds]),
                        x=np.array([d['x'] for d in ds]), lam_g=np.array([d['lam_g'] for d in ds]), lbg=ds[0]['lbg'], ubg=ds[0]['ubg'],
                        status=np.array([int(d['status']) for d in ds]), iters=np.array([int(d['iters']) for d in ds]))
    for f in files:
        os.remove(f)
    os.rmdir(tmp)


if __name__ == '__main__':
    which = sys.argv[1:] or ['dubins', 'revolving_door']
    if 'dubins' in which:
        make('p2p_dubins', 'dubins_fixedT.npz')
    if 'dubins_subst' in which:
        make('p2p_dubins', 'dubins_subst.npz', subst='1')
    if 'revolving_door' in which:
        make('revolving_door', 'revolving_door.npz')
    if 'bicycle' in which:
        make('p2p_bicycle', 'bicycle_fixedT.npz', slsqp=False)      # (SLSQP runs a denominator through zero on this class)
    if 'agv' in which:
        make('p2p_agv', 'agv_fixedT.npz')
    if 'agv_loop' in which:
        make_loop('p2p_agv', 'agv_loop.npz', 12)
    if 'dubins_shipped' in which:
        # `examples/p2p_dubins.py` exactly as shipped: substituted velocity splines AND a free end time (the example asks IPOPT for a
        # limited-memory Hessian: the solves of the tests set `hess_approx`)
        make('p2p_dubins', 'dubins_shipped.npz', subst='1', freeT='1', slsqp=False)
    if 'dubins_freeT' in which:
        make('p2p_dubins', 'dubins_freeT.npz', freeT='1', slsqp=False)
    if 'trailer' in which:
        make('p2p_trailer', 'trailer_freeT.npz', freeT='1', slsqp=False)
