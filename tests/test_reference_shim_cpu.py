"""The drop-in itself: the REFERENCE's own classes (imported unchanged from /root/reference) run on this
repository's solve path through `omgx_shim` -- a stand-in `casadi` whose graphs are evaluated once on
polynomial values, giving the NLP template the HIP solver works from (no re-typed front end in between).
CPU tier: the host build of the solver core is injected as the solver; the GPU tier drives the same
`NlpSolver` object from this repository's own front end (the reference is not on the GPU box).

Each case runs in its own process because there `omgtools` must be the reference package."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference/omgtools/__init__.py'


def _run(case, tmp_path):
    env = dict(os.environ, SHIM_DUMP=str(tmp_path / 'dump.npz'))
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'helpers', 'run_reference_on_shim.py'), case],
                       capture_output=True, text=True, timeout=900, env=env)
    lines = [l for l in r.stdout.splitlines() if l.startswith('SHIM_RESULT ')]
    assert lines, r.stdout[-2000:] + r.stderr[-4000:]
    return json.loads(lines[-1][len('SHIM_RESULT '):]), np.load(str(tmp_path / 'dump.npz'))


@pytest.mark.skipif(not os.path.exists(REF), reason='the reference tree is only present in the build container')
def test_reference_p2p_holonomic_runs_on_the_shim(tmp_path):
    """`examples/p2p_holonomic.py` of the reference: its construct code produces the template (SURVEY 8a
    sizes 98 / 325 / 17), the template reproduces the reference's own f and g graphs, the Simulator loop
    brings the vehicle to its target."""
    out, dump = _run('p2p_holonomic', tmp_path)
    assert (out['n_var'], out['n_con'], out['n_par']) == (98, 325, 17)
    assert out['first_status'] == 'Solve_Succeeded'
    assert out['graph_vs_template'] < 1e-12
    assert out['final_error'] < 2e-3 and out['steps'] > 50
    # the same problem from this repository's own front end: same rows, same bounds, same initial guess
    import omgtools.backend as be
    from test_golden_nlp import build
    from oracle.nlp_numpy import NumpyNLP
    saved = be.create_nlp
    be.create_nlp = lambda tpl, opt, name='': (None, 0.)
    try:
        pr = build('cfg1_p2p_holonomic')
        pr.reinitialize()
    finally:
        be.create_nlp = saved
    tpl = pr.father.template
    assert np.array_equal(tpl.lb, dump['lb']) and np.array_equal(tpl.ub, dump['ub'])
    assert np.allclose(np.asarray(pr.father.get_variables()).reshape(-1), dump['x0'], atol=1e-14)
    assert np.array_equal(tpl.row_ptr, dump['row_ptr'])          # the same number of polynomial terms in every row
    nn = NumpyNLP(tpl)
    for xv, pv, fs, gs in zip(dump['xs'], dump['ps'], dump['fs'], dump['gs']):
        f, g = nn.fg(xv, nn.term_coefs(pv))
        assert abs(f - fs) < 1e-12 * (1 + abs(fs)) and np.abs(g - gs).max() < 1e-12 * (1 + np.abs(gs).max())


@pytest.mark.skipif(not os.path.exists(REF), reason='the reference tree is only present in the build container')
def test_reference_rectangles_runs_on_the_shim(tmp_path):
    out, _ = _run('p2p_holonomic_rect', tmp_path)
    assert out['first_status'] == 'Solve_Succeeded' and out['graph_vs_template'] < 1e-12
    assert out['final_error'] < 2e-3


@pytest.mark.skipif(not os.path.exists(REF), reason='the reference tree is only present in the build container')
def test_reference_dubins_class_builds_and_solves_on_the_shim(tmp_path, monkeypatch):
    """a vehicle model this package's front end does not have -- the reference's own `Dubins`
    (`vehicles/dubins.py:47`, tangent-half-angle substitution; hyperplane rows of degree 4 in the variables) -- imported
    unchanged, built on the shim and solved; the committed fixture tests/golden/dubins_fixedT.npz (what the GPU tier
    solves) is this template.  (The full Simulator run of tests/golden/generate_shim_fixtures.py takes two minutes on the host
    build: 119 updates, target pose reached to 7e-3.)"""
    monkeypatch.setenv('SHIM_NO_SIM', '1')
    monkeypatch.setenv('DUBINS_SUBST', '0')
    out, dump = _run('p2p_dubins', tmp_path)
    assert (out['n_var'], out['n_con'], out['n_par'], out['n_terms']) == (105, 456, 19, 46912)
    assert out['first_status'] == 'Solve_Succeeded' and out['graph_vs_template'] < 1e-12
    gold = np.load(os.path.join(ROOT, 'tests', 'golden', 'dubins_fixedT.npz'))
    assert np.array_equal(gold['row_ptr'], dump['row_ptr']) and np.array_equal(gold['lb'], dump['lb'])
    assert np.allclose(gold['x0'], dump['x0'], atol=1e-14) and np.allclose(gold['p0'], dump['p0'], atol=1e-14)


@pytest.mark.skipif(not os.path.exists(REF), reason='the reference tree is only present in the build container')
def test_reference_revolving_door_runs_on_the_shim(tmp_path):
    """SURVEY 8(f)3, rotating obstacles: `examples/revolving_door.py` of the reference (two beams turning at 0.94 rad/s,
    `environment/obstacle.py:299-332`) on the shim: COS / SIN atoms, the Simulator run ends at the target."""
    out, dump = _run('revolving_door', tmp_path)
    assert (out['n_var'], out['n_con'], out['n_par']) == (184, 862, 58)
    assert out['first_status'] == 'Solve_Succeeded' and out['graph_vs_template'] < 1e-12
    assert out['final_error'] < 2e-3 and out['steps'] > 50
    gold = np.load(os.path.join(ROOT, 'tests', 'golden', 'revolving_door.npz'))
    assert np.array_equal(gold['row_ptr'], dump['row_ptr']) and np.array_equal(gold['lb'], dump['lb'])
