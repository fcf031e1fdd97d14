"""The committed problem bundles of bench.py's workloads (`omg-tools_amd/omgtools/data/*.npz`, generator
tools/generate_workload_bundles.py): the same template, the same seeded parameters and the same loop tables as the front
end builds -- array for array -- and nothing of the front end is imported on the benchmark path."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FRONT_END = ('omgtools.vehicles', 'omgtools.environment', 'omgtools.problems', 'omgtools.shapes', 'omgtools.execution')


def _front(fn, *a, **kw):
    import omgtools.backend as be
    saved = be.create_nlp
    be.create_nlp = lambda tpl, opt, name='': (None, 0.)
    try:
        return fn(*a, **kw)
    finally:
        be.create_nlp = saved


def _same_template(a, b):
    assert (a.n_var, a.n_par, a.n_con, a.n_atoms, a.n_slots, a.n_terms) == (b.n_var, b.n_par, b.n_con, b.n_atoms, b.n_slots, b.n_terms)
    for k, v in a.flat_arrays().items():
        assert np.array_equal(np.asarray(v), np.asarray(b.flat_arrays()[k])), k
    assert np.array_equal(a.lb, b.lb) and np.array_equal(a.ub, b.ub)
    # (labels carry a per-process object counter -- vehicle0, vehicle3, ...: compared without it)
    import re
    strip = lambda table: [(re.sub(r'(vehicle|obstacle|p2p|environment|admm)\d+', r'\1#', label), re.sub(r'(vehicle|obstacle|p2p|environment|admm)\d+', r'\1#', name), off, r, c)
                           for label, name, off, r, c in table]
    for which in ('var', 'par', 'con'):
        assert strip(a.block_table(which)) == strip(b.block_table(which))


@pytest.mark.parametrize('name,n', [('holonomic_p2p', 16), ('quadrotor_p2p', 4), ('holonomic3d_p2p', 4)])
def test_point_to_point_bundles_equal_the_front_end(name, n):
    from omgtools import scenarios, workloads
    from omgtools.batch import BatchP2P, dual_shift_perm
    pf, Pf = _front(getattr(scenarios, name), n)
    pb, Pb = getattr(workloads, name)(n)
    _same_template(pf.father.template, pb.father.template)
    for k in ('p', 'x0'):
        assert np.array_equal(Pf[k], Pb[k]), k
    assert Pf.get('solver_options', {}) == Pb.get('solver_options', {})
    assert np.array_equal(dual_shift_perm(pf.father), pb.father.dual_perm)
    # the receding-horizon driver derives the same tables from either object (no device: host mode with a dummy solver)
    a = BatchP2P(pf, Pf, ops=object())
    b = BatchP2P(pb, Pb, ops=object())
    for k in ('o_spl', 'p_offs', 'o_t', 'T', 'knot_time', 'n_spl', 'L', 'n_dim', 'obst'):
        assert getattr(a, k) == getattr(b, k), k
    assert np.array_equal(a.perm, b.perm) and np.array_equal(a.shift_entries, b.shift_entries) and np.array_equal(a.shift_mats, b.shift_mats)
    assert np.array_equal(a.basis.knots, b.basis.knots) and a.basis.degree == b.basis.degree


@pytest.mark.parametrize('name', ['formation_holonomic', 'rendezvous_holonomic'])
def test_fleet_bundles_equal_the_front_end(name):
    from omgtools import scenarios, workloads
    from omgtools.consensus import shift_tables
    n = 64
    pf, uf, ff, lf, Pf = _front(getattr(scenarios, name), n)
    pb, ub, fb, lb, Pb = getattr(workloads, name)(n)
    _same_template(ff.template, fb.template)
    for k in ('p', 'x0', 'nbr'):
        assert np.array_equal(Pf[k], Pb[k]), k
    for k, v in lf.__dict__.items():
        if isinstance(v, (int, float)):
            assert getattr(lb, k) == v, k
    spline = name == 'formation_holonomic'
    for ta, tb in zip(shift_tables(ff, ff.template, lf, lf.basis, spline), shift_tables(fb, fb.template, lb, lb.basis, spline)):
        assert np.array_equal(ta[0], tb[0]) and np.array_equal(ta[1], tb[1])
    if hasattr(lf, 'zupdate'):
        assert np.array_equal(lf.zupdate(0.0)[0], lb.zupdate(0.0)[0])


def test_the_benchmark_path_imports_no_front_end_module():
    """bench.py's default workloads, `__graft_entry__.smoke`'s inputs and the drivers they hand them to."""
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import bench\n"
            "from omgtools import workloads, batch, admm, backend, distributed\n"
            "class A: knot_intervals, obstacles, agents = 11, 3, 4\n"
            "bench.p2p_workload(A, 1); workloads.quadrotor_p2p(2); workloads.holonomic3d_p2p(2)\n"
            "workloads.formation_holonomic(512); workloads.rendezvous_holonomic(512)\n"
            "bad = [m for m in %r if m in sys.modules]\n"
            "assert not bad, bad\n" % (os.path.join(ROOT, 'omg-tools_amd'), ROOT, FRONT_END))
    subprocess.check_call([sys.executable, '-c', code])


def test_the_agv_bundle_is_the_fixture_of_the_reference_classes():
    """`omgtools/data/agv_fixedT_k5.npz` (bench.py's `lifted_class` leg): the template and the closed-loop inputs the reference's own
    AGV class produced on `omgx_shim` (tests/golden/agv_fixedT.npz, agv_loop.npz), array for array."""
    from omgtools import workloads
    from omgtools.template import NLPTemplate
    gold = os.path.join(ROOT, 'tests', 'golden')
    ref = NLPTemplate.from_npz(os.path.join(gold, 'agv_fixedT.npz'))
    loop = np.load(os.path.join(gold, 'agv_loop.npz'))
    tpl, P = workloads.agv_loop(26)
    assert (tpl.n_var, tpl.n_con, tpl.n_par, tpl.n_lift) == (ref.n_var, ref.n_con, ref.n_par, ref.n_lift) == (381, 2234, 52, 278)
    for k, v in ref.flat_arrays().items():
        assert np.array_equal(np.asarray(v), np.asarray(tpl.flat_arrays()[k])), k
    assert np.array_equal(tpl.lb, ref.lb) and np.array_equal(tpl.ub, ref.ub)
    assert np.array_equal(P['p'][:13], loop['p']) and np.array_equal(P['p'][13:], loop['p'])
    assert np.array_equal(P['x0'][:13], loop['x0']) and np.array_equal(P['lbg'], loop['lbg']) and np.array_equal(P['iters_host'][:13], loop['iters'])
