"""C-ABI surface and host-side logic (no compute on a device)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from omgtools.backend import LIB_PATH
    header = open(os.path.join(ROOT, 'include', 'omgx.h')).read()
    declared = set(re.findall(r'\b(omgx_[a-z_]+)\s*\(', header))
    assert {'omgx_batch_create', 'omgx_batch_solve', 'omgx_batch_sample', 'omgx_batch_shift',
            'omgx_batch_destroy'} <= declared
    lib = ctypes.CDLL(LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    lib.omgx_version.restype = ctypes.c_int
    assert lib.omgx_version() == 9
    lib.omgx_status_string.restype = ctypes.c_char_p
    assert lib.omgx_status_string(0) == b'Solve_Succeeded'


def test_no_device_is_a_loud_error(cfg2_small):
    """Without a GPU the product path must fail, not fall back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from omgtools.backend import BatchSolver, OmgxError
    problem, P = cfg2_small
    with pytest.raises(OmgxError):
        BatchSolver(problem.father.template, 4)


def test_dimensions_table():
    """SURVEY.md §8 problem sizes (K7)."""
    import omgtools.backend as be
    from omgtools.scenarios import holonomic_p2p
    saved = be.create_nlp
    be.create_nlp = lambda tpl, opt, name='': (None, 0.)
    try:
        for sd, dims in ((0., (164, 528, 35)), (0.1, (206, 612, 35))):
            problem, _ = holonomic_p2p(1, safety_distance=sd)
            t = problem.father.template
            assert (t.n_var, t.n_con, t.n_par) == dims
    finally:
        be.create_nlp = saved


def test_solver_plan_is_block_arrow(cfg2_small):
    """The plan is derived inside the library from the flat NLP (`omgx_plan_describe`, host only)."""
    from omgtools.backend import describe_plan
    problem, _ = cfg2_small
    tpl = problem.father.template
    plan = describe_plan(tpl)
    # three hyperplane leaves + the gathered terminal slacks g0, g1 (each coefficient a singleton)
    assert plan['n_leaf'] == 4 and plan['leaf_sizes'] == [36, 36, 36, 28]
    assert plan['n_root'] == 28 + 1 and plan['n_eq'] == 10
    order = plan['order']
    assert sorted(order.tolist()) == list(range(tpl.n_var + 1)) and order[-1] == tpl.n_var
    # hyperplane blocks of degree-1 splines are block tridiagonal by knot: (a0, a1, b) of neighbouring knots
    assert plan['leaf_bw'][:3] == [5, 5, 5] and plan['leaf_bw'][3] == 0
    assert plan['leaf_cpl'] == [29, 29, 29, 29]
    # register-resident wave path on the compact store: with the Jacobian values in a slab the rest fits half a CU (mode 5: row values in LDS; 4: those in the slab too)
    assert plan['wave_path'] == 1 and plan['ws_mode'] == 5 and plan['lds_bytes'] <= 80 * 1024


def test_workspace_modes_of_the_benchmark_classes():
    """Which workspace placement the library picks for BASELINE.json's classes (host only): config 2 on the
    register-resident path with the compact store (two agents per CU: mode 4 = Jacobian values and hv in a slab), config 3 with the KKT store in an HBM slab, config 5 with the row arrays there as well;
    the LDS part always fits one CU; the formation template (monomials with up to seven atoms) is accepted."""
    import omgtools.backend as be
    from omgtools import scenarios
    saved = be.create_nlp
    be.create_nlp = lambda tpl, opt, name='': (None, 0.)
    try:
        want = {'holonomic_p2p': (5, 1), 'quadrotor_p2p': (1, 0), 'holonomic3d_p2p': (3, 0)}
        for name, (mode, wave) in want.items():
            problem, _ = getattr(scenarios, name)(2)
            plan = be.describe_plan(problem.father.template)
            if mode in (0, 4, 5):
                assert plan['wave_path'] == wave, name
            assert plan['ws_mode'] == mode and 0 < plan['lds_bytes'] <= 160 * 1024, (name, plan['ws_mode'], plan['lds_bytes'])
            assert all(bw <= 8 for bw in plan['leaf_bw']), name              # banded leaves (Cuthill-McKee order)
        father = scenarios.formation_holonomic(4)[2]
        plan = be.describe_plan(father.template)
        assert plan['ws_mode'] in (0, 1, 4, 5) and plan['lds_bytes'] <= 160 * 1024
    finally:
        be.create_nlp = saved


def test_plan_without_root_hint_finds_a_separator(cfg2_small):
    """No structure hint from the caller: the library picks the root itself (highest-degree variables
    first) and still gets leaves that fit one wave."""
    import ctypes as C
    import omgtools.backend as be
    problem, _ = cfg2_small
    tpl = problem.father.template
    lib = be.load_library()
    ct, keep = be.make_ctemplate(tpl)
    ct.n_root_vars = 0
    info = be.CPlanInfo()
    lib.omgx_plan_describe.argtypes = [C.POINTER(be.CTemplate), C.POINTER(be.CPlanInfo), C.c_void_p]
    assert lib.omgx_plan_describe(C.byref(ct), C.byref(info), None) == 0
    assert info.n_leaf >= 3 and max(info.leaf_size[:info.n_leaf]) <= 64


def test_obstacle_position_spline_closed_form():
    """K8: the obstacle's quadratic position spline equals x + v t + a t^2/2."""
    import omgtools.backend as be
    from omgtools import Holonomic, Environment, Obstacle, Circle, Square, Point2point
    from omgtools.splines import BSpline
    saved = be.create_nlp
    be.create_nlp = lambda tpl, opt, name='': (None, 0.)
    try:
        veh = Holonomic()
        veh.set_initial_conditions([0., 0.]); veh.set_terminal_conditions([1., 1.])
        env = Environment(room={'shape': Square(5.)})
        obs = Obstacle({'position': [0.3, -0.2], 'velocity': [0.1, 0.05], 'acceleration': [0.02, -0.01]},
                       shape=Circle(0.2))
        env.add_obstacle(obs)
        prob = Point2point(veh, env, options={'verbose': 0})
        prob.init()
        tpl = prob.father.template
        p = prob.father.set_parameters(0.).cat.copy()
        t_now, T = 0.37, 10.
        lo, hi = tpl.entry_range(prob.label, 't', 'par'); p[lo:hi] = t_now
        atoms = tpl.eval_atoms_host(p)
        x0, v0, a0 = np.array([0.3, -0.2]), np.array([0.1, 0.05]), np.array([0.02, -0.01])
        for k in range(2):
            cf = [tpl.eval_poly_host(c, np.zeros(tpl.n_var), atoms) for c in obs.pos_spline[k].coeffs]
            for tau in (0., 0.5, 1.):
                dt_ = tau * T - t_now
                want = x0[k] + v0[k] * dt_ + 0.5 * a0[k] * dt_**2
                assert abs(BSpline(obs.basis, cf)(tau) - want) < 1e-12
    finally:
        be.create_nlp = saved


def test_bad_arguments_return_error_codes(cfg2_small):
    """Error behaviour of the C ABI: negative codes + `omgx_last_error`, nothing throws; argument
    checks come before any device work, so they are testable without a GPU."""
    import ctypes as C
    import omgtools.backend as be
    problem, _ = cfg2_small
    lib = be.load_library()
    ct, keep = be.make_ctemplate(problem.father.template)
    h = C.c_void_p()
    assert lib.omgx_batch_create(C.byref(ct), 0, 0, C.byref(h)) == -1            # OMGX_E_INVALID: empty batch
    assert b'bad argument' in lib.omgx_last_error()
    assert lib.omgx_batch_create(None, 4, 0, C.byref(h)) == -1
    assert lib.omgx_batch_create(C.byref(ct), 4, 0, None) == -1
    lib.omgx_batch_solve.restype = C.c_int
    assert lib.omgx_batch_solve(None, None, None, None, None, None, None, None, None, 0) == -1
    assert lib.omgx_batch_set_options(None, None) == -1
    assert lib.omgx_batch_sync(None) == -1
    lib.omgx_batch_destroy(None)                                                  # no-op
    for code, name in ((0, b'Solve_Succeeded'), (1, b'Maximum_Iterations_Exceeded'),
                       (2, b'Infeasible_Problem_Detected'), (3, b'Unsupported_Bounds'),
                       (4, b'Numerical_Failure'), (99, b'Unknown')):
        assert lib.omgx_status_string(code) == name
    opt = be.COptions()
    lib.omgx_default_options(C.byref(opt))
    assert (opt.tol, opt.max_iter, opt.warm_start) == (1e-3, 300, 0) and opt.dw_leaf_ratio_cold == 1.0
    # an inconsistent template is rejected, not executed
    ct2, keep2 = be.make_ctemplate(problem.father.template)
    ct2.n_var = -3
    assert lib.omgx_batch_create(C.byref(ct2), 4, 0, C.byref(h)) == -1
    info = be.CPlanInfo()
    lib.omgx_plan_describe.argtypes = [C.POINTER(be.CTemplate), C.POINTER(be.CPlanInfo), C.c_void_p]
    ct3, keep3 = be.make_ctemplate(problem.father.template)
    bad_eq = np.array([10 ** 6], dtype=np.int32)
    ct3.n_eq, ct3.eq_rows = 1, bad_eq.ctypes.data_as(C.POINTER(C.c_int32))
    assert lib.omgx_plan_describe(C.byref(ct3), C.byref(info), None) == -1
    assert b'equality row' in lib.omgx_last_error()


def test_product_package_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under omg-tools_amd/ may import or include it (host
    solvers for the CPU tier are injected by the tests)."""
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'omg-tools_amd')
    py = re.compile(r'^\s*(from|import)\s+oracle\b', re.M)
    inc = re.compile(r'#include\s+"[^"]*oracle/')
    hits = []
    for d, _, files in os.walk(root):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                txt = open(os.path.join(d, f), errors='ignore').read()
                if py.search(txt) or inc.search(txt):
                    hits.append(os.path.join(d, f))
    assert not hits, hits


def test_library_plan_agrees_with_the_oracle_partition(cfg2_small):
    """Leaf membership derived inside the library == the oracle's independent Python statement of the
    partition (oracle/plan_numpy.py); only the order inside a leaf differs (reverse Cuthill-McKee)."""
    from omgtools.backend import describe_plan
    from oracle.plan_numpy import SolverPlan
    problem, _ = cfg2_small
    tpl = problem.father.template
    lib = describe_plan(tpl)
    ref = SolverPlan(tpl)
    off = np.cumsum([0] + lib['leaf_sizes'])
    got = [sorted(lib['order'][off[l]:off[l + 1]].tolist()) for l in range(lib['n_leaf'])]
    assert got == [sorted(l) for l in ref.leaves]
    assert sorted(lib['order'][off[-1]:].tolist()) == sorted(ref.order[ref.root_off:].tolist())


def test_admm_table_keys_cover_the_full_period():
    """The z-update tables of the C++ ADMM classes (`omgtools.backend.save_admm_tables`): one key per time since the
    last knot an update can happen at, over the whole period -- also when update_time does not divide knot_time (the
    reference's generated updz takes the time as a continuous input, `export/export_admm.py`)."""
    import omgtools.backend as be
    assert be.admm_table_keys(1.0, 0.1) == [round(0.1 * k, 6) for k in range(10)]
    keys = be.admm_table_keys(1.0, 0.3)                       # 0.3 does not divide 1.0: the period is 10 updates long
    assert keys == [round(0.1 * k, 6) for k in range(10)]
    keys = be.admm_table_keys(10.0 / 11.0, 0.1)               # horizon 10 s, 11 knot intervals: 100 updates per period
    assert len(keys) == 100 and keys[0] == 0.0 and all(0.0 <= t < 10.0 / 11.0 for t in keys)
    for k in (1, 37, 99, 101, 250):                           # every update time is in the table, with the C++ side's rounding
        t_rel = round((k * 0.1) % (10.0 / 11.0), 6)
        if 10.0 / 11.0 - t_rel < 5e-7:
            t_rel = 0.0
        assert min(abs(t_rel - q) for q in keys) < 1.5e-6
    with pytest.raises(ValueError):
        be.admm_table_keys(1.0, 0.1 * 2 ** 0.5)


def test_sub_batch_sizes_follow_the_resident_workgroups():
    """`omgtools.batch.split_bounds` (the per-step product path): contiguous, complete, sizes in quarters of the resident workgroups
    with the larger sub-batches first; the even split when the number of slots is unknown or the batch is small."""
    from omgtools.batch import split_bounds
    sizes = lambda b: [hi - lo for lo, hi in b]
    assert sizes(split_bounds(1024, 3, 512)) == [384, 384, 256]
    assert sizes(split_bounds(1000, 3, 512)) == [384, 384, 232]
    assert sizes(split_bounds(4096, 3, 512)) == [1408, 1408, 1280]
    assert sizes(split_bounds(1024, 2, 512)) == [512, 512]
    assert sizes(split_bounds(1024, 3)) == [342, 341, 341] and sizes(split_bounds(300, 3, 512)) == [100, 100, 100]
    for B, n, slots in ((1024, 3, 512), (1000, 3, 512), (777, 4, 256), (5, 2, 512), (1500, 3, 512)):
        b = split_bounds(B, n, slots)
        assert b[0][0] == 0 and b[-1][1] == B and all(b[i][1] == b[i + 1][0] for i in range(n - 1)) and min(sizes(b)) >= 1
