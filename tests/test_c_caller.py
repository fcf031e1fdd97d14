"""The C ABI from C: tests/c/p2p_solve.c (plain C, compiled with gcc, linked against libomgx.so) loads a template
file written by the Python front end and calls `omgx_batch_solve` the way the reference's C++ export calls its
nlpsol (`export/point2point/Point2Point.cpp:80-91, 207-231`).  CPU tier: compile, link, read the template,
derive the plan.  GPU tier: the solve, bit-identical to the ctypes path."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'omg-tools_amd', 'csrc')


@pytest.fixture(scope='module')
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp('c') / 'p2p_solve')
    subprocess.check_call(['gcc', '-std=c99', '-Wall', '-Werror', '-O1', os.path.join(ROOT, 'tests', 'c', 'p2p_solve.c'),
                           '-I', os.path.join(ROOT, 'include'), '-L', CSRC, '-lomgx', '-Wl,-rpath,' + CSRC, '-o', out])
    return out


@pytest.fixture(scope='module')
def case(tmp_path_factory):
    import omgtools.backend as be
    from omgtools.scenarios import holonomic_p2p
    saved = be.create_nlp
    be.create_nlp = lambda tpl, opt, name='': (None, 0.)
    try:
        problem, P = holonomic_p2p(4)
    finally:
        be.create_nlp = saved
    tpl = problem.father.template
    d = tmp_path_factory.mktemp('tpl')
    path = be.save_template(tpl, str(d / 'cfg2.omgx'))
    return tpl, P, path, d


def test_c_program_links_and_reads_the_template(exe, case):
    import omgtools.backend as be
    tpl, P, path, d = case
    out = subprocess.check_output([exe, path, 'plan']).decode().split()
    got = dict(zip(out[0::2], out[1::2]))
    plan = be.describe_plan(tpl)
    assert int(got['version']) == 9
    assert (int(got['n_var']), int(got['n_con']), int(got['n_par'])) == (tpl.n_var, tpl.n_con, tpl.n_par)
    assert (int(got['n_leaf']), int(got['n_root']), int(got['nnz_j'])) == (plan['n_leaf'], plan['n_root'], plan['nnz_j'])
    assert int(got['lds_bytes']) == plan['lds_bytes']


def test_block_table_travels_through_the_abi_and_fills_p_by_name(exe, case):
    """the table (name, offset, rows, cols) of x / p / g -- what the reference's exporter hard-codes into the
    generated C++ (`export/export.py:302-353`) -- is part of the template and of the template file; the C program lists it
    and fills p (conditions, obstacles) and the initial guess by entry name like `Point2Point::fillParameterDict`
    (`Point2Point.cpp:263-294`): the same arrays as the Python scenario builder's."""
    tpl, P, path, d = case
    lines = subprocess.check_output([exe, path, 'blocks']).decode().strip().split('\n')
    got = [(k, n, int(o), int(r), int(c)) for k, n, o, r, c in (ln.split() for ln in lines)]
    want = [(which, '%s.%s' % (label, name), off, r, c) for which in ('var', 'par', 'con')
            for label, name, off, r, c in tpl.block_table(which)]
    assert got == want and len(got) > 20
    # the documented order of x (`problem.py:45-48`): vehicle splines first, then the problem's g0, g1, then a / b per obstacle
    xs = [n.split('.')[1] for k, n, o, r, c in got if k == 'var']
    assert xs[0] == 'splines_seg0' and xs[1:3] == ['g0', 'g1'] and xs[3].startswith('a_') and xs[4].startswith('b_')
    B = P['p'].shape[0]
    veh_state = tpl.entry_range([n for k, n, *_ in got if n.endswith('.state0')][0].split('.')[0], 'state0', 'par')[0]
    cond = str(d / 'cond.bin')
    obs = [(n.split('.')[0]) for k, n, *_ in got if k == 'par' and n.endswith('.rad')]
    with open(cond, 'wb') as fp:
        fp.write(np.array([B, len(obs)], dtype=np.int32).tobytes())
        for b in range(B):
            pose = tpl.entry_range([n for k, n, *_ in got if n.endswith('.poseT')][0].split('.')[0], 'poseT', 'par')[0]
            o_T = [o for k, n, o, r, c in got if k == 'par' and n.endswith('.T')][0]
            fp.write(np.r_[P['p'][b, veh_state:veh_state + 2], P['p'][b, pose:pose + 2], P['p'][b, o_T]].tobytes())
            for lab in obs:
                ox, orad = tpl.entry_range(lab, 'x', 'par')[0], tpl.entry_range(lab, 'rad', 'par')[0]
                fp.write(np.r_[P['p'][b, ox:ox + 2], P['p'][b, orad]].tobytes())
    out = str(d / 'filled.bin')
    subprocess.check_call([exe, path, 'fill', cond, out])
    raw = np.frombuffer(open(out, 'rb').read())
    p_c, x_c = raw[:B * tpl.n_par].reshape(B, tpl.n_par), raw[B * tpl.n_par:].reshape(B, tpl.n_var)
    assert np.array_equal(p_c, P['p'])
    assert np.abs(x_c - P['x0']).max() < 1e-14


def test_template_file_round_trip_and_rejects_garbage(case, tmp_path):
    import ctypes as C
    import omgtools.backend as be
    tpl, P, path, d = case
    lib = be.load_library()
    lib.omgx_template_read.argtypes = [C.c_char_p, C.POINTER(C.POINTER(be.CTemplate))]
    lib.omgx_template_free.argtypes = [C.POINTER(be.CTemplate)]
    lib.omgx_template_write.argtypes = [C.POINTER(be.CTemplate), C.c_char_p]
    ptr = C.POINTER(be.CTemplate)()
    assert lib.omgx_template_read(os.fsencode(path), C.byref(ptr)) == 0
    again = str(tmp_path / 'again.omgx')
    assert lib.omgx_template_write(ptr, os.fsencode(again)) == 0
    lib.omgx_template_free(ptr)
    assert open(path, 'rb').read() == open(again, 'rb').read()
    bad = str(tmp_path / 'bad.omgx')
    open(bad, 'wb').write(open(path, 'rb').read()[:1000])
    ptr = C.POINTER(be.CTemplate)()
    assert lib.omgx_template_read(os.fsencode(bad), C.byref(ptr)) < 0 and not ptr
    assert b'truncated' in lib.omgx_last_error()
    open(bad, 'wb').write(b'not a template')
    assert lib.omgx_template_read(os.fsencode(bad), C.byref(ptr)) < 0


@pytest.mark.gpu
def test_c_program_solves_like_the_ctypes_path(exe, case):
    import omgtools.backend as be
    tpl, P, path, d = case
    B = P['p'].shape[0]
    inp, out = str(d / 'in.bin'), str(d / 'out.bin')
    with open(inp, 'wb') as fp:
        fp.write(np.int32(B).tobytes())
        for a in (P['p'], P['x0'], tpl.lb, tpl.ub):
            fp.write(np.ascontiguousarray(a, dtype=np.float64).tobytes())
    subprocess.check_call([exe, path, 'solve', inp, out], timeout=120)
    raw = open(out, 'rb').read()
    nx, nl = B * tpl.n_var * 8, B * tpl.n_con * 8
    x = np.frombuffer(raw[:nx]).reshape(B, tpl.n_var)
    lam = np.frombuffer(raw[nx:nx + nl]).reshape(B, tpl.n_con)
    status = np.frombuffer(raw[nx + nl:nx + nl + 4 * B], dtype=np.int32)
    solver = be.BatchSolver(tpl, B, options=dict(tol=1e-6, max_iter=500))
    res = solver.solve(P['p'], P['x0'])
    solver.close()
    assert (status == 0).all() and np.array_equal(status, res['status'])
    assert np.array_equal(x, res['x']) and np.array_equal(lam, res['lam_g'])       # the same bits from C and from Python
