"""ORACLE (test infrastructure): ctypes access to oracle/_build/libomgx_port.so,
the host build of the solver core (oracle/port/omgx_port.cpp).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it."""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
PORT_PATH = os.path.join(_DIR, '_build', 'libomgx_port.so')


def build():
    subprocess.check_call(['make', '-C', _DIR, '-s'])


def load():
    if not os.path.exists(PORT_PATH):
        build()
    lib = C.CDLL(PORT_PATH)
    lib.omgx_port_solve.restype = C.c_int
    lib.omgx_port_solve_mt.restype = C.c_int
    return lib


def _backend():
    """The product's ctypes layer: `omgtools.backend`, or -- in a process where `omgtools` is the reference
    package running on the shim -- the same module under its shim name."""
    try:
        import omgtools.backend as be
        if hasattr(be, 'make_ctemplate'):
            return be
    except ImportError:
        pass
    import omgx_shim
    return omgx_shim._mod('backend')


def solve(template, p, x0, lbg=None, ubg=None, plan=None, lam_g0=None, status0=None, n_threads=1,
          dw_state=None, **options):
    """Solve B agents on `n_threads` host threads (one agent per thread at a time); returns a
    dict like BatchSolver.solve."""
    be = _backend()
    make_ctemplate, make_options = be.make_ctemplate, be.make_options
    lib = load()
    ct, keep = make_ctemplate(template, plan)
    opt = make_options(**options)
    p = np.ascontiguousarray(np.atleast_2d(np.asarray(p, float)))
    x0 = np.ascontiguousarray(np.atleast_2d(np.asarray(x0, float)))
    B = p.shape[0]
    lbg = np.ascontiguousarray(template.lb if lbg is None else lbg, dtype=float)
    ubg = np.ascontiguousarray(template.ub if ubg is None else ubg, dtype=float)
    shared = int(lbg.size == template.n_con)
    x = np.empty((B, template.n_var))
    lam = np.empty((B, template.n_con)) if lam_g0 is None else \
        np.ascontiguousarray(np.atleast_2d(np.asarray(lam_g0, float))).copy()
    status = np.zeros(B, dtype=np.int32) if status0 is None else \
        np.ascontiguousarray(status0, dtype=np.int32).copy()
    iters = np.empty(B, dtype=np.int32)
    rc = lib.omgx_port_solve_mt(C.byref(ct), C.byref(opt), C.c_int32(B),
                             C.c_void_p(p.ctypes.data), C.c_void_p(x0.ctypes.data),
                             C.c_void_p(lbg.ctypes.data), C.c_void_p(ubg.ctypes.data),
                             C.c_int32(shared), C.c_void_p(x.ctypes.data),
                             C.c_void_p(lam.ctypes.data), C.c_void_p(status.ctypes.data),
                             C.c_void_p(iters.ctypes.data), C.c_int32(int(n_threads)),
                             C.c_void_p(dw_state.ctypes.data if dw_state is not None else None))
    if rc != 0:
        raise RuntimeError('omgx_port_solve failed: %d' % rc)
    return dict(x=x, lam_g=lam, status=status, iters=iters)
