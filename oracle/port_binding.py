"""ORACLE (test infrastructure): ctypes access to oracle/_build/libomgx_port.so,
the host build of the solver core (oracle/port/omgx_port.cpp).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it."""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
PORT_PATH = os.environ.get('OMGX_PORT_LIB') or os.path.join(_DIR, '_build', 'libomgx_port.so')      # (OMGX_PORT_LIB: developer override, e.g. a build with other constants)


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
    # two-sided rows lb < g < ub: solved with every such row doubled, like the library does around its kernel (oracle/range_rows.py)
    from oracle import range_rows as rr
    if len(rr.range_rows(template)):
        t2, src, dup = rr.expand_template(template)
        lb2, ub2 = rr.expand_bounds(template.lb if lbg is None else lbg, template.ub if ubg is None else ubg, src, dup)
        shared = lb2.shape[0] == 1
        res = solve(t2, p, x0, lb2[0] if shared else lb2, ub2[0] if shared else ub2, None,
                    None if lam_g0 is None else rr.expand_lam(lam_g0, src, dup), status0, n_threads, dw_state, **options)
        res['lam_g'] = rr.contract_lam(res['lam_g'], template.n_con, src, dup)
        return res
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


# -- bench.py's cpu_baseline: persistent pinned pool, plan built once, step glue in C ---------------------
class StepDesc(C.Structure):
    _fields_ = [('o_spl', C.c_int32), ('n_dim', C.c_int32), ('L', C.c_int32), ('o_state0', C.c_int32),
                ('o_input0', C.c_int32), ('o_t', C.c_int32), ('t_rel', C.c_double), ('dt', C.c_double),
                ('E', C.c_void_p), ('Ed', C.c_void_p), ('n_obst', C.c_int32), ('obst', C.c_void_p),
                ('crossed', C.c_int32), ('n_shift', C.c_int32), ('shift_entries', C.c_void_p),
                ('shift_mats', C.c_void_p), ('perm', C.c_void_p)]


def physical_cpus():
    """One logical cpu per physical core (first sibling of every /sys topology group); all cpus the process
    may run on when the topology files are missing."""
    allowed = sorted(os.sched_getaffinity(0))
    seen, out = set(), []
    for cpu in allowed:
        path = '/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list' % cpu
        try:
            key = open(path).read().strip()
        except OSError:
            key = str(cpu)
        if key not in seen:
            seen.add(key)
            out.append(cpu)
    return out


def cpu_quota():
    """cpus the cgroup grants (cpu.max quota / period, cgroup v2; cfs_quota_us / cfs_period_us, v1); None = no limit."""
    try:
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()
        return None if q == 'max' else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
        per = float(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def granted_cpus():
    """The cpus a fair baseline runs on: one logical cpu per physical core, no more of them than the cgroup's
    cpu quota pays for (threads beyond the quota only get throttled: tools/cpu_pool_sweep.py)."""
    cpus = physical_cpus()
    quota = cpu_quota()
    if quota is not None:
        cpus = cpus[:max(1, int(quota + 0.5))]
    return cpus


class PortPool(object):
    """Host solver with the life cycle of the HIP handle: created once per template, reused by every step."""

    def __init__(self, template, cpus=None, n_threads=None):
        be = _backend()
        self.lib = lib = load()
        lib.omgx_port_pool_create.restype = C.c_void_p
        lib.omgx_port_pool_create.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        lib.omgx_port_pool_destroy.argtypes = [C.c_void_p]
        lib.omgx_port_pool_solve.restype = C.c_int
        lib.omgx_port_pool_solve.argtypes = [C.c_void_p] * 2 + [C.c_int32] + [C.c_void_p] * 9
        self.tpl, self._be = template, be
        ct, self._keep = be.make_ctemplate(template)
        if n_threads is None:
            cpus = granted_cpus() if cpus is None else list(cpus)
            n_threads = len(cpus)
        self.n_threads = int(n_threads)
        arr = np.ascontiguousarray(cpus, dtype=np.int32) if cpus is not None else None
        self._h = lib.omgx_port_pool_create(C.addressof(ct), self.n_threads, arr.ctypes.data if arr is not None else None)
        if not self._h:
            raise RuntimeError('omgx_port_pool_create failed')
        self.lb = np.ascontiguousarray(template.lb, dtype=float)
        self.ub = np.ascontiguousarray(template.ub, dtype=float)

    def solve(self, p, x, lam, status, iters, dw, step=None, **options):
        """In place on the caller's arrays (p, x, lam, status, iters, dw: C-contiguous, float64 / int32)."""
        opt = self._be.make_options(**options)
        rc = self.lib.omgx_port_pool_solve(self._h, C.addressof(opt), p.shape[0], p.ctypes.data, x.ctypes.data,
                                           self.lb.ctypes.data, self.ub.ctypes.data, lam.ctypes.data,
                                           status.ctypes.data, iters.ctypes.data, dw.ctypes.data,
                                           C.addressof(step) if step is not None else None)
        if rc != 0:
            raise RuntimeError('omgx_port_pool_solve failed: %d' % rc)

    def step_desc(self):
        return StepDesc()

    def close(self):
        if self._h:
            self.lib.omgx_port_pool_destroy(self._h)
            self._h = None
