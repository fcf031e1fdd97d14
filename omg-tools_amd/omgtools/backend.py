"""ctypes binding of libomgx.so (include/omgx.h) and the solver objects built on it.

`create_nlp` is the stand-in for the reference's solver factory
(`basics/optilayer.py:49-104`): it returns an object with the call shape the
reference's `Problem.solve` uses (`problems/problem.py:113-119`):

    result = solver(x0=, p=, lbg=, ubg=)   ->  {'x': ..., 'lam_g': ...}
    solver.stats()['return_status']        ->  'Solve_Succeeded' | ...

There is NO CPU fallback: if libomgx.so (hand-written HIP, csrc/) cannot be
loaded or no HIP device is present, construction raises.
"""
import ctypes as C
import os
import time

import numpy as np


_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('OMGX_LIB') or os.path.join(os.path.dirname(_HERE), 'csrc', 'libomgx.so')      # (OMGX_LIB: developer override, e.g. an experimental build)

PTR_DEVICE, BOUNDS_SHARED, BOUNDS_DEVICE = 1, 2, 4

STATUS_STRINGS = {0: 'Solve_Succeeded', 1: 'Maximum_Iterations_Exceeded',
                  2: 'Infeasible_Problem_Detected', 3: 'Unsupported_Bounds',
                  4: 'Numerical_Failure'}

_I32P, _F64P = C.POINTER(C.c_int32), C.POINTER(C.c_double)


class CTemplate(C.Structure):
    """`omgx_template` (include/omgx.h): the flat NLP; the solver plan is derived inside the library."""
    _fields_ = [(n, C.c_int32) for n in
                ('n_var', 'n_par', 'n_con', 'n_atoms', 'n_slots', 'n_terms',
                 'n_prog', 'n_knots', 'n_pp', 'n_mono', 'n_matom')] + \
               [('prog', _I32P), ('knots', _F64P), ('pp_ptr', _I32P), ('pm_coef', _F64P),
                ('pm_ptr', _I32P), ('pm_atom', _I32P), ('slot_pp', _I32P),
                ('row_ptr', _I32P), ('t_coef', _F64P), ('t_slot', _I32P), ('t_var', _I32P)] + \
               [('n_eq', C.c_int32), ('eq_rows', _I32P), ('n_root_vars', C.c_int32), ('root_vars', _I32P)] + \
               [('n_blocks', C.c_int32), ('block_names_len', C.c_int32), ('block_names', C.c_char_p),
                ('block_kind', _I32P), ('block_off', _I32P), ('block_rows', _I32P), ('block_cols', _I32P)] + \
               [('has_bounds', C.c_int32), ('lbg_def', _F64P), ('ubg_def', _F64P)] + \
               [('n_lift', C.c_int32), ('lift_row0', C.c_int32)]


class CPlanInfo(C.Structure):
    """`omgx_plan_info` (include/omgx.h)."""
    _fields_ = [(n, C.c_int32) for n in ('n_leaf', 'n_root', 'n_eq', 'nnz_j', 'kkt_doubles',
                                         'wave_path', 'ws_mode')] + \
               [('lds_bytes', C.c_int64)] + \
               [(n, C.c_int32 * 16) for n in ('leaf_size', 'leaf_bw', 'leaf_cpl')] + \
               [(n, C.c_int32) for n in ('n_pairs', 'ka_len', 'kh_len', 'kg_len')]


class COptions(C.Structure):
    _fields_ = [('tol', C.c_double), ('max_iter', C.c_int32), ('mu_init', C.c_double),
                ('kappa_push', C.c_double), ('nu_init', C.c_double), ('scale_gmax', C.c_double),
                ('warm_start', C.c_int32), ('kappa_warm', C.c_double),
                ('dw_leaf_ratio_cold', C.c_double), ('warm_mu_factor', C.c_double), ('warm_z_floor', C.c_double), ('warm_z_cap', C.c_double), ('max_soc', C.c_int32), ('hess_approx', C.c_int32),
                ('compl_inf_tol', C.c_double), ('constr_viol_tol', C.c_double), ('refine', C.c_int32)]


class CRolloutSpec(C.Structure):
    """include/omgx.h omgx_rollout_spec"""
    _fields_ = [('n_steps', C.c_int32), ('tau', C.c_void_p), ('t_rel', C.c_void_p), ('crossed', C.c_void_p),
                ('coeff_off', C.c_int32), ('n_spl', C.c_int32), ('degree', C.c_int32), ('n_knots', C.c_int32), ('n_out', C.c_int32),
                ('knots', C.c_void_p), ('p_off', C.c_void_p), ('p_t', C.c_int32), ('inv_T', C.c_double),
                ('n_obst', C.c_int32), ('obst', C.c_void_p), ('dt', C.c_double),
                ('shift_entries', C.c_void_p), ('n_ent', C.c_int32), ('shift_T', C.c_void_p), ('n_tmat', C.c_int32),
                ('lam_perm', C.c_void_p), ('cross_options', C.c_void_p), ('iters_log', C.c_void_p), ('status_log', C.c_void_p)]


DEFAULT_OPTIONS = dict(tol=1e-3, max_iter=300, mu_init=0.1, kappa_push=1.0,
                       nu_init=100.0, scale_gmax=100.0, warm_start=0, kappa_warm=1e-3,
                       dw_leaf_ratio_cold=1.0, warm_mu_factor=1.0, warm_z_floor=0.1, warm_z_cap=0.01, max_soc=1, hess_approx=0, compl_inf_tol=0.0, constr_viol_tol=0.0, refine=0)


def make_options(**kw):
    vals = dict(DEFAULT_OPTIONS)
    vals.update(kw)
    return COptions(**vals)


def options_from_problem(options):
    """Map the reference's option names onto the HIP solver's settings
    (`problems/problem.py:54-62`: ipopt.tol, ipopt.max_iter)."""
    ipopt = options.get('solver_options', {}).get(options.get('solver', 'ipopt'), {})
    kw = {}
    if 'ipopt.tol' in ipopt:
        kw['tol'] = float(ipopt['ipopt.tol'])
    if 'ipopt.max_iter' in ipopt:
        kw['max_iter'] = int(ipopt['ipopt.max_iter'])
    # (`examples/p2p_dubins.py:42`, `p2p_agv.py:43` ask IPOPT for a limited-memory Hessian on the nonholonomic classes: here the
    # Hessian without the curvature of the rows, damped by the accepted step length -- include/omgx.h `hess_approx`)
    if ipopt.get('ipopt.hessian_approximation') == 'limited-memory':
        kw['hess_approx'] = 1
    # (IPOPT's absolute tolerances on the unscaled problem, when a caller sets them -- include/omgx.h; left alone, the scaled error
    # `tol` alone decides, as in rounds 1-5: IPOPT_DEFAULT_TOLERANCES below is what IPOPT itself would keep in force)
    for key, name in (('ipopt.compl_inf_tol', 'compl_inf_tol'), ('ipopt.constr_viol_tol', 'constr_viol_tol')):
        if key in ipopt:
            kw[name] = float(ipopt[key])
    kw.update(options.get('omgx', {}))
    kw.pop('hess_fallback', None)      # (a switch of the drop-in solver object, `NlpSolver`: not a solver setting)
    return kw


def template_is_general(tpl):
    """True for templates the library runs on its general kernel instance (omgx_plan.h `general`: lifted auxiliaries, terms with four
    variable factors, cos / sin atoms) -- the instances that honour `hess_approx`."""
    if getattr(tpl, 'n_lift', 0):
        return True
    t_var = np.asarray(tpl.t_var).reshape(-1, 4) if np.asarray(tpl.t_var).ndim == 1 else np.asarray(tpl.t_var)
    if t_var.shape[1] >= 4 and (t_var[:, 3] >= 0).any():
        return True
    prog = np.asarray(tpl.prog).reshape(-1, 6)
    return bool(len(prog) and np.isin(prog[:, 0], (2, 3)).any())


# IPOPT's documented defaults of the two absolute tolerances: what `nlpsol('ipopt')` with the reference's options (ipopt.tol = 1e-3
# only, `problems/problem.py:57`) tests beside the scaled error -- `dict(tol=1e-3, **IPOPT_DEFAULT_TOLERANCES)` is the like-for-like setting
IPOPT_DEFAULT_TOLERANCES = dict(compl_inf_tol=1e-4, constr_viol_tol=1e-4)

ROOT_HINT = ('splines_seg',)


def root_hint_vars(tpl, root_hint=ROOT_HINT):
    """Variables handed to the library as the root of the block-arrow KKT matrix: the vehicle's spline
    coefficients (`vehicles/vehicle.py:105-120`); every other variable meets the rest of the problem
    through them."""
    idx = []
    for (label, name), (off, r, c) in tpl.var_layout.items():
        if any(name.startswith(h) for h in root_hint):
            idx.extend(range(off, off + r * c))
    return np.array(idx, dtype=np.int32)


def make_ctemplate(tpl, plan=None):
    """(CTemplate, keepalive list).  `plan` is accepted for backward compatibility and ignored: the
    solver plan is derived inside the library (csrc/omgx_plan.h)."""
    keep = []

    def i32(a):
        a = np.ascontiguousarray(a, dtype=np.int32)
        keep.append(a)
        return a.ctypes.data_as(_I32P)

    def f64(a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        keep.append(a)
        return a.ctypes.data_as(_F64P)

    ct = CTemplate()
    ct.n_var, ct.n_par, ct.n_con = tpl.n_var, tpl.n_par, tpl.n_con
    ct.n_atoms, ct.n_slots, ct.n_terms = tpl.n_atoms, tpl.n_slots, tpl.n_terms
    ct.n_prog, ct.n_knots = len(tpl.prog), len(tpl.knots)
    ct.n_pp, ct.n_mono, ct.n_matom = len(tpl.pp_ptr) - 1, len(tpl.pm_coef), len(tpl.pm_atom)
    ct.prog, ct.knots = i32(tpl.prog.reshape(-1)), f64(np.r_[tpl.knots, 0.0])
    ct.pp_ptr, ct.pm_coef = i32(tpl.pp_ptr), f64(np.r_[tpl.pm_coef, 0.0])
    ct.pm_ptr, ct.pm_atom = i32(tpl.pm_ptr), i32(np.r_[tpl.pm_atom, 0])
    ct.slot_pp = i32(np.r_[tpl.slot_pp, 0])
    ct.row_ptr, ct.t_coef = i32(tpl.row_ptr), f64(np.r_[tpl.t_coef, 0.0])
    ct.t_slot, ct.t_var = i32(np.r_[tpl.t_slot, 0]), i32(np.r_[tpl.t_var.reshape(-1), 0])
    eq_rows = np.nonzero(np.isfinite(tpl.lb) & (tpl.lb == tpl.ub))[0]
    ct.n_eq, ct.eq_rows = len(eq_rows), i32(np.r_[eq_rows, 0])
    root = root_hint_vars(tpl)
    ct.n_root_vars, ct.root_vars = len(root), i32(np.r_[root, 0])
    # block table: name, kind, offset, shape of every entry of x / p / g, in the reference's struct order
    # (`basics/optilayer.py:225-272`; the offsets `export/export.py:302-353` hard-codes into the generated C++)
    names, kinds, offs, rows, cols = [], [], [], [], []
    for kind, which in enumerate(('var', 'par', 'con')):
        for label, name, off, r, c in tpl.block_table(which):
            names.append('%s.%s' % (label, name)); kinds.append(kind); offs.append(off); rows.append(r); cols.append(c)
    blob = b''.join(n.encode() + b'\0' for n in names)
    keep.append(blob)
    ct.n_blocks, ct.block_names_len, ct.block_names = len(names), len(blob), blob
    ct.block_kind, ct.block_off = i32(np.r_[kinds, 0]), i32(np.r_[offs, 0])
    ct.block_rows, ct.block_cols = i32(np.r_[rows, 0]), i32(np.r_[cols, 0])
    # default bounds of g (LBG_DEF / UBG_DEF of the reference's generated C++)
    ct.has_bounds, ct.lbg_def, ct.ubg_def = 1, f64(tpl.lb), f64(tpl.ub)
    # lifted products / quotients (template.py `_append_lifted`): the last variables and rows
    ct.n_lift = int(getattr(tpl, 'n_lift', 0))
    ct.lift_row0 = int(getattr(tpl, 'lift_row0', tpl.n_con - ct.n_lift))
    return ct, keep


class CStoreSpec(C.Structure):
    """include/omgx.h omgx_store_spec"""
    _fields_ = [('out', C.c_void_p), ('v_tot', C.c_void_p), ('t0', C.c_void_p), ('knots', C.c_void_p),
                ('coeff_off', C.c_int32), ('n_spl', C.c_int32), ('degree', C.c_int32), ('n_knots', C.c_int32),
                ('n_der', C.c_int32), ('n_samp', C.c_int32), ('dt', C.c_double), ('inv_T', C.c_double)]


PREDICT_IDEAL, PREDICT_RK4 = 0, 1
ONLY_FAILED = 8


def save_template(tpl, path, lib=None):
    """Write the template file C/C++ callers load with `omgx_template_read` (include/omgx.h): what the
    reference's exporter does with the generated nlp.so (`export/export.py:236-262`)."""
    lib = lib or load_library()
    ct, keep = make_ctemplate(tpl)
    lib.omgx_template_write.argtypes = [C.POINTER(CTemplate), C.c_char_p]
    _check(lib, lib.omgx_template_write(C.byref(ct), os.fsencode(path)), 'omgx_template_write')
    return path


def read_template_counts(path, lib=None):
    """The counts of a template file as `omgx_template_read` returns them (the arrays stay in the library)."""
    lib = lib or load_library()
    out = C.POINTER(CTemplate)()
    lib.omgx_template_read.argtypes = [C.c_char_p, C.POINTER(C.POINTER(CTemplate))]
    _check(lib, lib.omgx_template_read(os.fsencode(path), C.byref(out)), 'omgx_template_read')
    t = out.contents
    counts = {k: int(getattr(t, k)) for k in ('n_var', 'n_par', 'n_con', 'n_atoms', 'n_slots', 'n_terms', 'n_eq', 'n_blocks',
                                               'has_bounds', 'n_lift', 'lift_row0')}
    lib.omgx_template_free.argtypes = [C.POINTER(CTemplate)]
    lib.omgx_template_free.restype = None
    lib.omgx_template_free(out)
    return counts


def describe_plan(tpl, lib=None):
    """What the library derives from the template (host only, no device needed): dict with the leaf
    sizes / bandwidths / coupling counts, the root size and `order` (position -> variable)."""
    lib = lib or load_library()
    ct, keep = make_ctemplate(tpl)
    info = CPlanInfo()
    order = np.zeros(tpl.n_var + 1, dtype=np.int32)
    lib.omgx_plan_describe.argtypes = [C.POINTER(CTemplate), C.POINTER(CPlanInfo), C.c_void_p]
    _check(lib, lib.omgx_plan_describe(C.byref(ct), C.byref(info), order.ctypes.data), 'omgx_plan_describe')
    nl = info.n_leaf
    return dict(n_leaf=nl, n_root=info.n_root, n_eq=info.n_eq, nnz_j=info.nnz_j, kkt_doubles=info.kkt_doubles,
                wave_path=info.wave_path, ws_mode=info.ws_mode, lds_bytes=info.lds_bytes,
                leaf_sizes=list(info.leaf_size[:nl]), leaf_bw=list(info.leaf_bw[:nl]),
                leaf_cpl=list(info.leaf_cpl[:nl]), order=order, n_pairs=info.n_pairs,
                ka_len=info.ka_len, kh_len=info.kh_len, kg_len=info.kg_len)


_lib = None


def load_library(path=None):
    """Load the HIP library; raises OSError with a build hint when missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise OSError('%s not found: build it with `python __graft_entry__.py` or '
                      '`make -C omg-tools_amd/csrc` (hipcc, gfx950). There is no CPU '
                      'fallback for the solve path.' % path)
    # One ROCm runtime per process: torch ships its own libamdhip64, and whichever copy initialises the
    # device second reports "No HIP GPUs are available".  Importing torch first makes the loader bind
    # libomgx.so to the copy torch uses (torch is the allocator / stream / RCCL host of this package
    # anyway); without torch the library stands alone on /opt/rocm's runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(path)
    lib.omgx_version.restype = C.c_int
    lib.omgx_last_error.restype = C.c_char_p
    lib.omgx_status_string.restype = C.c_char_p
    lib.omgx_status_string.argtypes = [C.c_int32]
    lib.omgx_default_options.argtypes = [C.POINTER(COptions)]
    lib.omgx_batch_create.argtypes = [C.POINTER(CTemplate), C.c_int32, C.c_int32,
                                      C.POINTER(C.c_void_p)]
    lib.omgx_batch_destroy.argtypes = [C.c_void_p]
    lib.omgx_batch_destroy.restype = None
    lib.omgx_batch_set_options.argtypes = [C.c_void_p, C.POINTER(COptions)]
    lib.omgx_batch_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    lib.omgx_batch_lds_bytes.argtypes = [C.c_void_p]
    lib.omgx_batch_set_order.argtypes = [C.c_void_p, C.c_void_p]
    lib.omgx_batch_order_by_iters.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.omgx_batch_workspace.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int64),
                                         C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
    lib.omgx_batch_solve.argtypes = [C.c_void_p] + [C.c_void_p] * 8 + [C.c_int32]
    lib.omgx_batch_sync.argtypes = [C.c_void_p]
    lib.omgx_batch_transfer.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.omgx_batch_last_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    lib.omgx_batch_shift.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_int32, C.c_void_p, C.c_int32, C.c_int32]
    lib.omgx_batch_predict.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                       C.c_void_p, C.c_int32, C.c_double, C.c_double, C.c_int32, C.c_int32,
                                       C.c_int32, C.c_double]
    lib.omgx_batch_sample.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                      C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_double,
                                      C.c_int32, C.c_void_p, C.c_int32, C.c_int32]
    lib.omgx_batch_predict_ex.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                          C.c_void_p, C.c_int32, C.c_double, C.c_double, C.c_int32, C.c_void_p,
                                          C.c_int32, C.c_double, C.c_int32, C.c_void_p, C.c_int32, C.c_double]
    lib.omgx_batch_predict_quadrotor.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                                 C.c_double, C.c_double, C.c_int32, C.c_void_p, C.c_int32, C.c_double, C.c_void_p,
                                                 C.c_void_p, C.c_int32, C.c_double, C.c_double]
    lib.omgx_batch_store.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(CStoreSpec)]
    lib.omgx_batch_set_store.argtypes = [C.c_void_p, C.POINTER(CStoreSpec)]
    if path == LIB_PATH:
        _lib = lib
    return lib


class OmgxError(RuntimeError):
    pass


def _check(lib, rc, what):
    if rc != 0:
        raise OmgxError('%s failed (%d): %s' % (what, rc, lib.omgx_last_error().decode()))


class BatchSolver(object):
    """Owns one `omgx_batch` handle: B agents sharing one NLP template.

    Host-array interface (`solve`) for the drop-in path and the tests;
    device-pointer interface (`solve_device`) for resident data (bench.py, the
    batched deployer) where nothing crosses PCIe inside the timed region.
    """

    def __init__(self, template, n_agents, device=0, options=None, plan=None):
        self.lib = load_library()
        self.template = template
        self.n_agents = int(n_agents)
        self._ct, self._keep = make_ctemplate(template)
        self._h = C.c_void_p()
        _check(self.lib, self.lib.omgx_batch_create(C.byref(self._ct), self.n_agents,
                                                    int(device), C.byref(self._h)),
               'omgx_batch_create')
        self.set_options(**(options or {}))

    def set_options(self, **kw):
        merged = dict(DEFAULT_OPTIONS)
        merged.update(getattr(self, 'options', {}))
        merged.update(kw)
        self.options = merged
        opt = COptions(**self.options)
        _check(self.lib, self.lib.omgx_batch_set_options(self._h, C.byref(opt)),
               'omgx_batch_set_options')

    def set_stream(self, stream_ptr):
        _check(self.lib, self.lib.omgx_batch_set_stream(self._h, C.c_void_p(stream_ptr)),
               'omgx_batch_set_stream')

    @property
    def lds_bytes(self):
        return self.lib.omgx_batch_lds_bytes(self._h)

    def set_order(self, order):
        """Launch order of the next solves: int32 device tensor / pointer (None = identity)."""
        ptr = None if order is None else (order.data_ptr() if hasattr(order, 'data_ptr') else int(order))
        self._order_keep = order
        _check(self.lib, self.lib.omgx_batch_set_order(self._h, ptr), 'omgx_batch_set_order')

    def order_by_iters(self, iters, order):
        """order <- agents sorted by their previous iteration count (largest first); becomes the launch order."""
        self._order_keep = order
        _check(self.lib, self.lib.omgx_batch_order_by_iters(self._h, iters.data_ptr(), order.data_ptr()),
               'omgx_batch_order_by_iters')

    def workspace(self):
        """Placement chosen by the library: mode 0 = all per-agent arrays in LDS, 1..3 = KKT /
        Jacobian / row arrays spilled to an HBM slab per persistent workgroup."""
        mode, nslab = C.c_int32(), C.c_int32()
        lds, hbm = C.c_int64(), C.c_int64()
        _check(self.lib, self.lib.omgx_batch_workspace(self._h, C.byref(mode), C.byref(lds), C.byref(hbm),
                                                       C.byref(nslab)), 'omgx_batch_workspace')
        return dict(mode=mode.value, lds_bytes=lds.value, hbm_bytes_per_slab=hbm.value, n_slabs=nslab.value)

    def close(self):
        if getattr(self, '_h', None) is not None and self._h:
            self.lib.omgx_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- host arrays ----------------------------------------------------------------
    def solve(self, p, x0, lbg=None, ubg=None, lam_g0=None, status0=None):
        """lam_g0 is used (and required) when the option warm_start is set."""
        t = self.template
        B = self.n_agents
        p = np.ascontiguousarray(np.asarray(p, float).reshape(B, t.n_par))
        x0 = np.ascontiguousarray(np.asarray(x0, float).reshape(B, t.n_var))
        lbg = t.lb if lbg is None else lbg
        ubg = t.ub if ubg is None else ubg
        lbg = np.ascontiguousarray(np.asarray(lbg, float))
        ubg = np.ascontiguousarray(np.asarray(ubg, float))
        shared = lbg.size == t.n_con
        flags = BOUNDS_SHARED if shared else 0
        if lbg.size != (t.n_con if shared else B * t.n_con) or ubg.size != lbg.size:
            raise ValueError('lbg/ubg have the wrong size')
        x = np.empty((B, t.n_var))
        lam = np.empty((B, t.n_con)) if lam_g0 is None else \
            np.ascontiguousarray(np.asarray(lam_g0, float).reshape(B, t.n_con)).copy()
        if self.options.get('warm_start') and lam_g0 is None:
            raise ValueError('warm_start is set: pass lam_g0')
        status = np.zeros(B, dtype=np.int32) if status0 is None else \
            np.ascontiguousarray(status0, dtype=np.int32).copy()
        iters = np.empty(B, dtype=np.int32)
        _check(self.lib, self.lib.omgx_batch_solve(
            self._h, p.ctypes.data, x0.ctypes.data, lbg.ctypes.data, ubg.ctypes.data,
            x.ctypes.data, lam.ctypes.data, status.ctypes.data, iters.ctypes.data, flags),
            'omgx_batch_solve')
        return dict(x=x, lam_g=lam, status=status, iters=iters)

    # -- device pointers (torch tensors / raw ints) -------------------------------------
    def solve_device(self, p, x0, lbg, ubg, x, lam_g, status, iters, bounds_shared=True, only_failed=False):
        """only_failed: restart pass -- agents whose `status` is 0 keep x / lam_g / status / iters, the others are
        solved from x0 (OMGX_ONLY_FAILED)."""
        def ptr(a):
            return a.data_ptr() if hasattr(a, 'data_ptr') else int(a)
        flags = PTR_DEVICE | BOUNDS_DEVICE | (BOUNDS_SHARED if bounds_shared else 0) | (ONLY_FAILED if only_failed else 0)
        _check(self.lib, self.lib.omgx_batch_solve(
            self._h, ptr(p), ptr(x0), ptr(lbg), ptr(ubg), ptr(x), ptr(lam_g), ptr(status),
            ptr(iters), flags), 'omgx_batch_solve')

    def set_restarts(self, x0_alt=None, attempts=None):
        """Restart guesses of the following cold device solves (include/omgx.h omgx_batch_set_restarts): x0_alt
        [n_alt, B, n_var] device tensor or None, attempts [B] int32 device tensor or None.  The caller keeps both
        alive until the solves are done."""
        self.lib.omgx_batch_set_restarts.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
        n_alt = 0 if x0_alt is None else int(x0_alt.shape[0])
        if n_alt and (tuple(x0_alt.shape[1:]) != (self.n_agents, self.template.n_var) or not x0_alt.is_contiguous()):
            raise ValueError("x0_alt must be a contiguous [n_alt, %d, %d] tensor" % (self.n_agents, self.template.n_var))
        _check(self.lib, self.lib.omgx_batch_set_restarts(
            self._h, x0_alt.data_ptr() if n_alt else None, n_alt,
            attempts.data_ptr() if attempts is not None else None), 'omgx_batch_set_restarts')
        self._restart_keep = (x0_alt, attempts)

    def set_stats(self, stats=None):
        """Launch statistics on the device (include/omgx.h omgx_batch_set_stats): stats [n_slots, 4] int64 device
        tensor (zeroed by the caller, kept alive by it) or None; row k % n_slots of the k-th following solve gets
        {solved agents, sum of iterations, largest iteration count, agents solved}."""
        self.lib.omgx_batch_set_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        if stats is not None and (stats.dim() != 2 or stats.shape[1] != 4 or not stats.is_contiguous() or stats.element_size() != 8):
            raise ValueError('stats must be a contiguous [n_slots, 4] int64 tensor')
        _check(self.lib, self.lib.omgx_batch_set_stats(self._h, stats.data_ptr() if stats is not None else None,
                                                        int(stats.shape[0]) if stats is not None else 0), 'omgx_batch_set_stats')
        self._stats_keep = stats

    def set_launch_events(self, start, stop):
        """Attach two timing events to the next solve launch (include/omgx.h omgx_batch_set_launch_events): torch.cuda
        events (already recorded once, so that their handles exist) or raw hipEvent_t handles.  They get the begin /
        end stamps of the solve kernel itself; nothing else is put on the stream."""
        def handle(e):
            h = e.cuda_event if hasattr(e, 'cuda_event') else int(e)
            if not h:
                raise ValueError('the event has no handle yet: record it once before handing it over')
            return h
        self.lib.omgx_batch_set_launch_events.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _check(self.lib, self.lib.omgx_batch_set_launch_events(self._h, handle(start), handle(stop)),
               'omgx_batch_set_launch_events')

    def eval(self, p, x, lam_g):
        """Verification entry (`omgx_batch_eval`): g, f, dense Jacobian of (g, f) and dense Hessian of f + lam_g' g at
        x, evaluated by the device code of the solve from its own tables."""
        t, B = self.template, self.n_agents
        p = np.ascontiguousarray(np.asarray(p, float).reshape(B, t.n_par))
        x = np.ascontiguousarray(np.asarray(x, float).reshape(B, t.n_var))
        lam = np.ascontiguousarray(np.asarray(lam_g, float).reshape(B, t.n_con))
        g, f = np.empty((B, t.n_con)), np.empty(B)
        jac, hess = np.empty((B, t.n_con + 1, t.n_var)), np.empty((B, t.n_var, t.n_var))
        self.lib.omgx_batch_eval.argtypes = [C.c_void_p] * 8
        _check(self.lib, self.lib.omgx_batch_eval(self._h, p.ctypes.data, x.ctypes.data, lam.ctypes.data, g.ctypes.data,
                                                  f.ctypes.data, jac.ctypes.data, hess.ctypes.data), 'omgx_batch_eval')
        return dict(g=g, f=f, jac=jac, hess=hess)

    def transfer_plan(self, pairs):
        """A prepared `transfer` of fixed tensors: the argument checks (pinned? contiguous?) and the ctypes arrays once, every
        `run()` one library call -- for per-step loops (`is_pinned()` alone asks the runtime about the pointer every time)."""
        n = len(pairs)
        srcs, dsts, nbytes = (C.c_void_p * n)(), (C.c_void_p * n)(), (C.c_int64 * n)()
        for i, (dst, src) in enumerate(pairs):
            if dst.numel() != src.numel() or dst.dtype != src.dtype or not dst.is_contiguous() or not src.is_contiguous():
                raise ValueError('transfer: contiguous tensors of equal size and type')
            for t in (dst, src):
                if not (t.is_cuda or t.is_pinned()):
                    raise ValueError('transfer: device or pinned host tensors only')
            srcs[i], dsts[i], nbytes[i] = src.data_ptr(), dst.data_ptr(), src.numel() * src.element_size()
        solver, keep = self, list(pairs)                    # (the tensors stay alive with the plan)

        class Plan(object):
            def run(self_):
                _check(solver.lib, solver.lib.omgx_batch_transfer(solver._h, C.c_int32(n), srcs, dsts, nbytes), 'omgx_batch_transfer')
            tensors = keep
        return Plan()

    def transfer(self, pairs):
        """[(dst, src), ...] (at most 6): dst <- src for tensors that live in device memory or in PINNED host memory, by one small
        kernel on the handle's stream (`omgx_batch_transfer`: no copy-engine hand-over; a pinned tensor is read / written over
        the host link by the kernel itself)."""
        self.transfer_plan(pairs).run()

    def sync(self):
        _check(self.lib, self.lib.omgx_batch_sync(self._h), 'omgx_batch_sync')

    def set_timing(self, on):
        """Event records around every solve kernel (needed by last_kernel_ms; ~20 us of stream time per solve)."""
        self.lib.omgx_batch_set_timing.argtypes = [C.c_void_p, C.c_int32]
        _check(self.lib, self.lib.omgx_batch_set_timing(self._h, int(bool(on))), 'omgx_batch_set_timing')

    def set_prepare(self, on):
        """The setup of every solve as ONE launch for the whole batch ahead of the solve kernel (`omgx_batch_set_prepare`, ABI 8;
        off by default -- measured slower on the benchmark batch): off = every solve does its own setup inside the solve kernel
        (the same statements, the same bits)."""
        self.lib.omgx_batch_set_prepare.argtypes = [C.c_void_p, C.c_int32]
        _check(self.lib, self.lib.omgx_batch_set_prepare(self._h, int(bool(on))), 'omgx_batch_set_prepare')

    def set_stop(self, o_state0=0, o_input0=0, o_poseT=0, n_dim=0, stop_tol=1e-3, under_way=None):
        """The reference's stop criterion inside the solve launch (include/omgx.h omgx_batch_set_stop, ABI 9): under_way [B] int32
        device tensor (kept alive by the caller too), 1 = the agent's loop is running; None switches the rule off."""
        self.lib.omgx_batch_set_stop.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_void_p]
        if under_way is not None and (under_way.dim() != 1 or under_way.shape[0] != self.n_agents or not under_way.is_contiguous()
                                      or under_way.element_size() != 4 or under_way.is_floating_point()):
            raise ValueError('under_way must be a contiguous [n_agents] int32 device tensor')
        _check(self.lib, self.lib.omgx_batch_set_stop(self._h, int(o_state0), int(o_input0), int(o_poseT), int(n_dim), float(stop_tol),
                                                       under_way.data_ptr() if under_way is not None else None), 'omgx_batch_set_stop')
        self._under_way_keep = under_way

    def last_kernel_ms(self):
        ms = C.c_double()
        _check(self.lib, self.lib.omgx_batch_last_kernel_ms(self._h, C.byref(ms)),
               'omgx_batch_last_kernel_ms')
        return ms.value

    def predict(self, x, p, coeff_off, n_spl, degree, knots, tau, inv_T, p_state0, p_input0, p_t, t_value):
        """Ideal prediction on device-resident x / p (torch tensors or raw device pointers)."""
        knots = np.ascontiguousarray(knots, dtype=np.float64)

        def ptr(a):
            return a.data_ptr() if hasattr(a, 'data_ptr') else int(a)
        _check(self.lib, self.lib.omgx_batch_predict(
            self._h, ptr(x), ptr(p), int(coeff_off), int(n_spl), int(degree), knots.ctypes.data, len(knots),
            float(tau), float(inv_T), int(p_state0), int(p_input0), int(p_t), float(t_value)),
            'omgx_batch_predict')

    def predict_ex(self, x, p, coeff_off, n_spl, degree, knots, tau, inv_T, p_off, p_t, t_value,
                   mode=PREDICT_IDEAL, state_in=None, n_sub=0, dtau=0.0):
        """p[p_off[o] + k] <- o-th time derivative of spline k at tau (o < len(p_off), -1 skips); mode
        PREDICT_RK4: p[p_off[0] + k] <- state_in integrated over the n_sub sample intervals that end at tau
        (include/omgx.h omgx_batch_predict_ex).  Device-resident tensors / pointers."""
        knots = np.ascontiguousarray(knots, dtype=np.float64)
        off = np.ascontiguousarray(p_off, dtype=np.int32)

        def ptr(a):
            return None if a is None else (a.data_ptr() if hasattr(a, 'data_ptr') else int(a))
        _check(self.lib, self.lib.omgx_batch_predict_ex(
            self._h, ptr(x), ptr(p), int(coeff_off), int(n_spl), int(degree), knots.ctypes.data, len(knots),
            float(tau), float(inv_T), len(off), off.ctypes.data, int(p_t), float(t_value), int(mode), ptr(state_in),
            int(n_sub), float(dtau)), 'omgx_batch_predict_ex')

    def rollout(self, p, x, lbg, ubg, lam_g, status, iters, tau, t_rel, crossed, coeff_off, n_spl, degree, knots, inv_T, p_off, p_t,
                obstacles=(), dt=0.0, shift_entries=None, shift_T=None, lam_perm=None, cross_options=None, iters_log=None,
                status_log=None, bounds_shared=True):
        """n_steps = len(tau) receding-horizon steps of every agent in one launch (include/omgx.h omgx_batch_rollout): device
        tensors p, x, lam_g, status, iters updated in place; tau / t_rel / crossed: per-step host arrays; the rest as
        `predict_ex` / `shift`; obstacles: [(p_x, p_v, p_a, n_dim)] of the ones that move; lam_perm: multiplier map of a crossing."""
        def ptr(a):
            return None if a is None else (a.data_ptr() if hasattr(a, 'data_ptr') else int(a))
        keep = dict(tau=np.ascontiguousarray(tau, dtype=np.float64), t_rel=np.ascontiguousarray(t_rel, dtype=np.float64),
                    crossed=np.ascontiguousarray(crossed, dtype=np.uint8), knots=np.ascontiguousarray(knots, dtype=np.float64),
                    p_off=np.ascontiguousarray(p_off, dtype=np.int32),
                    obst=np.ascontiguousarray(np.array(list(obstacles), dtype=np.int32).reshape(-1, 4)))
        sp = CRolloutSpec()
        sp.n_steps = len(keep['tau'])
        sp.tau, sp.t_rel, sp.crossed = keep['tau'].ctypes.data, keep['t_rel'].ctypes.data, keep['crossed'].ctypes.data
        sp.coeff_off, sp.n_spl, sp.degree, sp.n_knots, sp.n_out = int(coeff_off), int(n_spl), int(degree), len(keep['knots']), len(keep['p_off'])
        sp.knots, sp.p_off, sp.p_t, sp.inv_T = keep['knots'].ctypes.data, keep['p_off'].ctypes.data, int(p_t), float(inv_T)
        sp.n_obst, sp.obst, sp.dt = len(keep['obst']), (keep['obst'].ctypes.data if len(keep['obst']) else None), float(dt)
        if shift_entries is not None and len(shift_entries):
            keep['ents'] = np.ascontiguousarray(shift_entries, dtype=np.int32)
            keep['mats'] = np.ascontiguousarray(shift_T, dtype=np.float64)
            keep['perm'] = np.ascontiguousarray(lam_perm, dtype=np.int32)
            sp.shift_entries, sp.n_ent, sp.shift_T, sp.n_tmat = keep['ents'].ctypes.data, len(keep['ents']), keep['mats'].ctypes.data, keep['mats'].size
            sp.lam_perm = keep['perm'].ctypes.data
        if cross_options:
            keep['copt'] = COptions(**dict(self.options, **cross_options))
            sp.cross_options = C.addressof(keep['copt'])
        sp.iters_log, sp.status_log = ptr(iters_log), ptr(status_log)
        self.lib.omgx_batch_rollout.argtypes = [C.c_void_p, C.POINTER(CRolloutSpec)] + [C.c_void_p] * 7 + [C.c_int32]
        flags = PTR_DEVICE | BOUNDS_DEVICE | (BOUNDS_SHARED if bounds_shared else 0)
        _check(self.lib, self.lib.omgx_batch_rollout(self._h, C.byref(sp), ptr(p), ptr(x), ptr(lbg), ptr(ubg), ptr(lam_g), ptr(status),
                                                     ptr(iters), flags), 'omgx_batch_rollout')

    def predict_quadrotor(self, x, p, coeff_off, degree, knots, tau, inv_T, p_off, p_t, t_value, state_in, state_out, n_sub, dtau, g=9.81):
        """Non-ideal prediction of the Quadrotor model: state_in [B, 5] integrated over the n_sub sample intervals that end at tau
        with the inputs the plan holds there (include/omgx.h omgx_batch_predict_quadrotor).  Device tensors / pointers."""
        knots = np.ascontiguousarray(knots, dtype=np.float64)
        off = np.ascontiguousarray(p_off, dtype=np.int32)

        def ptr(a):
            return None if a is None else (a.data_ptr() if hasattr(a, 'data_ptr') else int(a))
        _check(self.lib, self.lib.omgx_batch_predict_quadrotor(
            self._h, ptr(x), ptr(p), int(coeff_off), int(degree), knots.ctypes.data, len(knots), float(tau), float(inv_T), len(off),
            off.ctypes.data, int(p_t), float(t_value), ptr(state_in), ptr(state_out), int(n_sub), float(dtau), float(g)),
            'omgx_batch_predict_quadrotor')

    def _store_spec(self, out, v_tot, t0, coeff_off, n_spl, degree, knots, n_der, n_samp, dt, inv_T):
        knots = np.ascontiguousarray(knots, dtype=np.float64)
        sp = CStoreSpec(out.data_ptr(), v_tot.data_ptr() if v_tot is not None else None, t0.data_ptr(),
                        knots.ctypes.data, int(coeff_off), int(n_spl), int(degree), len(knots), int(n_der),
                        int(n_samp), float(dt), float(inv_T))
        sp._keep = (knots, out, v_tot, t0)
        return sp

    def store(self, x, out, v_tot, t0, coeff_off, n_spl, degree, knots, n_der, n_samp, dt, inv_T):
        """`Vehicle.store` of the batch on device tensors: out [B, n_der, n_spl, n_samp] time derivatives on the
        grid t0[b] + i dt (spline domain), v_tot [B, n_samp] or None."""
        sp = self._store_spec(out, v_tot, t0, coeff_off, n_spl, degree, knots, n_der, n_samp, dt, inv_T)
        _check(self.lib, self.lib.omgx_batch_store(self._h, x.data_ptr(), C.byref(sp)), 'omgx_batch_store')

    def set_store(self, out=None, v_tot=None, t0=None, coeff_off=0, n_spl=0, degree=0, knots=None, n_der=0, n_samp=0,
                  dt=0.0, inv_T=1.0):
        """Every following solve writes the trajectories of its solutions inside the solve kernel (out=None: off)."""
        if out is None:
            self._store = None
            _check(self.lib, self.lib.omgx_batch_set_store(self._h, None), 'omgx_batch_set_store')
            return
        self._store = self._store_spec(out, v_tot, t0, coeff_off, n_spl, degree, knots, n_der, n_samp, dt, inv_T)
        _check(self.lib, self.lib.omgx_batch_set_store(self._h, C.byref(self._store)), 'omgx_batch_set_store')

    def sample(self, x, coeff_off, n_spl, degree, knots, n_der, t0, dt, n_samp,
               out=None, as_f32=False, device=False):
        knots = np.ascontiguousarray(knots, dtype=np.float64)

        def ptr(a):
            return a.data_ptr() if hasattr(a, 'data_ptr') else a.ctypes.data
        if out is None:
            out = np.empty((self.n_agents, n_der, n_spl, n_samp),
                           dtype=np.float32 if as_f32 else np.float64)
        _check(self.lib, self.lib.omgx_batch_sample(
            self._h, ptr(x), coeff_off, n_spl, degree, knots.ctypes.data, len(knots), n_der,
            ptr(t0), float(dt), n_samp, ptr(out), int(as_f32), PTR_DEVICE if device else 0),
            'omgx_batch_sample')
        return out

    def shift(self, x, mask, entries, tmats, device=False):
        def ptr(a):
            return a.data_ptr() if hasattr(a, 'data_ptr') else a.ctypes.data
        entries = np.ascontiguousarray(entries, dtype=np.int32)
        tmats = np.ascontiguousarray(tmats, dtype=np.float64)
        _check(self.lib, self.lib.omgx_batch_shift(
            self._h, ptr(x), ptr(mask), entries.ctypes.data, len(entries), tmats.ctypes.data,
            tmats.size, PTR_DEVICE if device else 0), 'omgx_batch_shift')


ADMM_TABLE_MAX_KEYS = 4096


def admm_table_keys(knot_time, step, max_keys=ADMM_TABLE_MAX_KEYS):
    """Times since the last knot at which an update can happen: the multiples of `step` modulo knot_time over one full
    period (until a multiple lands on a knot again), rounded to 1e-6 -- the rounding `omg::ADMMPoint2Point::table`
    applies to its clock.  The reference's generated updz takes the time as a continuous input (`export/export_admm.py`);
    a table needs the two times to be commensurable and raises otherwise."""
    keys, k = [0.0], 1
    while True:
        t = (k * step) % knot_time
        if knot_time - t < 5e-7 or t < 5e-7:
            break                                           # back on a knot: the period is complete
        keys.append(round(t, 6))
        k += 1
        if len(keys) > max_keys:
            raise ValueError('z-update tables: no multiple of the step %g lands on a knot (knot_time %g) within %d '
                             'updates -- update_time / sample_time and knot_time must be commensurable' % (step, knot_time, max_keys))
    return sorted(set(keys))


def save_admm_tables(path, layout, horizon_time, knot_time, update_time, sample_time=None):
    """(sample_time: give it when updates may be shifted by whole samples -- `update1(..., predict_shift)` of the export
    classes advances the clock by predict_shift * sample_time -- the tables then cover its multiples.)
    The z-update tables the C++ ADMM classes of `omg-tools_amd/compat` read (OMG_ADMM_TABLES): for every time since the
    last knot an update can happen at (multiples of update_time modulo knot_time), the consensus projector M and the knot
    transform F of `formation.zupdate_matrices` -- what the reference's exporter generates as updz.so / updres.so
    (`export/export_admm.py`).  Layout: "OMGXADM1", int32 {n_all, n_keys}, per key: t_rel, M, F (row-major doubles)."""
    from .consensus import zupdate_matrices
    keys = admm_table_keys(knot_time, update_time if sample_time is None else sample_time)
    with open(path, 'wb') as fp:
        blobs = []
        for t_rel in keys:
            if hasattr(layout, 'zupdate'):
                M, F = layout.zupdate(round(t_rel / horizon_time, 12))
            else:
                M, F = zupdate_matrices(layout.basis, layout.n_dim, layout.n_nghb, round(t_rel / horizon_time, 12))
            blobs.append((t_rel, np.ascontiguousarray(M, dtype=np.float64), np.ascontiguousarray(F, dtype=np.float64)))
        na = blobs[0][1].shape[0]
        fp.write(b'OMGXADM1')
        fp.write(np.array([na, len(blobs)], dtype=np.int32).tobytes())
        for t_rel, M, F in blobs:
            fp.write(np.float64(t_rel).tobytes()); fp.write(M.tobytes()); fp.write(F.tobytes())
    return path


def second_attempt(solve, set_options, opts, enabled, p, x0, lbg, ubg):
    """One solve of the single-agent solver object.  A solve that ends in Infeasible_Problem_Detected -- phase I given up -- is taken
    once more from the same point with `hess_approx` (include/omgx.h: the Hessian without the curvature of the rows, damped by the
    accepted step length; up to IPOPT's default of 3000 iterations): where IPOPT would enter its restoration phase.  Honoured by
    templates on the general kernel instance; the others solve twice to the same end.  The iteration count reported is the sum."""
    res = solve(p, x0, lbg, ubg)
    if not enabled or int(res['status'][0]) != 2 or opts.get('hess_approx'):
        return res
    max_iter = int(opts.get('max_iter', DEFAULT_OPTIONS['max_iter']))
    # (the retry runs under the caller's own iteration limit when one was given; IPOPT's default of 3000 otherwise)
    set_options(**dict(opts, hess_approx=1, max_iter=max_iter if 'max_iter' in opts else max(max_iter, 3000)))
    try:
        again = solve(p, x0, lbg, ubg)
    finally:
        set_options(**dict(opts, hess_approx=0, max_iter=max_iter))      # (set_options merges: the two keys are put back by name)
    if int(again['status'][0]) == 0:
        again['iters'] = again['iters'] + res['iters']
        return again
    return res


class NlpSolver(object):
    """Single-agent solver object with the reference's `nlpsol` call shape."""

    def __init__(self, template, options):
        self.template = template
        self.opts = options_from_problem(options)
        # (options['omgx']['hess_fallback'] = False switches the second attempt off; templates off the general kernel instance
        # ignore `hess_approx`: a second attempt would repeat the first solve to the same end)
        self.fallback = bool(options.get('omgx', {}).get('hess_fallback', True)) and template_is_general(template)
        self.batch = BatchSolver(template, 1, options=self.opts)
        self._stats = {'return_status': 'Not_Solved', 'iter_count': 0}

    def __call__(self, x0=None, p=None, lbg=None, ubg=None, **kwargs):
        res = second_attempt(self.batch.solve, self.batch.set_options, self.opts, self.fallback,
                             np.asarray(p), np.asarray(x0), np.asarray(lbg), np.asarray(ubg))
        self._stats = {'return_status': STATUS_STRINGS[int(res['status'][0])],
                       'iter_count': int(res['iters'][0])}
        return {'x': res['x'][0], 'lam_g': res['lam_g'][0]}

    def stats(self):
        return dict(self._stats)


def create_nlp(template, options, name=''):
    if options.get('verbose', 0) >= 1:
        print('Building nlp ... ', end=' ')
    t0 = time.time()
    solver = NlpSolver(template, options)
    dt = time.time() - t0
    if options.get('verbose', 0) >= 1:
        print('in %5f s' % dt)
    return solver, dt
