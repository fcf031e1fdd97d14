"""omgx_shim -- omg-tools' OWN classes on the MI355X solve path.

The reference builds its NLP as CasADi graphs and calls `nlpsol(...)` once per problem
(`basics/optilayer.py:49-60`); every receding-horizon step then calls the returned object
(`problems/problem.py:113`).  CasADi is the only thing between omg-tools and the solver, so the drop-in
sits exactly there: `omgx_shim.install()` puts a stand-in `casadi` module on `sys.path`
(omgx_shim/casadi: lazily evaluated closures), the user's omg-tools install is imported unchanged, and
when it calls `nlpsol` the closures are evaluated once on polynomial values (`omgtools/symbolic.py` of
this repository) -- which yields every row of g and the objective as explicit polynomials, i.e. an
`NLPTemplate` -- and the hand-written HIP solver (libomgx.so, include/omgx.h) is created behind the
reference's call shape.  No re-typed copy of the front end is involved.

    import omgx_shim
    omgx_shim.install()             # before the first `import omgtools`
    from omgtools import *          # the reference package, unmodified
    ...                             # examples/p2p_holonomic.py verbatim

Scope: the rows the reference's in-scope classes define are polynomial of degree <= 3 in the variables
with parameter-polynomial coefficients; the one non-polynomial construct on the path, `evalspline` with
a symbolic argument (`basics/spline_extra.py:28-55`: Cox-de Boor on indicator functions of t/T), is
replaced at import time by B-spline basis atoms of the template (same values, evaluated per agent on the
device).  Anything else raises when the template is built -- never a silent fallback.
"""
import importlib
import importlib.abc
import importlib.util
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_NATIVE = None
solver_factory = None      # tests inject a host solver here (callable(template, options) -> nlpsol-shaped object)


def native():
    """This repository's modules (symbolic, template, backend ...) under the package name `omgx_native`:
    the name `omgtools` belongs to the reference install in a shim process."""
    global _NATIVE
    if _NATIVE is None:
        path = os.path.join(os.path.dirname(_HERE), 'omgtools')
        if 'omgtools' in sys.modules and os.path.dirname(getattr(sys.modules['omgtools'], '__file__', '') or '') == path:
            _NATIVE = sys.modules['omgtools']            # (running inside this repository's own package)
            return _NATIVE

        class _Pkg(object):
            pass
        spec = importlib.util.spec_from_file_location('omgx_native', os.path.join(path, '__init__.py'),
                                                      submodule_search_locations=[path])
        mod = importlib.util.module_from_spec(spec)
        sys.modules['omgx_native'] = mod
        # only the modules the shim needs: not the package's __init__ (it mirrors the reference's API)
        mod.__path__ = [path]
        for name in ('symbolic', 'splines', 'template', 'backend'):
            importlib.import_module('omgx_native.' + name)
        _NATIVE = mod
    return _NATIVE


def _mod(name):
    nat = native()
    return importlib.import_module(nat.__name__ + '.' + name)


# ---------------------------------------------------------------------------------------------------
# install: the stand-in casadi + the evalspline replacement
# ---------------------------------------------------------------------------------------------------
class _PatchSplineExtra(importlib.abc.MetaPathFinder):
    """Replaces `evalspline` of the reference's `basics/spline_extra.py` right after that module is
    executed, i.e. before any other module of the reference imports the name."""
    target = 'omgtools.basics.spline_extra'

    def find_spec(self, fullname, path, target=None):
        if fullname != self.target:
            return None
        for finder in sys.meta_path:
            if finder is self or not hasattr(finder, 'find_spec'):
                continue
            spec = finder.find_spec(fullname, path, target)
            if spec is not None and spec.loader is not None:
                inner = spec.loader

                class Loader(importlib.abc.Loader):
                    def create_module(self_, sp):
                        return inner.create_module(sp)

                    def exec_module(self_, module):
                        inner.exec_module(module)
                        patch_spline_extra(module)
                spec.loader = Loader()
                return spec
        return None


def install():
    """Put the stand-in `casadi` on sys.path and arm the evalspline patch.  Call before `import omgtools`."""
    if _HERE not in sys.path:
        sys.path.insert(0, _HERE)
    if 'casadi' in sys.modules and not getattr(sys.modules['casadi'], '__file__', '').startswith(_HERE):
        raise RuntimeError('casadi is already imported: call omgx_shim.install() first')
    if not any(isinstance(f, _PatchSplineExtra) for f in sys.meta_path):
        sys.meta_path.insert(0, _PatchSplineExtra())
    if 'omgtools.basics.spline_extra' in sys.modules:
        patch_spline_extra(sys.modules['omgtools.basics.spline_extra'])


def patch_spline_extra(module):
    """`evalspline(s, x)` with a symbolic x: sum_i coeffs_i * B_i(x) with the basis values as atoms of the
    template when the closure is evaluated on polynomials, numerically (the reference's span convention,
    `basics/spline.py:131-136`) when it is evaluated on numbers."""
    import casadi
    original = module.evalspline
    if getattr(original, '_omgx_patched', False):
        return

    def evalspline(s, x):
        if not isinstance(x, casadi.MX):
            return original(s, x)
        basis, coeffs = s.basis, s.coeffs
        knots, degree = np.asarray(basis.knots, float), int(basis.degree)
        cm = coeffs if isinstance(coeffs, casadi.MX) else casadi.MX.const(np.asarray(coeffs, float))
        deps = casadi._merge(x._deps, cm._deps)

        def fn(env):
            xv = x.eval(env).reshape(-1)[0]
            cv = cm.eval(env).reshape(-1)
            sym = _mod('symbolic')
            if isinstance(xv, sym.Poly) and not xv.is_param_only():
                # The evaluation point involves VARIABLES: t / T with a free end time T (`problems/point2point.py:269-369`: T is
                # the variable, t the parameter -- 0 throughout a free-T run, the time axis resets with every update -- so the
                # point is the lifted quotient q = t / T, symbolic.py).  Basis functions at a variable point are not atoms; on the
                # FIRST knot span, where q lives, every one of them is a polynomial of the spline's degree in q: the value is
                # Horner's scheme in q over combinations of the leading coefficients (products beyond four factors are lifted).
                # Valid only while q stays on the first span [0, first interior knot): the series is in powers of (u - knots[0]) and
                # Horner below runs in q itself, so the basis must start at 0; the one use in the reference is q = t / T with the
                # parameter t = 0 (free end time) -- a numerator that is not a bare parameter would be a point somewhere else on
                # the horizon, where this series is the wrong polynomial: refused rather than built silently.
                if knots[0] != 0.0:
                    raise NotImplementedError('evalspline at a variable point: the basis must start at 0 (knots[0] = %g)' % knots[0])
                only = list(xv.terms.items())
                q_sym = only[0][0][0][0] if len(only) == 1 and only[0][1] == 1.0 and len(only[0][0][0]) == 1 and not only[0][0][1] else None
                num = sym.SymbolTable.current().quotient_num.get(q_sym)
                if num is None or not num.is_param_only():
                    raise NotImplementedError('evalspline at a variable point: only parameter / variable (t / T with t inside the first '
                                              'knot span) is supported')
                M = _mod('splines').first_span_power_series(knots, degree)          # B_i(u) = sum_k M[i, k] u^k for 0 <= u < first knot
                C = []
                for k in range(degree + 1):
                    ck = 0.0
                    for i in range(degree + 1):
                        if M[i, k] != 0.0:
                            ck = ck + cv[i] * float(M[i, k])
                    C.append(ck)
                acc = C[degree]
                for k in range(degree - 1, -1, -1):
                    acc = acc * xv + C[k]
                out = np.empty((1, 1), dtype=object)
                out[0, 0] = acc
                return out
            if isinstance(xv, sym.Poly):
                table = sym.SymbolTable.current()
                B = [sym.Poly.symbol(i) for i in table.new_bspl_atoms(knots, degree, xv)]
            else:
                B = _mod('splines').BSplineBasis(knots, degree).eval_basis([float(xv)])[0]
            acc = 0.0
            for l in range(len(cv)):
                acc = acc + cv[l] * B[l]
            out = np.empty((1, 1), dtype=object)
            out[0, 0] = acc
            return out
        return casadi.MX((1, 1), fn, deps)
    evalspline._omgx_patched = True
    module.evalspline = evalspline


# ---------------------------------------------------------------------------------------------------
# nlpsol: CasADi graphs -> polynomial template -> HIP solver
# ---------------------------------------------------------------------------------------------------
def _layout(st, prefix=()):
    """{(label, name): (offset, rows, cols)} of a casadi.tools struct (nested one level, like the reference's
    `_var_struct` / `_par_struct` / `_con_struct`, `basics/optilayer.py:225-272`)."""
    out = {}
    for e in st.entries:
        if e.struct is not None:
            for e2 in e.struct.entries:
                off, shape = st.flat((e.name, e2.name))
                out[(e.name, e2.name)] = (off, shape[0], shape[1])
        else:
            off, shape = st.flat(e.name)
            out[(e.name, e.name)] = (off, shape[0], shape[1])
    return out


def template_from_nlp(nlp, lbg, ubg):
    """Evaluate the reference's f and g closures on polynomial values: x -> variable symbols, p -> raw
    atoms; quotients of parameters become DIV atoms, `evalspline` at t/T BSPL atoms."""
    sym, tmpl = _mod('symbolic'), _mod('template')
    X, Pm = nlp['x'], nlp['p']
    xs, ps = (X.cat if hasattr(X, 'cat') else X), (Pm.cat if hasattr(Pm, 'cat') else Pm)
    f, g = nlp['f'], nlp['g']
    gs = g.cat if hasattr(g, 'cat') else g
    n, npar = xs.shape[0] * xs.shape[1], ps.shape[0] * ps.shape[1]
    table = sym.SymbolTable()
    var_syms, par_syms = table.new_vars(n), table.new_raw_atoms(npar)
    env = {xs: np.array([sym.Poly.symbol(s) for s in var_syms], dtype=object).reshape(xs.shape, order='F'),
           ps: np.array([sym.Poly.symbol(s) for s in par_syms], dtype=object).reshape(ps.shape, order='F')}
    with table:
        rows = [sym.Poly.lift(v) for v in np.asarray(gs.eval(env), dtype=object).reshape(-1, order='F')]
        import casadi
        objective = sym.Poly.lift(np.asarray(casadi.MX.lift(f).eval(env), dtype=object).reshape(-1)[0])
    kw = {}
    if hasattr(X, 'struct'):
        kw['var_layout'] = _layout(X.struct)
    if hasattr(Pm, 'struct'):
        kw['par_layout'] = _layout(Pm.struct)
    if hasattr(g, 'struct'):
        kw['con_layout'] = _layout(g.struct)
    return tmpl.NLPTemplate.from_polys(table, var_syms, par_syms, rows, objective, lbg, ubg, **kw)


class ShimSolver(object):
    """What `nlpsol` returns: `solver(x0=, p=, lbg=, ubg=) -> {'x', 'lam_g'}`, `solver.stats()`
    (`problems/problem.py:113-128`).  The template (and with it the device batch, B = 1) is built at the
    first call, when the bounds -- which rows are equalities -- are known."""

    def __init__(self, nlp, opts, plugin='ipopt'):
        self.nlp, self.opts, self.plugin = nlp, dict(opts), plugin
        self.template, self._impl = None, None
        self._stats = {'return_status': 'Not_Solved', 'iter_count': 0}

    def _options(self):
        kw = {}
        for key, name in (('ipopt.tol', 'tol'), ('ipopt.max_iter', 'max_iter')):
            if key in self.opts:
                kw[name] = type(_mod('backend').DEFAULT_OPTIONS[name])(self.opts[key])
        return kw

    def _vec(self, v):
        import casadi
        return np.asarray(casadi._arr(v), dtype=float).reshape(-1, order='F')

    def __call__(self, x0=None, p=None, lbg=None, ubg=None, **kwargs):
        x0, p, lbg, ubg = self._vec(x0), self._vec(p), self._vec(lbg), self._vec(ubg)
        if self._impl is None:
            import casadi
            self.template = casadi.deep(template_from_nlp, self.nlp, lbg, ubg)
            options = {'solver': 'ipopt', 'solver_options': {'ipopt': {k: v for k, v in self.opts.items() if k.startswith('ipopt.')}},
                       'verbose': 0}
            if solver_factory is not None:
                self._impl = solver_factory(self.template, options)
            else:
                self._impl = _mod('backend').NlpSolver(self.template, options)     # raises without libomgx.so / a HIP device
        # (a template with lifted products -- symbolic.py, `Poly.__mul__` -- carries auxiliary variables and rows behind the caller's:
        # they are filled in from their defining rows here and cut off again below; a template without them passes through)
        x0f, lbf, ubf = self.template.lift_extend(x0, p, lbg, ubg)
        res = self._impl(x0=x0f, p=p, lbg=lbf, ubg=ubf)
        self._stats = self._impl.stats()
        x, lam = self.template.lift_strip(np.asarray(res['x']).reshape(-1), np.asarray(res['lam_g']).reshape(-1))
        return {'x': np.asarray(x), 'lam_g': np.asarray(lam)}

    def stats(self):
        return dict(self._stats)
