"""Stand-in for the parts of the CasADi Python API that omg-tools touches (omgx_shim: the reference's own
classes on the MI355X solve path, see ../__init__.py).

omg-tools builds its NLP as CasADi graphs (`basics/optilayer.py:556-669`) and hands them to
`nlpsol('solver', 'ipopt', {x, p, f, g}, opts)` (`optilayer.py:49-60`).  Here every MX is a lazily
evaluated closure over its leaf symbols.  Evaluated on numbers it gives f(x, p) and g(x, p) at a point
(tests/golden uses that to pin the product's own front end to the reference's construct code).
Evaluated on `Poly` values (omgtools/symbolic.py) it gives every row of g and f as an explicit
polynomial in the variables with parameter-polynomial coefficients -- all the HIP solver needs
(template.NLPTemplate.from_polys): `nlpsol` below does exactly that and returns a solver object with
the call shape the reference uses (`problems/problem.py:113-119`).  No AD, no graph optimisation:
the in-scope rows are polynomial (degree <= 3 in x), their derivatives are taken term by term on the
device.
"""
import sys

import numpy as np

inf = float('inf')
_CACHE = '__omgx_cache__'


def deep(fn, *args):
    """Run fn(*args) where recursion may go deep: the graphs are evaluated recursively and the reference builds
    long chains (obj += ..., matrices filled entry by entry), so the evaluation gets its own thread with a
    large stack and a matching recursion limit."""
    import threading
    if getattr(deep, 'inside', False):
        return fn(*args)
    box = {}

    def run():
        deep.inside = True
        old = sys.getrecursionlimit()
        sys.setrecursionlimit(200000)
        try:
            box['out'] = fn(*args)
        except BaseException as e:          # noqa: BLE001  (re-raised in the caller's thread)
            box['err'] = e
        finally:
            sys.setrecursionlimit(old)
            deep.inside = False
    old_size = threading.stack_size(1 << 29)
    try:
        th = threading.Thread(target=run)
        th.start()
        th.join()
    finally:
        threading.stack_size(old_size)
    if 'err' in box:
        raise box['err']
    return box['out']


def _fork(env):
    """Copy of an evaluation environment for a substitution (the memo of the original does not apply)."""
    e2 = dict(env)
    e2.pop(_CACHE, None)
    return e2


def _val(v):
    """Evaluation result as an array: float if it can be, object (Poly entries) otherwise."""
    a = np.asarray(v)
    if a.dtype == object:
        try:
            return a.astype(float)
        except (TypeError, ValueError):
            return a
    return a.astype(float)


def _arr(v):
    if isinstance(v, list) and len(v) == 0:
        return np.zeros((0, 1))
    if hasattr(v, 'toarray'):
        v = v.toarray()
    if hasattr(v, 'cat') and not isinstance(v, MX):
        v = v.cat
    a = _val(v)
    if a.ndim == 0:
        a = a.reshape(1, 1)
    elif a.ndim == 1:
        a = a.reshape(-1, 1)
    return a


def _pow(a, b):
    """a ** b for float or Poly-valued arrays (integral exponents for polynomials)."""
    if np.asarray(a).dtype != object:
        return np.power(a, b)
    b = np.broadcast_to(np.asarray(b, dtype=float), np.broadcast(a, b).shape)
    a = np.broadcast_to(a, b.shape)
    out = np.empty(b.shape, dtype=object)
    for i in np.ndindex(*b.shape):
        e = float(b[i])
        if e != int(e) or e < 0:
            raise TypeError('non-integral power of a polynomial')
        out[i] = a[i] ** int(e)
    return out


class MX(object):
    __array_ufunc__ = None
    __hash__ = object.__hash__

    def __init__(self, shape, fn, deps=(), name=None):
        self.shape = (int(shape[0]), int(shape[1]))
        self._fn = fn
        self._deps = tuple(deps)
        self._name = name

    # -- construction -------------------------------------------------------
    @classmethod
    def sym(cls, name, n=1, m=1):
        if isinstance(n, tuple):
            n, m = n
        out = cls((n, m), None, name=name)
        out._fn = lambda env, key=out: env[key]
        out._deps = (out,)
        return out

    @staticmethod
    def const(v):
        a = _arr(v)
        return MX(a.shape, lambda env: a)

    @staticmethod
    def lift(v):
        if isinstance(v, MX):
            # a matrix that is being filled entry by entry (__setitem__) is captured as it is NOW: a later
            # assignment to it must not change -- or recurse into -- expressions built from it before
            # (also one that is assigned to only later: `m[sl] = f(m[sl])`, `problems/dualmethod.py:174-176`).
            # Leaf symbols are identities (keys of the evaluation environment) and never assigned to.
            if v._name is not None and len(v._deps) == 1 and v._deps[0] is v:
                return v
            return MX(v.shape, v._fn, v._deps)
        if hasattr(v, 'cat') and isinstance(v.cat, MX):      # struct views
            return v.cat
        return MX.const(v)

    @classmethod
    def zeros(cls, *shape):
        if len(shape) == 1 and isinstance(shape[0], tuple):
            shape = shape[0]
        n, m = (tuple(shape) + (1,))[:2]
        a = np.zeros((n, m))
        return cls((n, m), lambda env: a)

    @classmethod
    def eye(cls, n):
        a = np.eye(n)
        return cls((n, n), lambda env: a)

    # -- queries --------------------------------------------------------------
    def name(self):
        return self._name

    def size(self, *a):
        return self.shape if not a else self.shape[a[0] - 1]

    def size1(self):
        return self.shape[0]

    def size2(self):
        return self.shape[1]

    def numel(self):
        return self.shape[0] * self.shape[1]

    def __len__(self):
        return self.shape[0]

    def eval(self, env):
        cache = env.get(_CACHE)
        if cache is None:
            cache = env[_CACHE] = {}
        key = id(self._fn)                      # (snapshots of one node share its closure, hence its memo)
        hit = cache.get(key)
        if hit is not None and hit[0] is self._fn:
            return hit[1].reshape(self.shape)
        out = _val(self._fn(env)).reshape(self.shape)
        cache[key] = (self._fn, out)
        return out

    # -- helpers ------------------------------------------------------------------
    def _bin(self, other, op, reflected=False):
        if isinstance(other, DM):
            other = np.asarray(other)
        if not isinstance(other, (MX, int, float, np.ndarray, np.floating, np.integer, list)):
            return NotImplemented
        other = MX.lift(other)
        me = MX.lift(self)
        a, b = (other, me) if reflected else (me, other)
        shape = a.shape if a.numel() >= b.numel() else b.shape
        deps = _merge(a._deps, b._deps)
        return MX(shape, lambda env: op(a.eval(env), b.eval(env)) + np.zeros(shape), deps)

    def __add__(self, o): return self._bin(o, np.add)
    def __radd__(self, o): return self._bin(o, np.add, True)
    def __sub__(self, o): return self._bin(o, np.subtract)
    def __rsub__(self, o): return self._bin(o, np.subtract, True)
    def __mul__(self, o): return self._bin(o, np.multiply)
    def __rmul__(self, o): return self._bin(o, np.multiply, True)
    def __truediv__(self, o): return self._bin(o, np.divide)
    def __rtruediv__(self, o): return self._bin(o, np.divide, True)
    def __pow__(self, o): return self._bin(o, _pow)
    def __neg__(self):
        me = MX.lift(self)
        return MX(me.shape, lambda env: -me.eval(env), me._deps)
    def __ge__(self, o): return self._bin(o, lambda a, b: (a >= b) * 1.0)
    def __le__(self, o): return self._bin(o, lambda a, b: (a <= b) * 1.0)
    def __gt__(self, o): return self._bin(o, lambda a, b: (a > b) * 1.0)
    def __lt__(self, o): return self._bin(o, lambda a, b: (a < b) * 1.0)

    @property
    def T(self):
        me = MX.lift(self)
        return MX(me.shape[::-1], lambda env: me.eval(env).T, me._deps)

    def __getitem__(self, idx):
        self = MX.lift(self)
        probe = np.zeros(self.shape)
        if not isinstance(idx, tuple):
            if self.shape[1] == 1 or isinstance(idx, (list, np.ndarray)):
                idx = (idx, slice(None))
            else:                      # linear (column-major) indexing
                lin = idx
                sub = np.zeros(self.numel()).reshape(-1, 1)[lin]
                shape = _arr(sub).shape
                return MX(shape, lambda env: _arr(self.eval(env).reshape(-1, order='F')[lin]), self._deps)
        sub = _arr(probe[idx])
        if isinstance(idx[0], (int, np.integer)) and not isinstance(idx[1], (int, np.integer)):
            sub = sub.reshape(1, -1)
        shape = sub.shape
        return MX(shape, lambda env: _val(self.eval(env)[idx]).reshape(shape), self._deps)


    def __setitem__(self, idx, value):
        """In-place element / slice assignment (`basics/spline_extra.py:220-255` fills its transformation
        matrices entry by entry): the closure is re-bound to "old value with these entries replaced"."""
        old_fn, old_deps = self._fn, self._deps
        val = MX.lift(value)
        shape = self.shape
        self._mutable = True

        def fn(env, old_fn=old_fn):
            a = np.array(_val(old_fn(env)).reshape(shape))
            v = val.eval(env)
            if v.dtype == object and a.dtype != object:
                a = a.astype(object)
            tgt = a[idx]
            a[idx] = v.reshape(np.shape(tgt)) if np.ndim(tgt) else v.reshape(-1)[0]
            return a
        self._fn = fn
        self._deps = _merge(tuple(d for d in old_deps if d is not self), val._deps)


class SX(MX):
    """(the reference tells SX and MX apart with isinstance, `problems/dualmethod.py:104-127`)"""


class DM(np.ndarray):
    """Numeric matrix (`casadi.DM`): a 2-D numpy array under that name, so that numpy arithmetic keeps the
    type (the reference tells DM, MX and SX apart with isinstance, `problems/dualmethod.py:104-127`)."""
    __array_priority__ = 20.0

    def __new__(cls, v=0.):
        return _arr(v).astype(float).view(cls)

    def toarray(self):
        return np.asarray(self)

    full = toarray

    def size1(self):
        return self.shape[0]

    def size2(self):
        return self.shape[1] if self.ndim > 1 else 1

    def __float__(self):
        return float(np.asarray(self).reshape(-1)[0])


def _merge(a, b):
    out = list(a)
    for d in b:
        if not any(d is e for e in out):
            out.append(d)
    return tuple(out)


def symvar(expr):
    return list(expr._deps) if isinstance(expr, MX) else []


def mtimes(a, b, *more):
    a, b = MX.lift(a), MX.lift(b)
    if a.numel() == 1 or b.numel() == 1:
        out = a * b
    else:
        out = MX((a.shape[0], b.shape[1]), lambda env: a.eval(env) @ b.eval(env), _merge(a._deps, b._deps))
    for m in more:
        out = mtimes(out, m)
    return out


def vertcat(*args):
    args = [a for a in args if not (isinstance(a, list) and len(a) == 0)]
    if args and not any(isinstance(a, MX) or hasattr(a, 'cat') for a in args):
        return DM(np.vstack([_arr(a) for a in args]))
    parts = [MX.lift(a) for a in args]
    if not parts:
        return MX.const(np.zeros((0, 1)))
    n = sum(p.shape[0] for p in parts)
    deps = ()
    for p in parts:
        deps = _merge(deps, p._deps)
    return MX((n, parts[0].shape[1]), lambda env: np.vstack([p.eval(env) for p in parts]), deps)


def horzcat(*args):
    args = [a for a in args if not (isinstance(a, list) and len(a) == 0)]
    parts = [MX.lift(a) for a in args]
    m = sum(p.shape[1] for p in parts)
    deps = ()
    for p in parts:
        deps = _merge(deps, p._deps)
    return MX((parts[0].shape[0], m), lambda env: np.hstack([p.eval(env) for p in parts]), deps)


def vertsplit(x):
    return [x[i] for i in range(x.shape[0])]


def reshape(x, *shape):
    if len(shape) == 1:
        shape = shape[0]
    x = MX.lift(x)
    return MX(shape, lambda env: x.eval(env).reshape(shape, order='F'), x._deps)


def vec(x):
    return reshape(x, (x.numel(), 1))


def substitute(expr, sym, val):
    if not isinstance(expr, MX):
        return expr
    expr, val = MX.lift(expr), MX.lift(val)
    deps = _merge(tuple(d for d in expr._deps if d is not sym), val._deps)

    def fn(env):
        env2 = _fork(env)
        env2[sym] = val.eval(env)
        return expr.eval(env2)
    return MX(expr.shape, fn, deps)


def _unary(f):
    def g(x):
        if isinstance(x, MX):
            x = MX.lift(x)
            return MX(x.shape, lambda env: f(x.eval(env)), x._deps)
        return f(x)
    return g


cos, sin, sqrt, fabs, exp, log = (_unary(f) for f in (np.cos, np.sin, np.sqrt, np.abs, np.exp, np.log))


class Function(object):
    def __init__(self, name, inputs, outputs=None, *a, **k):
        if outputs is None:
            raise NotImplementedError
        self.name_, self.inputs = name, list(inputs)
        self.outputs = [o.cat if hasattr(o, 'cat') and not isinstance(o, MX) else o for o in outputs]

    def expand(self):
        return self

    def __call__(self, *args):
        keys = [s.cat if hasattr(s, 'cat') and not isinstance(s, MX) else s for s in self.inputs]
        args = [a.cat if hasattr(a, 'cat') and not isinstance(a, MX) else a for a in args]
        if any(isinstance(a, MX) and a._deps for a in args):
            # symbolic call: the outputs with the inputs replaced by the argument expressions
            margs = [MX.lift(a) for a in args]
            deps = ()
            for a in margs:
                deps = _merge(deps, a._deps)

            def make(o):
                o = MX.lift(o)

                def fn(env):
                    env2 = _fork(env)
                    for key, a in zip(keys, margs):
                        env2[key] = a.eval(env).reshape(key.shape, order='F')
                    return o.eval(env2)
                return MX(o.shape, fn, deps)
            outs = [make(o) for o in self.outputs]
            return outs[0] if len(outs) == 1 else outs
        env = {}
        for key, a in zip(keys, args):
            if isinstance(a, MX):
                a = a.eval({})                              # DM(...) is a constant closure here
            env[key] = _arr(a).reshape(key.shape, order='F')
        outs = deep(lambda: [DM(MX.lift(o).eval(env)) for o in self.outputs])
        return outs[0] if len(outs) == 1 else outs

    def call(self, args):
        return [self(*args)]

    def sparsity_jac(self, iind=0, oind=0):
        return deep(self._sparsity_jac, iind, oind)

    def _sparsity_jac(self, iind=0, oind=0):
        """Sparsity of d output[oind] / d input[iind] (`problems/distributedproblem.py:26-33` reads
        `J.T.row()`: the input elements the output depends on).  Found by perturbing one input element at
        a time at a random point."""
        rng = np.random.default_rng(12345)
        base = {}
        for s_ in self.inputs:
            key = s_.cat if hasattr(s_, 'cat') and not isinstance(s_, MX) else s_
            base[key] = 0.3 + 0.4 * rng.random(key.shape)
        out = MX.lift(self.outputs[oind])
        y0 = np.asarray(out.eval(base), float).reshape(-1, order='F')
        key = self.inputs[iind]
        key = key.cat if hasattr(key, 'cat') and not isinstance(key, MX) else key
        n_in = key.shape[0] * key.shape[1]
        pat = np.zeros((y0.size, n_in), dtype=bool)
        for e in range(n_in):
            env = _fork(base)
            v = base[key].copy().reshape(-1, order='F')
            v[e] += 0.137
            env[key] = v.reshape(key.shape, order='F')
            y = np.asarray(out.eval(env), float).reshape(-1, order='F')
            pat[:, e] = np.abs(y - y0) > 1e-14 * (1 + np.abs(y0))
        return _Sparsity(pat)


class _Sparsity(object):
    def __init__(self, pattern):
        self.pattern = np.asarray(pattern, dtype=bool)

    @property
    def T(self):
        return _Sparsity(self.pattern.T)

    def row(self):
        """Row index of every structural non-zero (column-major order, like casadi.Sparsity.row)."""
        return [int(r) for c in range(self.pattern.shape[1]) for r in np.nonzero(self.pattern[:, c])[0]]

    def size1(self):
        return self.pattern.shape[0]

    def size2(self):
        return self.pattern.shape[1]


def nlpsol(name, solver, nlp, opts=None):
    """`nlpsol('solver', 'ipopt', {'x','p','f','g'}, opts)` (`basics/optilayer.py:60`): the solver object of
    the MI355X path (omgx_shim.ShimSolver: polynomial template -> libomgx.so).  The NLP dictionary stays
    accessible as `.nlp` (tests evaluate f and g through it)."""
    from omgx_shim import ShimSolver
    return ShimSolver(nlp, opts or {}, solver)


def external(*a, **k):
    raise NotImplementedError


def _dep_prune(expr, candidates, probe_keys):
    """Leaf symbols among `candidates` the closure really depends on (numerically: the value changes when the
    symbol is perturbed at a random point)."""
    rng = np.random.default_rng(999)
    base = {k: 0.2 + 0.5 * rng.random(k.shape) for k in probe_keys}
    y0 = np.asarray(expr.eval(base), float)
    keep = []
    for s_ in candidates:
        env = _fork(base)
        env[s_] = base[s_] + 0.173 + 0.1 * rng.random(s_.shape)
        if np.abs(np.asarray(expr.eval(env), float) - y0).max() > 1e-13 * (1 + np.abs(y0).max()):
            keep.append(s_)
    return tuple(keep)


def jacobian(expr, var):
    """d expr / d var by differences with step 1 -- exact for the affine coupling constraints it is used on
    (`problems/admm.py:313-354` builds the z-update's A from it and rejects anything else: the symbols the
    result still depends on are pruned numerically, so a non-affine constraint shows up there)."""
    expr, var = MX.lift(expr), MX.lift(var)
    n_out, n_in = expr.numel(), var.numel()

    def fn(env):
        y0 = np.asarray(expr.eval(env), float).reshape(-1, order='F')
        J = np.zeros((n_out, n_in))
        v0 = np.asarray(env[var], float).reshape(-1, order='F') if var in env else np.zeros(n_in)
        for e in range(n_in):
            env2 = _fork(env)
            v = v0.copy()
            v[e] += 1.0
            env2[var] = v.reshape(var.shape, order='F')
            J[:, e] = np.asarray(expr.eval(env2), float).reshape(-1, order='F') - y0
        return J
    out = MX((n_out, n_in), fn, expr._deps)
    # an entry of the Jacobian of an affine expression does not depend on the variables any more
    probe = _merge(expr._deps, (var,))

    def fn_safe(env):
        env2 = _fork(env)
        for k in probe:
            env2.setdefault(k, np.zeros(k.shape))
        return fn(env2)
    out._fn = fn_safe
    out._deps = deep(_dep_prune, MX((n_out, n_in), fn_safe, ()), expr._deps, probe)
    return out


def solve(A, b, *a, **k):
    A, b = MX.lift(A), MX.lift(b)
    return MX((A.shape[1], b.shape[1]), lambda env: np.linalg.solve(np.asarray(A.eval(env), float), np.asarray(b.eval(env), float)),
              _merge(A._deps, b._deps))


class Importer(object):
    def __init__(self, *a, **k):
        raise NotImplementedError


Compiler = Importer
