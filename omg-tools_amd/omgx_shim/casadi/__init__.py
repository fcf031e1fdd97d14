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
import numpy as np

inf = float('inf')


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
    @staticmethod
    def sym(name, n=1, m=1):
        if isinstance(n, tuple):
            n, m = n
        out = MX((n, m), None, name=name)
        out._fn = lambda env, key=out: env[key]
        out._deps = (out,)
        return out

    @staticmethod
    def const(v):
        a = _arr(v)
        return MX(a.shape, lambda env: a)

    @staticmethod
    def lift(v):
        return v if isinstance(v, MX) else MX.const(v)

    @staticmethod
    def zeros(*shape):
        if len(shape) == 1 and isinstance(shape[0], tuple):
            shape = shape[0]
        n, m = (shape + (1,))[:2]
        return MX.const(np.zeros((n, m)))

    @staticmethod
    def eye(n):
        return MX.const(np.eye(n))

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
        return _val(self._fn(env)).reshape(self.shape)

    # -- helpers ------------------------------------------------------------------
    def _bin(self, other, op, reflected=False):
        if not isinstance(other, (MX, int, float, np.ndarray, np.floating, np.integer, list)):
            return NotImplemented
        other = MX.lift(other)
        a, b = (other, self) if reflected else (self, other)
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
    def __neg__(self): return MX(self.shape, lambda env: -self.eval(env), self._deps)
    def __ge__(self, o): return self._bin(o, lambda a, b: (a >= b) * 1.0)
    def __le__(self, o): return self._bin(o, lambda a, b: (a <= b) * 1.0)
    def __gt__(self, o): return self._bin(o, lambda a, b: (a > b) * 1.0)
    def __lt__(self, o): return self._bin(o, lambda a, b: (a < b) * 1.0)

    @property
    def T(self):
        return MX(self.shape[::-1], lambda env: self.eval(env).T, self._deps)

    def __getitem__(self, idx):
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


SX = MX


def DM(v):
    return MX.const(v)


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
    parts = [MX.lift(a) for a in args]
    if not parts:
        return MX.const(np.zeros((0, 1)))
    n = sum(p.shape[0] for p in parts)
    deps = ()
    for p in parts:
        deps = _merge(deps, p._deps)
    return MX((n, parts[0].shape[1]), lambda env: np.vstack([p.eval(env) for p in parts]), deps)


def horzcat(*args):
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
    return MX(shape, lambda env: x.eval(env).reshape(shape, order='F'), x._deps)


def vec(x):
    return reshape(x, (x.numel(), 1))


def substitute(expr, sym, val):
    if not isinstance(expr, MX):
        return expr
    val = MX.lift(val)
    deps = _merge(tuple(d for d in expr._deps if d is not sym), val._deps)

    def fn(env):
        env2 = dict(env)
        env2[sym] = val.eval(env)
        return expr.eval(env2)
    return MX(expr.shape, fn, deps)


def _unary(f):
    def g(x):
        if isinstance(x, MX):
            return MX(x.shape, lambda env: f(x.eval(env)), x._deps)
        return f(x)
    return g


cos, sin, sqrt, fabs, exp, log = (_unary(f) for f in (np.cos, np.sin, np.sqrt, np.abs, np.exp, np.log))


class Function(object):
    def __init__(self, name, inputs, outputs=None, *a, **k):
        if outputs is None:
            raise NotImplementedError
        self.name_, self.inputs, self.outputs = name, list(inputs), list(outputs)

    def expand(self):
        return self

    def __call__(self, *args):
        env = {}
        for s, a in zip(self.inputs, args):
            key = s.cat if hasattr(s, 'cat') and not isinstance(s, MX) else s
            if isinstance(a, MX):
                raise NotImplementedError('symbolic Function call')
            env[key] = _arr(a).reshape(key.shape, order='F')
        outs = [MX.lift(o).eval(env) for o in self.outputs]
        return outs[0] if len(outs) == 1 else outs

    def call(self, args):
        return [self(*args)]


def nlpsol(name, solver, nlp, opts=None):
    """`nlpsol('solver', 'ipopt', {'x','p','f','g'}, opts)` (`basics/optilayer.py:60`): the solver object of
    the MI355X path (omgx_shim.ShimSolver: polynomial template -> libomgx.so).  The NLP dictionary stays
    accessible as `.nlp` (tests evaluate f and g through it)."""
    from omgx_shim import ShimSolver
    return ShimSolver(nlp, opts or {}, solver)


def external(*a, **k):
    raise NotImplementedError


def jacobian(*a, **k):
    raise NotImplementedError


def solve(*a, **k):
    raise NotImplementedError


class Importer(object):
    def __init__(self, *a, **k):
        raise NotImplementedError


Compiler = Importer
