"""casadi.tools stand-ins: entry / struct / struct_symMX / struct_MX (see
../__init__.py; test infrastructure only)."""
import numpy as np
from . import MX, vertcat, reshape, vec


class entry(object):
    def __init__(self, name, shape=None, struct=None, expr=None, **kw):
        self.name, self.struct, self.expr = name, struct, expr
        if expr is not None:
            e = MX.lift(expr)
            self.shape = e.shape
            self.expr = e
        elif struct is not None:
            self.shape = (struct.size, 1)
        else:
            if shape is None:
                shape = (1, 1)
            if isinstance(shape, int):
                shape = (shape, 1)
            self.shape = tuple(shape) if len(shape) == 2 else (shape[0], 1)


class struct(object):
    def __init__(self, entries):
        self.entries = list(entries)
        self.layout, off = {}, 0
        for e in self.entries:
            n = e.shape[0] * e.shape[1]
            self.layout[e.name] = (off, e.shape, e.struct)
            off += n
        self.size = off

    def __call__(self, init=0.):
        return _Numeric(self, init)

    def flat(self, key):
        if not isinstance(key, tuple):
            key = (key,)
        off, shape, sub = self.layout[key[0]]
        if len(key) == 1:
            return off, shape
        o2, shape2 = sub.flat(key[1:])
        return off + o2, shape2


class _Numeric(object):
    def __init__(self, st, init=0.):
        self.struct = st
        a = init.cat if hasattr(init, 'cat') else init
        a = np.asarray(a.eval({}) if isinstance(a, MX) else a, dtype=float).reshape(-1)
        self.cat = np.full(st.size, a[0]) if a.size == 1 else a.copy()

    def __getitem__(self, key):
        off, shape = self.struct.flat(key)
        return self.cat[off:off + shape[0] * shape[1]].reshape(shape, order='F')

    def __setitem__(self, key, value):
        off, shape = self.struct.flat(key)
        n = shape[0] * shape[1]
        v = np.asarray(value, dtype=float)
        self.cat[off:off + n] = v.reshape(-1)[0] if v.size == 1 else \
            (v.reshape(-1, order='F') if v.shape == tuple(shape) else v.reshape(-1))

    def prefix(self, label):
        return _Prefix(self, label)


class _Prefix(object):
    def __init__(self, num, label):
        self.num, self.label = num, label

    def __getitem__(self, key):
        return self.num[(self.label,) + (key if isinstance(key, tuple) else (key,))]


class struct_symMX(object):
    def __init__(self, st):
        self.struct = st
        self.cat = MX.sym('struct', st.size, 1)
        self.shape = self.cat.shape

    def __getitem__(self, key):
        off, shape = self.struct.flat(key)
        return reshape(self.cat[off:off + shape[0] * shape[1]], shape)

    def __call__(self, init=0.):
        return self.struct(init)


class struct_MX(object):
    def __init__(self, entries):
        self.struct = struct(entries)
        self.cat = vertcat(*[vec(e.expr) for e in entries]) if entries else MX.const(np.zeros((0, 1)))
        self.shape = self.cat.shape

    def __call__(self, init=0.):
        return self.struct(init)


struct_SX = struct_MX
struct_MX_mutable = struct_MX
