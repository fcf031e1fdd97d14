"""casadi.tools stand-ins: entry / struct / struct_symMX / struct_MX (see
../__init__.py; test infrastructure only)."""
import numpy as np
from . import MX, SX, DM, vertcat, reshape, vec, _arr


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
        self.shape = (off, 1)

    def __call__(self, init=0.):
        a = init.cat if hasattr(init, 'cat') and not isinstance(init, MX) else init
        if isinstance(a, MX) and a._deps:
            return _SymView(self, a)               # struct(mx): symbolic vector seen through the struct
        return _Numeric(self, init)

    def flat(self, key):
        if not isinstance(key, tuple):
            key = (key,)
        off, shape, sub = self.layout[key[0]]
        if len(key) == 1:
            return off, shape
        o2, shape2 = sub.flat(key[1:])
        return off + o2, shape2


class _Numeric(struct):
    """A numeric vector indexed through a struct (`struct(0)`, `struct(values)`); `.cat` is the (n x 1)
    column and entries come back as DM, like casadi's DMStruct."""

    def __init__(self, st, init=0.):
        self.struct = st
        self.entries, self.layout, self.size = st.entries, st.layout, st.size
        a = init.cat if hasattr(init, 'cat') else init
        a = np.asarray(a.eval({}) if isinstance(a, MX) else _arr(a), dtype=float).reshape(-1, order='F')
        self._flat = np.full(st.size, a[0]) if a.size == 1 else a.copy()
        self.shape = (st.size, 1)

    @property
    def cat(self):
        return self._flat.reshape(-1, 1).view(DM)          # (a view: writes go through)

    def __call__(self, init=0.):
        return self.struct(init)

    def __getitem__(self, key):
        key = tuple(str(k) for k in key) if isinstance(key, tuple) else str(key)
        off, shape = self.struct.flat(key)
        return self._flat[off:off + shape[0] * shape[1]].reshape(shape, order='F').view(DM)

    def __setitem__(self, key, value):
        key = tuple(str(k) for k in key) if isinstance(key, tuple) else str(key)
        off, shape = self.struct.flat(key)
        n = shape[0] * shape[1]
        v = np.asarray(_arr(value), dtype=float)
        self._flat[off:off + n] = v.reshape(-1)[0] if v.size == 1 else \
            (v.reshape(-1, order='F') if v.shape == tuple(shape) else v.reshape(-1))

    @property
    def prefix(self):
        return _PrefixIndexer(self)


class _SymView(struct):
    """A symbolic vector indexed through a struct (`casadi.tools`: `struct(mx)[...]`, `struct_MX_mutable`);
    entries can be re-assigned."""

    def __init__(self, st, mx):
        self.struct = st
        self.entries, self.layout, self.size = st.entries, st.layout, st.size
        self._vec = MX.lift(mx)
        self.shape = self._vec.shape

    @property
    def cat(self):
        return self._vec

    def __call__(self, init=0.):
        return self.struct(init)

    def __getitem__(self, key):
        key = tuple(str(k) for k in key) if isinstance(key, tuple) else str(key)
        off, shape = self.struct.flat(key)
        return reshape(self._vec[off:off + shape[0] * shape[1]], shape)

    def __setitem__(self, key, value):
        key = tuple(str(k) for k in key) if isinstance(key, tuple) else str(key)
        off, shape = self.struct.flat(key)
        n = shape[0] * shape[1]
        new = MX(self._vec.shape, self._vec._fn, self._vec._deps)      # fresh node: the old vector may be shared
        new[off:off + n] = vec(MX.lift(value))
        self._vec = new

    @property
    def prefix(self):
        return _PrefixIndexer(self)


class _PrefixIndexer(object):
    """`s.prefix(label)` (`basics/optilayer.py:352`) and `s.prefix[label]` (`problems/admm.py:470`)."""

    def __init__(self, num):
        self.num = num

    def __call__(self, label):
        return _Prefix(self.num, str(label))

    __getitem__ = __call__


class _Prefix(object):
    def __init__(self, num, label):
        self.num, self.label = num, label

    def __getitem__(self, key):
        return self.num[(self.label,) + (key if isinstance(key, tuple) else (key,))]

    def cast(self):
        """The whole sub-vector of this prefix as a column."""
        v = self.num[self.label]
        return DM(v) if not isinstance(v, MX) else v


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
    def __new__(cls, entries):
        if isinstance(entries, struct):              # struct_MX_mutable(struct): an all-zero symbolic struct to fill
            return _SymView(entries, MX.zeros(entries.size, 1))
        return object.__new__(cls)

    def __init__(self, entries):
        self.struct = struct(entries)
        self.cat = vertcat(*[vec(e.expr) for e in entries]) if entries else MX.const(np.zeros((0, 1)))
        self.shape = self.cat.shape

    def __call__(self, init=0.):
        return self.struct(init)


struct_SX = struct_MX
struct_MX_mutable = struct_MX
