"""Sparse polynomial coefficients: the numeric stand-in for CasADi MX/SX.

The reference builds every constraint as a CasADi expression graph
(`basics/optilayer.py:556-669`, `_define_mx` 610-617) and lets CasADi derive
Jacobians/Hessians by AD.  All constraints on the hot path are *polynomial* in
the decision variables (degree <= 3) with coefficients that
are polynomials in the parameters and a few derived parameter quantities
(t/T, basis functions evaluated at t/T).  So instead of a graph we carry each
spline coefficient as an explicit sparse polynomial

    sum_k  c_k * prod(atoms in A_k) * prod(variables in V_k)

`Poly` implements exactly that ring.  Symbols are plain ints handed out by a
`SymbolTable`; variable symbols and parameter ("atom") symbols live in
disjoint id ranges so a monomial key is `(vars_tuple, atoms_tuple)`.

Derived atoms (division of parameter polynomials, B-spline basis functions
evaluated at a parameter-dependent point — the numeric twin of `evalspline`
with a symbolic argument, `basics/spline_extra.py:28-55`) are recorded as a
small straight-line program that the device evaluates per agent
(csrc/omgx_kernels.hip, `eval_atoms`).
"""
import numpy as np

VAR, ATOM = 0, 1
_ATOM_BASE = 1 << 30          # ids >= _ATOM_BASE are atoms, below are variables
LIFT_CAP = 4                  # variable factors a template term carries (include/omgx.h OMGX_TERM_VARS); products beyond it are lifted


def is_atom(sym):
    return sym >= _ATOM_BASE


def _exact_key(v):
    """Hashable identity of a Poly (every term with its exact coefficient) or of a number."""
    terms = getattr(v, 'terms', None)
    return tuple(sorted(terms.items())) if terms is not None else ('num', float(v))


class Poly(object):
    """Sparse multivariate polynomial with float coefficients."""
    __slots__ = ('terms',)
    __array_ufunc__ = None      # make numpy defer to our reflected operators

    def __init__(self, terms=None):
        self.terms = terms if terms is not None else {}

    # -- constructors ------------------------------------------------------
    @staticmethod
    def const(value):
        value = float(value)
        return Poly({((), ()): value} if value != 0.0 else {})

    @staticmethod
    def symbol(sym):
        if is_atom(sym):
            return Poly({((), (sym,)): 1.0})
        return Poly({((sym,), ()): 1.0})

    @staticmethod
    def lift(value):
        if isinstance(value, Poly):
            return value
        return Poly.const(value)

    # -- queries -------------------------------------------------------------
    def is_constant(self):
        return all(k == ((), ()) for k in self.terms)

    def is_param_only(self):
        return all(len(k[0]) == 0 for k in self.terms)

    def constant_value(self):
        if not self.is_constant():
            raise ValueError('polynomial is not constant')
        return self.terms.get(((), ()), 0.0)

    def var_degree(self):
        return max([len(k[0]) for k in self.terms] + [0])

    def variables(self):
        out = set()
        for k in self.terms:
            out.update(k[0])
        return out

    def atoms(self):
        out = set()
        for k in self.terms:
            out.update(k[1])
        return out

    def by_var_monomial(self):
        """Group as {vars_tuple: Poly over atoms only}."""
        out = {}
        for (v, a), c in self.terms.items():
            out.setdefault(v, {})[((), a)] = c
        return {v: Poly(t) for v, t in out.items()}

    def substitute_symbols(self, mapping):
        """Rename symbols (int -> int); used to resolve `define_symbol`
        placeholders (`basics/optilayer.py:209-230` translate_symbols)."""
        out = {}
        for (v, a), c in self.terms.items():
            syms = [mapping.get(s, s) for s in v + a]
            nv = tuple(sorted(s for s in syms if not is_atom(s)))
            na = tuple(sorted(s for s in syms if is_atom(s)))
            key = (nv, na)
            out[key] = out.get(key, 0.0) + c
        return Poly({k: c for k, c in out.items() if c != 0.0})

    def evaluate(self, values):
        """values: dict sym -> float (or callable sym -> float)."""
        get = values.__getitem__ if not callable(values) else values
        total = 0.0
        for (v, a), c in self.terms.items():
            m = c
            for s in v:
                m *= get(s)
            for s in a:
                m *= get(s)
            total += m
        return total

    # -- ring operations -------------------------------------------------------
    def _broadcast(self, other, op):
        return np.array([op(o) for o in other.ravel()],
                        dtype=object).reshape(other.shape)

    def __add__(self, other):
        if isinstance(other, np.ndarray):
            return self._broadcast(other, lambda o: self + o)
        if not isinstance(other, _SCALARS):
            return NotImplemented
        other = Poly.lift(other)
        out = dict(self.terms)
        for k, c in other.terms.items():
            nc = out.get(k, 0.0) + c
            if nc == 0.0:
                out.pop(k, None)
            else:
                out[k] = nc
        return Poly(out)

    __radd__ = __add__

    def __neg__(self):
        return Poly({k: -c for k, c in self.terms.items()})

    def __sub__(self, other):
        if isinstance(other, np.ndarray):
            return self._broadcast(other, lambda o: self - o)
        if not isinstance(other, _SCALARS):
            return NotImplemented
        return self + (-Poly.lift(other))

    def __rsub__(self, other):
        if isinstance(other, np.ndarray):
            return self._broadcast(other, lambda o: o - self)
        if not isinstance(other, _SCALARS):
            return NotImplemented
        return Poly.lift(other) + (-self)

    def __mul__(self, other):
        if isinstance(other, np.ndarray):
            return self._broadcast(other, lambda o: self * o)
        if isinstance(other, (int, float, np.floating, np.integer)):
            other = float(other)
            if other == 0.0:
                return Poly()
            return Poly({k: c * other for k, c in self.terms.items()})
        if not isinstance(other, Poly):
            return NotImplemented
        a, b = self, other
        da, db = a.var_degree(), b.var_degree()
        if da + db > LIFT_CAP and SymbolTable._stack:
            # LIFTING (round 5): the product would have more variable factors than a template term carries (include/omgx.h
            # OMGX_TERM_VARS = 4).  The factor of higher degree is replaced by an auxiliary variable `aux` with the equality row
            # aux - factor = 0 (SymbolTable.lift: one auxiliary per distinct polynomial, e.g. per coefficient of a product
            # spline), until the product fits: the NLP the solver sees is the lifted one -- same minimisers in the caller's
            # variables -- with its auxiliaries and their rows appended behind the caller's x and g (template.py).
            table = SymbolTable.current()
            while da + db > LIFT_CAP and max(da, db) > 1:
                if da >= db:
                    a, da = table.lift(a), 1
                else:
                    b, db = table.lift(b), 1
        out = {}
        for (v1, a1), c1 in a.terms.items():
            for (v2, a2), c2 in b.terms.items():
                key = (tuple(sorted(v1 + v2)), tuple(sorted(a1 + a2)))
                out[key] = out.get(key, 0.0) + c1 * c2
        return Poly({k: c for k, c in out.items() if c != 0.0})

    __rmul__ = __mul__

    def __pow__(self, power):
        if not isinstance(power, (int, np.integer)) or power < 0:
            raise TypeError('exponent must be a non-negative integer')
        out = Poly.const(1.0)
        for _ in range(int(power)):
            out = out * self
        return out

    def __truediv__(self, other):
        if isinstance(other, (int, float, np.floating, np.integer)):
            return self * (1.0 / float(other))
        other = Poly.lift(other)
        if other.is_constant():
            return self * (1.0 / other.constant_value())
        table = SymbolTable.current()
        if not (self.is_param_only() and other.is_param_only()):
            # a quotient that involves variables (e.g. t / T with a free end time T): the auxiliary variable q with the
            # equality row q * denominator - numerator = 0 (round 5, see __mul__)
            return table.quotient(self, other)
        return Poly.symbol(table.new_div_atom(self, other))

    def __rtruediv__(self, other):
        return Poly.lift(other) / self

    def _unary_atom(self, kind, fun):
        """cos / sin of a parameter-only polynomial (the orientation of a rotating obstacle, `environment/obstacle.py:299-306`:
        cos(theta - t omega)) as a derived atom; numpy's ufuncs reach these methods on object arrays."""
        if self.is_constant():
            return Poly.const(float(fun(self.constant_value())))
        if not self.is_param_only():
            raise TypeError('%s is only defined for parameter-only polynomials' % kind)
        return Poly.symbol(SymbolTable.current().new_fun_atom(kind, self))

    def cos(self):
        return self._unary_atom('cos', np.cos)

    def sin(self):
        return self._unary_atom('sin', np.sin)

    def __repr__(self):
        if not self.terms:
            return 'Poly(0)'
        parts = []
        for (v, a), c in sorted(self.terms.items()):
            parts.append('%g' % c + ''.join('*x%d' % s for s in v) +
                         ''.join('*p%d' % (s - _ATOM_BASE) for s in a))
        return 'Poly(' + ' + '.join(parts) + ')'


_SCALARS = (Poly, int, float, np.floating, np.integer)


def as_poly_array(values):
    """1-D object array of Poly from numbers / Polys."""
    values = list(values) if not isinstance(values, np.ndarray) else values
    out = np.empty(len(values), dtype=object)
    for i, v in enumerate(values):
        out[i] = Poly.lift(v)
    return out


def is_symbolic(x):
    if isinstance(x, Poly):
        return True
    return isinstance(x, np.ndarray) and x.dtype == object


def vertcat(*args):
    """Stack scalars/vectors into one coefficient vector (numeric if all
    entries are numeric, object array of Poly otherwise)."""
    flat = []
    for a in args:
        if isinstance(a, np.ndarray):
            flat.extend(a.ravel().tolist())
        elif isinstance(a, (list, tuple)):
            flat.extend(a)
        else:
            flat.append(a)
    if any(isinstance(f, Poly) for f in flat):
        return as_poly_array(flat)
    return np.array(flat, dtype=float)


def matvec(T, coeffs):
    """T (dense ndarray or scipy sparse) times a coefficient vector that may
    hold Polys.  Mirrors `csr_matrix_alt.dot` (`basics/spline.py:97-114`)."""
    if not is_symbolic(coeffs):
        return np.asarray(T @ np.asarray(coeffs, dtype=float))
    T = np.asarray(T.todense()) if hasattr(T, 'todense') else np.asarray(T)
    coeffs = np.asarray(coeffs, dtype=object)
    out = np.empty(T.shape[0], dtype=object)
    for i in range(T.shape[0]):
        acc = Poly()
        for j in np.nonzero(T[i])[0]:
            acc = acc + coeffs[j] * float(T[i, j])
        out[i] = acc
    return out


class SymbolTable(object):
    """Allocates symbol ids and records derived atoms.

    atoms layout per agent at run time:  [raw parameters p | derived atoms],
    derived atoms evaluated in creation order by a straight-line program:

      ('div', num_poly, den_poly)             -> 1 atom
      ('cos' | 'sin', arg_poly)               -> 1 atom
      ('bspl', knots, degree, u_atom)         -> len(basis) consecutive atoms,
                                                 B_i(u) with the reference's
                                                 interval convention
                                                 (`basics/spline.py:131-136`)
    """
    _stack = []

    def __init__(self):
        self.n_var_syms = 0
        self.n_atoms = 0            # raw + derived, in creation order
        self.atom_info = []         # per atom: ('raw',) | ('div', id) | ('bspl', id, i)
        self.derived = []           # program entries
        self._bspl_cache = {}
        self._div_cache = {}
        self.lifted = []            # auxiliary variables in creation order: (symbol, row polynomial that must vanish)
        self._lift_cache = {}
        self.quotient_num = {}      # quotient auxiliary -> its numerator (omgx_shim checks what a variable evaluation point is)

    # context handling so Poly.__truediv__ can reach the active table
    def __enter__(self):
        SymbolTable._stack.append(self)
        return self

    def __exit__(self, *exc):
        SymbolTable._stack.pop()

    @staticmethod
    def current():
        if not SymbolTable._stack:
            raise RuntimeError('no active SymbolTable (problem not under '
                               'construction)')
        return SymbolTable._stack[-1]

    def new_vars(self, n):
        ids = list(range(self.n_var_syms, self.n_var_syms + n))
        self.n_var_syms += n
        return ids

    def lift(self, poly):
        """The auxiliary variable that stands for `poly` (one per distinct polynomial): Poly of degree 1."""
        key = _exact_key(poly)
        if key not in self._lift_cache:
            sym = self.new_vars(1)[0]
            self._lift_cache[key] = sym
            self.lifted.append((sym, Poly.symbol(sym) + (-poly)))
        return Poly.symbol(self._lift_cache[key])

    def quotient(self, num, den):
        """The auxiliary variable q = num / den, defined by the row q * den - num = 0."""
        key = ('quot', _exact_key(num), _exact_key(den))
        if key not in self._lift_cache:
            sym = self.new_vars(1)[0]
            self._lift_cache[key] = sym
            with self:
                row = Poly.symbol(sym) * Poly.lift(den) + (-Poly.lift(num))      # (may lift `den` first: its auxiliary then precedes q's row)
            self.lifted.append((sym, row))
            self.quotient_num[sym] = Poly.lift(num)
        return Poly.symbol(self._lift_cache[key])

    def new_raw_atoms(self, n):
        ids = []
        for _ in range(n):
            ids.append(_ATOM_BASE + self.n_atoms)
            self.atom_info.append(('raw',))
            self.n_atoms += 1
        return ids

    def new_div_atom(self, num, den):
        key = (_exact_key(num), _exact_key(den))      # (repr prints six digits: two offsets that differ beyond them would share an atom)
        if key in self._div_cache:
            return self._div_cache[key]
        sym = _ATOM_BASE + self.n_atoms
        self.atom_info.append(('div', len(self.derived)))
        self.derived.append(('div', num, den, sym))
        self.n_atoms += 1
        self._div_cache[key] = sym
        return sym

    def new_fun_atom(self, kind, arg):
        key = (kind, _exact_key(arg))
        if key in self._div_cache:
            return self._div_cache[key]
        sym = _ATOM_BASE + self.n_atoms
        self.atom_info.append((kind, len(self.derived)))
        self.derived.append((kind, arg, None, sym))
        self.n_atoms += 1
        self._div_cache[key] = sym
        return sym

    def new_bspl_atoms(self, knots, degree, u_poly):
        """Atoms for all basis functions of (knots, degree) evaluated at the
        parameter-only polynomial `u_poly` (must be a single atom)."""
        (key_u, coef), = u_poly.terms.items()
        if coef != 1.0 or key_u[0] or len(key_u[1]) != 1:
            raise TypeError('evaluation point must be a single atom (e.g. t/T)')
        u_sym = key_u[1][0]
        key = (tuple(np.asarray(knots, float).tolist()), int(degree), u_sym)
        if key in self._bspl_cache:
            return self._bspl_cache[key]
        n = len(knots) - degree - 1
        first = _ATOM_BASE + self.n_atoms
        for i in range(n):
            self.atom_info.append(('bspl', len(self.derived), i))
            self.n_atoms += 1
        ids = list(range(first, first + n))
        self.derived.append(('bspl', np.asarray(knots, float), int(degree),
                             u_sym, first))
        self._bspl_cache[key] = ids
        return ids
