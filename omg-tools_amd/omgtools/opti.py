# This file is derived from OMG-tools (meco-group/omg-tools, `omgtools/basics/optilayer.py` (API surface)).
#
# OMG-tools -- Optimal Motion Generation-tools
# Copyright (C) 2016 Ruben Van Parys & Tim Mercy, KU Leuven.
# All rights reserved.
#
# OMG-tools is free software; you can redistribute it and/or
# modify it under the terms of the GNU Lesser General Public
# License as published by the Free Software Foundation; either
# version 3 of the License, or (at your option) any later version.
# This software is distributed in the hope that it will be useful,
# but WITHOUT ANY WARRANTY; without even the implied warranty of
# MERCHANTABILITY or FITNESS FOR A PARTICULAR PURPOSE. See the GNU
# Lesser General Public License for more details.
#
# You should have received a copy of the GNU Lesser General Public
# License along with this program; if not, write to the Free Software
# Foundation, Inc., 51 Franklin Street, Fifth Floor, Boston, MA 02110-1301 USA
#
# Modifications: the public classes, option names, method order and messages of the files named
# above are kept so that scripts written for OMG-tools run unchanged where the original package is
# not installed (benchmark and test tiers of this repository); the CasADi expression layer underneath
# is replaced by explicit polynomials (symbolic.py) and the solver call by the HIP path (backend.py).
# Where the original package IS installed, use omgx_shim instead: it runs the original classes themselves.

"""Problem registry: `OptiChild` / `OptiFather` with a *numeric* NLP template.

Mirrors the public surface of the reference's `basics/optilayer.py`
(OptiChild.define_* 556-669, OptiFather.construct_problem 180-198,
get/set_variables 332-386, set_parameters 427-445, update_bounds 313-319,
transform_primal_splines 470-490) so vehicle/environment/problem classes read
like the reference's.  Instead of CasADi structs + `nlpsol`, `construct_problem`
flattens all registered polynomial constraints into an `NLPTemplate`
(template.py) that the HIP backend uploads once per batch.

Flat vector layout (identical to the reference, `optilayer.py:225-272`):
children in registration order, entries in definition order, every entry
stored column-major (spline k of an (L x n) entry occupies [k*L, (k+1)*L)).
"""
import collections as col
from itertools import groupby

import numpy as np

from .symbolic import (Poly, SymbolTable, as_poly_array, is_symbolic, matvec,
                       is_atom)
from .splines import BSpline

inf = float('inf')


class StructVector(object):
    """Flat float vector with ('child label', 'entry name') access; the
    stand-in for casadi.tools struct instances (`optilayer.py:225-247`)."""

    def __init__(self, layout, data=None):
        self.layout = layout                 # OrderedDict[(label,name)] -> (off, rows, cols)
        self.size = sum(r * c for _, r, c in layout.values())
        if data is None:
            self.cat = np.zeros(self.size)
        else:
            data = np.asarray(data, dtype=float).reshape(-1)
            if data.size == 1 and self.size != 1:
                data = np.full(self.size, data[0])
            if data.size != self.size:
                raise ValueError('expected %d values, got %d' % (self.size, data.size))
            self.cat = data.copy()

    def _key(self, key):
        if key not in self.layout:
            raise KeyError('no entry %s' % (key,))
        return self.layout[key]

    def __getitem__(self, key):
        if isinstance(key, str):
            return self.prefix(key)
        off, r, c = self._key(tuple(key))
        return self.cat[off:off + r * c].reshape((r, c), order='F')

    def __setitem__(self, key, value):
        if isinstance(key, str):
            lo, hi = self.child_range(key)
            self.cat[lo:hi] = np.asarray(value, dtype=float).reshape(-1)
            return
        off, r, c = self._key(tuple(key))
        value = np.asarray(value, dtype=float)
        if value.size == 1:
            self.cat[off:off + r * c] = value.reshape(-1)[0]
        elif value.ndim == 2 and value.shape == (r, c):
            self.cat[off:off + r * c] = value.reshape(-1, order='F')
        else:
            self.cat[off:off + r * c] = value.reshape(-1)

    def child_range(self, label):
        offs = [(o, o + r * c) for (l, _), (o, r, c) in self.layout.items() if l == label]
        if not offs:
            raise KeyError(label)
        return offs[0][0], offs[-1][1]

    def prefix(self, label):
        return {n: self[(l, n)] for (l, n) in self.layout if l == label}

    def copy(self):
        return StructVector(self.layout, self.cat)

    def __array__(self, dtype=None, copy=None):
        return self.cat if dtype is None else self.cat.astype(dtype)

    def __len__(self):
        return self.size


class OptiChild(object):
    _labels = []

    def __init__(self, label):
        self.label = OptiChild._make_label(label)
        self.father = None
        self._clear()

    def _clear(self):
        self._variables = col.OrderedDict()      # name -> (sym ids ndarray (r,c))
        self._parameters = col.OrderedDict()
        self._substitutes = col.OrderedDict()
        self._values = col.OrderedDict()
        self._splines_prim = col.OrderedDict()
        self._splines_dual = col.OrderedDict()
        self._constraints = col.OrderedDict()     # name -> (exprs, lb, ub, shutdown)
        self._objective = Poly()
        self._constraint_cnt = 0
        self.n_cons = 0

    def __str__(self):
        return self.label

    __repr__ = __str__

    def _add_label(self, name):
        return name + '_' + self.label

    @classmethod
    def _make_label(cls, label):
        parts = [''.join(g) for _, g in groupby(label, str.isalpha)]
        if parts[-1].isdigit():
            if label in cls._labels:
                return cls._make_label(''.join(parts[:-1]) + str(int(parts[-1]) + 1))
            cls._labels.append(label)
            return label
        return cls._make_label(label + '0')

    # -- symbols --------------------------------------------------------------
    def _table(self):
        if self.father is None:
            raise RuntimeError('%s is not attached to a problem' % self.label)
        return self.father.table

    def _define(self, name, size0, size1, store, value, kind):
        table = self._table()
        n = size0 * size1
        if name in store and store[name].shape == (size0, size1):
            # A name defined twice by one object is ONE entry of the reference's flat vectors: its
            # expressions are translated by symbol name (`optilayer.py:198-224, 595-602`), so both
            # definitions address the same variable -- e.g. the terminal slacks g0, g1 of a
            # multi-vehicle FixedTPoint2point (`point2point.py:160-163`) are shared by the vehicles.
            ids = store[name]
        else:
            ids = table.new_vars(n) if kind == 'var' else table.new_raw_atoms(n)
            ids = np.array(ids, dtype=np.int64).reshape((size0, size1), order='F')
            store[name] = ids
        if value is None:
            self._values[name] = np.zeros((size0, size1))
        else:
            arr = np.asarray(value, dtype=float)
            if arr.shape != (size0, size1):
                arr = arr.reshape((size0, size1), order='F')
            self._values[name] = arr
        syms = np.empty((size0, size1), dtype=object)
        for i in range(size0):
            for j in range(size1):
                syms[i, j] = Poly.symbol(int(ids[i, j]))
        if size1 == 1:
            return syms[:, 0] if size0 > 1 else syms[0, 0]
        return syms

    def define_variable(self, name, size0=1, size1=1, **kwargs):
        return self._define(name, size0, size1, self._variables,
                            kwargs.get('value'), 'var')

    def define_parameter(self, name, size0=1, size1=1, **kwargs):
        return self._define(name, size0, size1, self._parameters,
                            kwargs.get('value'), 'par')

    def define_symbol(self, name, size0=1, size1=1):
        """A quantity owned by another object (`optilayer.py:556-557`).  The
        owner must have defined it already (true for every in-scope problem:
        the problem defines T, t before its children are constructed)."""
        owner = self.father.find_definition(name)
        if owner is None:
            raise ValueError('Symbol %s, requested by %s, is not (yet) defined as '
                             'parameter or variable by any object' % (name, self.label))
        child, store = owner
        ids = store[name]
        syms = np.empty(ids.shape, dtype=object)
        for idx in np.ndindex(ids.shape):
            syms[idx] = Poly.symbol(int(ids[idx]))
        if ids.shape[1] == 1:
            return syms[:, 0] if ids.shape[0] > 1 else syms[0, 0]
        return syms

    def _define_spline(self, name, size0, size1, store, basis, value, kind):
        if size1 > 1:
            return [self._define_spline(name + str(l), size0, 1, store, basis, value, kind)
                    for l in range(size1)]
        syms = self._define(name, len(basis), size0, store, value, kind)
        syms = np.asarray(syms, dtype=object).reshape((len(basis), size0), order='F')
        self._splines_prim[name] = {'basis': basis, 'init': None}
        return [BSpline(basis, syms[:, k]) for k in range(size0)]

    def define_spline_variable(self, name, size0=1, size1=1, **kwargs):
        return self._define_spline(name, size0, size1, self._variables,
                                   kwargs.get('basis', getattr(self, 'basis', None)),
                                   kwargs.get('value'), 'var')

    def define_spline_parameter(self, name, size0=1, size1=1, **kwargs):
        return self._define_spline(name, size0, size1, self._parameters,
                                   kwargs.get('basis', getattr(self, 'basis', None)),
                                   kwargs.get('value'), 'par')

    def define_substitute(self, name, expr):
        """Named expression of variables/parameters (`optilayer.py:585-608`).
        Kept as the expression itself: no separate symbol is needed because
        constraints are expanded polynomials anyway."""
        if isinstance(expr, list):
            return [self.define_substitute(name + str(l), e) for l, e in enumerate(expr)]
        if name in self._substitutes:
            raise ValueError('Name %s already used for substitutes!' % name)
        if isinstance(expr, BSpline):
            self._splines_prim[name] = {'basis': expr.basis, 'init': None}
            self._substitutes[name] = as_poly_array(expr.coeffs)
        else:
            self._substitutes[name] = as_poly_array(np.atleast_1d(expr))
        return expr

    def set_value(self, name, value):
        self._values[name] = value

    def define_constraint(self, expr, lb, ub, shutdown=False, name=None, skip=[]):
        if isinstance(expr, (float, int)):
            return
        name = ('c_' if name is None else name + '_') + str(self._constraint_cnt)
        self._constraint_cnt += 1
        if isinstance(expr, BSpline):
            coeffs = expr.coeffs
            if skip:
                end = len(coeffs) - skip[1]
                coeffs = coeffs[skip[0]:end]
            self._splines_dual[name] = {'basis': expr.basis, 'init': None}
        else:
            coeffs = np.atleast_1d(expr)
        coeffs = as_poly_array(coeffs)
        n = len(coeffs)
        self._constraints[name] = (coeffs, lb * np.ones(n), ub * np.ones(n), shutdown)
        self.n_cons += n

    def define_objective(self, expr):
        self._objective = self._objective + expr

    def reset(self):
        self._clear()

    def set_parameters(self, time):
        return {}


class OptiFather(object):

    def __init__(self, children=None):
        self.children = col.OrderedDict()
        self.table = SymbolTable()
        for child in (children or []):
            self.add(child)

    def add(self, children):
        children = children if isinstance(children, list) else [children]
        for child in children:
            self.children[child.label] = child
            child.father = self

    def find_definition(self, name):
        for child in self.children.values():
            if name in child._variables:
                return child, child._variables
            if name in child._parameters:
                return child, child._parameters
        return None

    def reset(self):
        self.table = SymbolTable()
        for child in self.children.values():
            child.father = self
            child.reset()

    # -- layout ------------------------------------------------------------------
    @staticmethod
    def _layout(children, attr):
        layout, off = col.OrderedDict(), 0
        for label, child in children.items():
            for name, ids in getattr(child, attr).items():
                layout[(label, name)] = (off, ids.shape[0], ids.shape[1])
                off += ids.size
        return layout

    def construct_problem(self, options, name='', problem=None):
        from .template import NLPTemplate
        self._var_layout = self._layout(self.children, '_variables')
        self._par_layout = self._layout(self.children, '_parameters')
        con_layout, off = col.OrderedDict(), 0
        lb, ub = [], []
        self._constraint_shutdown = {}
        for label, child in self.children.items():
            for cname, (exprs, clb, cub, shutdown) in child._constraints.items():
                con_layout[(label, child._add_label(cname))] = (off, len(exprs), 1)
                off += len(exprs)
                lb.append(clb)
                ub.append(cub)
                if shutdown:
                    self._constraint_shutdown[(label, child._add_label(cname))] = shutdown
        self._con_layout = con_layout
        self._lb = StructVector(con_layout, np.concatenate(lb) if lb else [])
        self._ub = StructVector(con_layout, np.concatenate(ub) if ub else [])
        self.template = NLPTemplate.from_father(self)
        self.problem_description = {'template': self.template, 'opt': options}
        buildtime = 0.
        if problem is None:
            from .backend import create_nlp
            problem, buildtime = create_nlp(self.template, options, name)
        self.init_variables()
        self.init_parameters()
        return problem, buildtime

    # -- numeric state --------------------------------------------------------------
    def update_bounds(self, current_time):
        lb, ub = self._lb.copy(), self._ub.copy()
        for key, shutdown in self._constraint_shutdown.items():
            fun = shutdown if callable(shutdown) else eval('lambda t: %s' % shutdown)
            if fun(current_time):
                lb[key], ub[key] = -inf, +inf
        return lb, ub

    def init_variables(self):
        variables = StructVector(self._var_layout)
        for label, child in self.children.items():
            for name in child._variables:
                variables[(label, name)] = child._values[name]
        self._var_result = variables
        self._dual_var_result = StructVector(self._con_layout)

    def init_parameters(self):
        self.set_parameters(0.)

    def set_variables(self, variables, child=None, name=None):
        if child is None:
            self._var_result = StructVector(self._var_layout, variables)
        elif name is None:
            self._var_result[child.label] = variables
        else:
            self._var_result[(child.label, name)] = np.asarray(variables, dtype=float)

    def set_dual_variables(self, variables, child=None, name=None):
        if child is None:
            self._dual_var_result = StructVector(self._con_layout, variables)
        elif name is None:
            self._dual_var_result[child.label] = variables
        else:
            self._dual_var_result[(child.label, name)] = variables

    def _substitute_value(self, child, name):
        polys = child._substitutes[name]
        tpl = self.template
        x, atoms = self._var_result.cat, tpl.eval_atoms_host(self._par_result.cat)
        return np.array([tpl.eval_poly_host(p, x, atoms) for p in polys])

    def get_variables(self, child=None, name=None, **kwargs):
        if child is None:
            return self._var_result
        if name is None:
            return self._var_result.prefix(child.label)
        as_spline = (name in child._splines_prim and
                     not ('spline' in kwargs and not kwargs['spline']))
        if name in child._substitutes:
            coeffs = self._substitute_value(child, name).reshape(-1, 1)
        else:
            coeffs = np.array(self._var_result[(child.label, name)])
        if as_spline:
            basis = child._splines_prim[name]['basis']
            return [BSpline(basis, coeffs[:, k]) for k in range(coeffs.shape[1])]
        return coeffs

    def get_dual_variables(self, child=None, name=None, **kwargs):
        if child is None:
            return self._dual_var_result
        raise RuntimeError('Error dual variables')

    def get_parameters(self, child=None, name=None, **kwargs):
        if child is None:
            return self._par_result
        if name is None:
            return self._par_result.prefix(child.label)
        coeffs = np.array(self._par_result[(child.label, name)])
        if name in child._splines_prim and not ('spline' in kwargs and not kwargs['spline']):
            basis = child._splines_prim[name]['basis']
            return [BSpline(basis, coeffs[:, k]) for k in range(coeffs.shape[1])]
        return coeffs

    def set_parameters(self, time):
        self._par_result = StructVector(self._par_layout)
        parameters = {}
        for child in self.children.values():
            for chld, dic in child.set_parameters(time).items():
                mine = parameters.setdefault(chld, {})
                for key in dic:
                    if key in mine:
                        raise ValueError('Same parameter set multiple times!')
                mine.update(dic)
        for label, child in self.children.items():
            for name in child._parameters:
                if child in parameters and name in parameters[child]:
                    self._par_result[(label, name)] = parameters[child][name]
                else:
                    self._par_result[(label, name)] = child._values[name]
        return self._par_result

    # -- spline transformations ----------------------------------------------------------
    def init_transformations(self, init_primal_transform, init_dual_transform=None):
        cache = {}
        for child in self.children.values():
            for name, spl in child._splines_prim.items():
                if name in child._variables or name in child._substitutes:
                    basis = spl['basis']
                    if basis not in cache:
                        cache[basis] = init_primal_transform(basis)
                    spl['init'] = cache[basis]

    def shifted_entries(self, seg_shift=None, every_spline=False):
        """Variable entries that `transform_primal_splines` touches: names that
        contain 'seg<n>' with n in seg_shift (`optilayer.py:470-490`).  `every_spline`
        selects the rule of the generated C++ instead, which shifts every spline
        variable, `g*` / `eps_*` included (`export/export.py:414-439`)."""
        seg_shift = [0] if seg_shift is None else \
            (seg_shift if isinstance(seg_shift, list) else [seg_shift])
        out = []
        for label, child in self.children.items():
            for name, spl in child._splines_prim.items():
                if every_spline and name in child._variables:
                    out.append((label, name, spl))
                    continue
                if name in child._variables and 'seg' in name and \
                        int(name[name.index('seg') + 3]) in seg_shift:
                    out.append((label, name, spl))
        return out

    def transform_primal_splines(self, transform_fun, seg_shift=None):
        for label, name, spl in self.shifted_entries(seg_shift):
            cur = self._var_result[(label, name)]
            if spl['init'] is not None:
                self._var_result[(label, name)] = transform_fun(cur, spl['basis'], spl['init'])
            else:
                self._var_result[(label, name)] = transform_fun(cur, spl['basis'])
