"""`NLPTemplate`: the flat, numeric description of one agent's NLP.

This replaces the CasADi expression graphs the reference hands to `nlpsol`
(`basics/optilayer.py:49-60, 180-198`).  Everything the device needs to
evaluate f, g, their Jacobian and the Lagrangian Hessian for *any* parameter
vector p is a handful of integer/float arrays:

  atoms      a = [p (n_par) | derived atoms]; derived atoms come from a
             straight-line program (`prog`): DIV(ppoly, ppoly), BSPL(basis, u), COS(ppoly), SIN(ppoly)
  ppolys     polynomials over atoms (CSR over monomials over atom indices)
  slots      per-agent scalars: slot s = ppoly[slot_pp[s]](a)
  terms      row r:  g_r(x) = sum_t  coef_t * S_t * prod_{v in vars_t} x_v,
             S_t = 1 if slot_t < 0 else slot value; <= 4 variables per term;
             row n_con holds the objective.

The x / p / g orderings are the reference's (`optilayer.py:225-272`).
"""
import numpy as np

from .symbolic import Poly, is_atom, _ATOM_BASE

OP_DIV, OP_BSPL, OP_COS, OP_SIN = 0, 1, 2, 3
MAX_TERM_VARS = 4      # = OMGX_TERM_VARS of include/omgx.h


class NLPTemplate(object):

    # ------------------------------------------------------------------ build
    @classmethod
    def from_father(cls, father):
        """From this package's own front end (opti.OptiFather)."""
        self = cls()
        self.var_layout = father._var_layout
        self.par_layout = father._par_layout
        self.con_layout = father._con_layout
        self.lb = father._lb.cat.copy()
        self.ub = father._ub.cat.copy()
        # symbol id -> flat index
        var_index, atom_index = {}, {}
        for label, child in father.children.items():
            for name, ids in child._variables.items():
                off = self.var_layout[(label, name)][0]
                for k, sym in enumerate(ids.reshape(-1, order='F')):
                    var_index[int(sym)] = off + k
            for name, ids in child._parameters.items():
                off = self.par_layout[(label, name)][0]
                for k, sym in enumerate(ids.reshape(-1, order='F')):
                    atom_index[int(sym)] = off + k
        rows = []
        for label, child in father.children.items():
            for cname, (exprs, _, _, _) in child._constraints.items():
                rows.extend(list(exprs))
        objective = Poly()
        for child in father.children.values():
            objective = objective + child._objective
        self.var_layout, self.con_layout = dict(self.var_layout), dict(self.con_layout)
        self._append_lifted(father.table, var_index, rows, len(var_index))
        self._build(father.table, var_index, atom_index, rows, objective)
        # default values
        self.x_init = father_init_vector(father, '_variables', self.var_layout)
        return self

    @classmethod
    def from_polys(cls, table, var_syms, par_syms, rows, objective, lb, ub,
                   var_layout=None, par_layout=None, con_layout=None):
        """From polynomial rows produced by any front end: `var_syms` / `par_syms` are the symbol ids of
        x and p in flat order, `rows` the constraint polynomials g, `objective` f -- this is how the
        reference's own construct code (its CasADi graphs evaluated on `Poly` values, omgx_shim) becomes
        a template."""
        self = cls()
        self.var_layout = dict(var_layout or {('x', 'x'): (0, len(var_syms), 1)})
        self.par_layout = par_layout or {('p', 'p'): (0, len(par_syms), 1)}
        self.con_layout = dict(con_layout or {('g', 'g'): (0, len(rows), 1)})
        self.lb, self.ub = np.asarray(lb, float).copy(), np.asarray(ub, float).copy()
        var_index = {int(sym): k for k, sym in enumerate(var_syms)}
        atom_index = {int(sym): k for k, sym in enumerate(par_syms)}
        rows = list(rows)
        self._append_lifted(table, var_index, rows, len(var_syms))
        self._build(table, var_index, atom_index, rows, Poly.lift(objective))
        self.x_init = np.zeros(self.n_var)
        return self

    def _append_lifted(self, table, var_index, rows, n_user_var):
        """Auxiliary variables of products / quotients that were lifted while the rows were formed (symbolic.py,
        `Poly.__mul__`): a block ('lifted', 'aux') behind the caller's variables, their equality rows behind the caller's rows.
        The caller's x / g offsets do not move; `lift_extend` / `lift_strip` take a caller's vectors to the template's and back."""
        self.n_lift = len(table.lifted)
        if not self.n_lift:
            return
        n_user_con = len(rows)
        for k, (sym, row) in enumerate(table.lifted):
            var_index[int(sym)] = n_user_var + k
            rows.append(row)
        self.var_layout[('lifted', 'aux')] = (n_user_var, self.n_lift, 1)
        self.con_layout[('lifted', 'aux')] = (n_user_con, self.n_lift, 1)
        self.lb = np.r_[self.lb, np.zeros(self.n_lift)]
        self.ub = np.r_[self.ub, np.zeros(self.n_lift)]

    def _build(self, table, var_index, atom_index, rows, objective):
        self.n_var = sum(r * c for _, r, c in self.var_layout.values())
        self.n_par = sum(r * c for _, r, c in self.par_layout.values())
        self.n_con = sum(r * c for _, r, c in self.con_layout.values())
        n_atoms = self.n_par
        for entry in table.derived:
            if entry[0] in ('div', 'cos', 'sin'):
                atom_index[entry[3]] = n_atoms
                n_atoms += 1
            else:
                n = len(entry[1]) - entry[2] - 1
                for i in range(n):
                    atom_index[entry[4] + i] = n_atoms + i
                n_atoms += n
        self.n_atoms = n_atoms
        self._var_index, self._atom_index = var_index, atom_index

        # ppoly table
        self._pp_keys = {}
        self._pp_list = []          # list of [(coef, (atom idx...))]

        # derived-atom program
        prog, knots = [], []
        for entry in table.derived:
            if entry[0] == 'div':
                prog.append((OP_DIV, self._ppoly(entry[1]), self._ppoly(entry[2]),
                             atom_index[entry[3]], 0, 0))
            elif entry[0] in ('cos', 'sin'):
                prog.append((OP_COS if entry[0] == 'cos' else OP_SIN, self._ppoly(entry[1]), 0,
                             atom_index[entry[3]], 0, 0))
            else:
                _, kn, deg, u_sym, first = entry
                prog.append((OP_BSPL, len(knots), len(kn), deg,
                             self._atom(u_sym), atom_index[first]))
                knots.extend(kn.tolist())
        self.prog = np.array(prog, dtype=np.int32).reshape(-1, 6)
        self.knots = np.array(knots, dtype=np.float64)

        # rows -> terms
        self._slot_keys, slot_pp = {}, []
        assert len(rows) == self.n_con
        rows = list(rows) + [objective]

        row_ptr, t_coef, t_slot, t_nv, t_var = [0], [], [], [], []
        for poly in rows:
            poly = Poly.lift(poly)
            groups = poly.by_var_monomial()
            for vars_, ppoly in sorted(groups.items(), key=lambda kv: (len(kv[0]), kv[0])):
                if len(vars_) > MAX_TERM_VARS:
                    raise ValueError('constraint of degree %d in the variables is '
                                     'not supported' % len(vars_))
                coef, slot = self._coef_slot(ppoly, slot_pp)
                if coef == 0.0:
                    continue
                idx = sorted(self._var(v) for v in vars_)
                t_coef.append(coef)
                t_slot.append(slot)
                t_nv.append(len(idx))
                t_var.append(idx + [-1] * (MAX_TERM_VARS - len(idx)))
            row_ptr.append(len(t_coef))
        self.row_ptr = np.array(row_ptr, dtype=np.int32)
        self.t_coef = np.array(t_coef, dtype=np.float64)
        self.t_slot = np.array(t_slot, dtype=np.int32)
        self.t_nv = np.array(t_nv, dtype=np.int32)
        self.t_var = np.array(t_var, dtype=np.int32).reshape(-1, MAX_TERM_VARS)
        self.slot_pp = np.array(slot_pp, dtype=np.int32)
        self.n_slots = len(slot_pp)
        self.n_terms = len(t_coef)

        # freeze the ppoly CSR
        pp_ptr, pm_coef, pm_ptr, pm_atom = [0], [], [0], []
        for monos in self._pp_list:
            for c, atoms in monos:
                pm_coef.append(c)
                pm_atom.extend(atoms)
                pm_ptr.append(len(pm_atom))
            pp_ptr.append(len(pm_coef))
        self.pp_ptr = np.array(pp_ptr, dtype=np.int32)
        self.pm_coef = np.array(pm_coef, dtype=np.float64)
        self.pm_ptr = np.array(pm_ptr, dtype=np.int32)
        self.pm_atom = np.array(pm_atom, dtype=np.int32)

    def _var(self, sym):
        if sym not in self._var_index:
            raise ValueError('constraint references a variable that is not part '
                             'of the problem (re-defined or foreign symbol)')
        return self._var_index[sym]

    def _atom(self, sym):
        if sym not in self._atom_index:
            raise ValueError('constraint references a parameter that is not part '
                             'of the problem (re-defined or foreign symbol)')
        return self._atom_index[sym]

    def _ppoly(self, poly):
        monos = []
        for (v, a), c in sorted(poly.terms.items()):
            assert not v
            monos.append((float(c), tuple(sorted(self._atom(s) for s in a))))
        key = tuple(monos)
        if key not in self._pp_keys:
            self._pp_keys[key] = len(self._pp_list)
            self._pp_list.append(monos)
        return self._pp_keys[key]

    def _coef_slot(self, ppoly, slot_pp):
        """Split a coefficient polynomial into (constant factor, slot id)."""
        if ppoly.is_constant():
            return ppoly.constant_value() if ppoly.terms else 0.0, -1
        lead_key = sorted(ppoly.terms)[0]
        lead = ppoly.terms[lead_key]
        pp = self._ppoly(ppoly * (1.0 / lead))
        if pp not in self._slot_keys:
            self._slot_keys[pp] = len(slot_pp)
            slot_pp.append(pp)
        return lead, self._slot_keys[pp]

    # ------------------------------------------------------- host evaluation
    # (front-end bookkeeping only: substitutes, plotting; never the solve path)
    def eval_atoms_host(self, p):
        from .splines import BSplineBasis
        a = np.zeros(self.n_atoms)
        a[:self.n_par] = p
        for op, i0, i1, i2, i3, i4 in self.prog:
            if op == OP_DIV:
                a[i2] = self._pp_value(i0, a) / self._pp_value(i1, a)
            elif op in (OP_COS, OP_SIN):
                a[i2] = (np.cos if op == OP_COS else np.sin)(self._pp_value(i0, a))
            else:
                basis = BSplineBasis(self.knots[i0:i0 + i1], i2)
                a[i4:i4 + len(basis)] = basis.eval_basis([a[i3]])[0]
        return a

    def _pp_value(self, pp, a):
        total = 0.
        for m in range(self.pp_ptr[pp], self.pp_ptr[pp + 1]):
            v = self.pm_coef[m]
            for q in range(self.pm_ptr[m], self.pm_ptr[m + 1]):
                v *= a[self.pm_atom[q]]
            total += v
        return total

    def eval_slots_host(self, atoms):
        return np.array([self._pp_value(pp, atoms) for pp in self.slot_pp]) if self.n_slots else np.zeros(0)

    def eval_rows_host(self, x, atoms, rows, slots=None):
        """Values of the given rows (indices; n_con = the objective) at x from the term lists."""
        if slots is None:
            slots = self.eval_slots_host(atoms)
        out = np.zeros(len(rows))
        xe = np.r_[np.asarray(x, float), 1.0]                 # (index -1 -> the factor 1)
        for i, r in enumerate(rows):
            t0, t1 = int(self.row_ptr[r]), int(self.row_ptr[r + 1])
            c = self.t_coef[t0:t1] * np.where(self.t_slot[t0:t1] >= 0, slots[np.maximum(self.t_slot[t0:t1], 0)] if self.n_slots else 1.0, 1.0)
            out[i] = np.sum(c * np.prod(xe[self.t_var[t0:t1]], axis=1))
        return out

    def lift_extend(self, x_user, p, lbg=None, ubg=None):
        """A caller's x (and bounds) -> the template's: the auxiliary variables from their defining rows, in order (every row
        is linear in its own auxiliary: value at 0 and at 1 give it), the bounds of those rows zero."""
        n_lift = getattr(self, 'n_lift', 0)
        if not n_lift:
            return x_user, lbg, ubg
        nv, nc = self.n_var - n_lift, self.n_con - n_lift
        x = np.r_[np.asarray(x_user, float).reshape(-1)[:nv], np.zeros(n_lift)]
        atoms = self.eval_atoms_host(np.asarray(p, float).reshape(-1))
        slots = self.eval_slots_host(atoms)
        for k in range(n_lift):
            g0 = self.eval_rows_host(x, atoms, [nc + k], slots)[0]
            x[nv + k] = 1.0
            g1 = self.eval_rows_host(x, atoms, [nc + k], slots)[0]
            if not abs(g1 - g0) > 1e-300:
                raise ValueError('lifted auxiliary %d: its defining row does not depend on it at this point (a quotient by a variable '
                                 'that is zero at the initial guess): no admissible start' % k)
            x[nv + k] = -g0 / (g1 - g0)
        ext = lambda b: None if b is None else np.r_[np.asarray(b, float).reshape(-1)[:nc], np.zeros(n_lift)]
        return x, ext(lbg), ext(ubg)

    def lift_strip(self, x, lam_g):
        n_lift = getattr(self, 'n_lift', 0)
        if not n_lift:
            return x, lam_g
        return np.asarray(x)[..., :self.n_var - n_lift], np.asarray(lam_g)[..., :self.n_con - n_lift]

    def eval_poly_host(self, poly, x, atoms):
        def value(sym):
            return atoms[self._atom(sym)] if is_atom(sym) else x[self._var(sym)]
        return Poly.lift(poly).evaluate(value)

    # ------------------------------------------------------------- utilities
    def entry_range(self, label, name, which='var'):
        layout = {'var': self.var_layout, 'par': self.par_layout,
                  'con': self.con_layout}[which]
        off, r, c = layout[(label, name)]
        return off, off + r * c

    def block_table(self, which='var'):
        layout = {'var': self.var_layout, 'par': self.par_layout,
                  'con': self.con_layout}[which]
        return [(label, name, off, r, c) for (label, name), (off, r, c) in layout.items()]

    def flat_arrays(self):
        """Arrays handed through the C ABI (include/omgx.h `omgx_template`)."""
        return dict(prog=self.prog, knots=self.knots, pp_ptr=self.pp_ptr,
                    pm_coef=self.pm_coef, pm_ptr=self.pm_ptr, pm_atom=self.pm_atom,
                    slot_pp=self.slot_pp, row_ptr=self.row_ptr, t_coef=self.t_coef,
                    t_slot=self.t_slot, t_nv=self.t_nv, t_var=self.t_var)


    # ------------------------------------------------------------- files
    _SCALARS = ('n_var', 'n_par', 'n_con', 'n_atoms', 'n_slots', 'n_terms', 'n_lift')

    def to_npz(self, path, **extra):
        """The template as one .npz (flat arrays, bounds, layouts): fixtures of problem classes whose front end is not in
        this package (tests/golden/dubins_fixedT.npz comes from the reference's own Dubins class on `omgx_shim`)."""
        lay = {}
        for which in ('var', 'par', 'con'):
            table = self.block_table(which)
            lay[which + '_names'] = np.array(['%s\t%s' % (label, name) for label, name, _, _, _ in table])
            lay[which + '_dims'] = np.array([[off, r, c] for _, _, off, r, c in table], dtype=np.int64).reshape(-1, 3)
        np.savez_compressed(path, lb=self.lb, ub=self.ub, x_init=getattr(self, 'x_init', np.zeros(self.n_var)),
                            scalars=np.array([getattr(self, k, 0) for k in self._SCALARS], dtype=np.int64),
                            **self.flat_arrays(), **lay, **extra)
        return path

    @classmethod
    def from_npz(cls, path):
        d = np.load(path)
        self = cls()
        self.n_lift = 0                    # (files written before round 5 carry six scalars)
        for k, v in zip(cls._SCALARS, d['scalars']):
            setattr(self, k, int(v))
        for k in ('prog', 'knots', 'pp_ptr', 'pm_coef', 'pm_ptr', 'pm_atom', 'slot_pp', 'row_ptr', 't_coef', 't_slot',
                  't_nv', 't_var', 'lb', 'ub', 'x_init'):
            setattr(self, k, d[k])
        for which in ('var', 'par', 'con'):
            layout = {}
            for key, (off, r, c) in zip(d[which + '_names'], d[which + '_dims']):
                label, name = str(key).split('\t')
                layout[(label, name)] = (int(off), int(r), int(c))
            setattr(self, which + '_layout', layout)
        return self


def father_init_vector(father, attr, layout):
    out = np.zeros(sum(r * c for _, r, c in layout.values()))
    for label, child in father.children.items():
        for name in getattr(child, attr):
            off, r, c = layout[(label, name)]
            out[off:off + r * c] = np.asarray(child._values[name], float).reshape(-1, order='F')
    return out
