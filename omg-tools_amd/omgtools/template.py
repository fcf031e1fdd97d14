"""`NLPTemplate`: the flat, numeric description of one agent's NLP.

This replaces the CasADi expression graphs the reference hands to `nlpsol`
(`basics/optilayer.py:49-60, 180-198`).  Everything the device needs to
evaluate f, g, their Jacobian and the Lagrangian Hessian for *any* parameter
vector p is a handful of integer/float arrays:

  atoms      a = [p (n_par) | derived atoms]; derived atoms come from a
             straight-line program (`prog`): DIV(ppoly, ppoly) and BSPL(basis, u)
  ppolys     polynomials over atoms (CSR over monomials over atom indices)
  slots      per-agent scalars: slot s = ppoly[slot_pp[s]](a)
  terms      row r:  g_r(x) = sum_t  coef_t * S_t * prod_{v in vars_t} x_v,
             S_t = 1 if slot_t < 0 else slot value; <= 3 variables per term;
             row n_con holds the objective.

The x / p / g orderings are the reference's (`optilayer.py:225-272`).
"""
import numpy as np

from .symbolic import Poly, is_atom, _ATOM_BASE

OP_DIV, OP_BSPL = 0, 1
MAX_TERM_VARS = 3


class NLPTemplate(object):

    # ------------------------------------------------------------------ build
    @classmethod
    def from_father(cls, father):
        self = cls()
        table = father.table
        self.var_layout = father._var_layout
        self.par_layout = father._par_layout
        self.con_layout = father._con_layout
        self.n_var = sum(r * c for _, r, c in self.var_layout.values())
        self.n_par = sum(r * c for _, r, c in self.par_layout.values())
        self.n_con = sum(r * c for _, r, c in self.con_layout.values())
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
        n_atoms = self.n_par
        for entry in table.derived:
            if entry[0] == 'div':
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
            else:
                _, kn, deg, u_sym, first = entry
                prog.append((OP_BSPL, len(knots), len(kn), deg,
                             self._atom(u_sym), atom_index[first]))
                knots.extend(kn.tolist())
        self.prog = np.array(prog, dtype=np.int32).reshape(-1, 6)
        self.knots = np.array(knots, dtype=np.float64)

        # rows -> terms
        self._slot_keys, slot_pp = {}, []
        rows = []
        for label, child in father.children.items():
            for cname, (exprs, _, _, _) in child._constraints.items():
                rows.extend(list(exprs))
        assert len(rows) == self.n_con
        objective = Poly()
        for child in father.children.values():
            objective = objective + child._objective
        rows.append(objective)

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

        # default values
        self.x_init = father_init_vector(father, '_variables', self.var_layout)
        return self

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


def father_init_vector(father, attr, layout):
    out = np.zeros(sum(r * c for _, r, c in layout.values()))
    for label, child in father.children.items():
        for name in getattr(child, attr):
            off, r, c = layout[(label, name)]
            out[off:off + r * c] = np.asarray(child._values[name], float).reshape(-1, order='F')
    return out


# ---------------------------------------------------------------------------
# Solver plan: static structure the HIP interior-point kernel works from
# ---------------------------------------------------------------------------

class SolverPlan(object):
    """Static, agent-independent structure for the per-agent KKT solve.

    Variables are permuted into  [leaf_0 | leaf_1 | ... | root | t]  where no
    constraint row couples two different leaves (each obstacle's hyperplane
    variables form a leaf, the trajectory/slack coefficients the root), so the
    condensed KKT matrix is block-arrow and is factorised leaf by leaf with a
    Schur complement onto the root.  `t` is the phase-I variable
    the kernel appends.  Equality rows may only touch root variables.
    """
    MIN_LEAF = 8

    def __init__(self, tpl, root_hint=('splines_seg',), gather_small=True):
        n, m = tpl.n_var, tpl.n_con
        self.n_var, self.n_con = n, m
        rows_vars = []
        for r in range(m):
            vs = set()
            for t in range(tpl.row_ptr[r], tpl.row_ptr[r + 1]):
                vs.update(int(v) for v in tpl.t_var[t] if v >= 0)
            rows_vars.append(vs)
        # nonlinear objective terms couple their variables exactly like a constraint row
        obj_couplings = []
        for t in range(tpl.row_ptr[m], tpl.row_ptr[m + 1]):
            vs = set(int(v) for v in tpl.t_var[t] if v >= 0)
            if len(vs) > 1:
                obj_couplings.append(vs)
        eq = np.isfinite(tpl.lb) & (tpl.lb == tpl.ub)
        self.eq_rows = np.nonzero(eq)[0].astype(np.int32)
        self.n_eq = len(self.eq_rows)
        self.eq_index = -np.ones(m, dtype=np.int32)
        self.eq_index[self.eq_rows] = np.arange(self.n_eq)

        root = set()
        for r in self.eq_rows:
            root.update(rows_vars[r])
        for (label, name), (off, rr, cc) in tpl.var_layout.items():
            if any(name.startswith(h) for h in root_hint):
                root.update(range(off, off + rr * cc))
        # connected components of the remaining variables
        parent = list(range(n))

        def find(a):
            while parent[a] != a:
                parent[a] = parent[parent[a]]
                a = parent[a]
            return a
        for vs in rows_vars + obj_couplings:
            rest = [v for v in vs if v not in root]
            for a, b in zip(rest[:-1], rest[1:]):
                parent[find(a)] = find(b)
        comps = {}
        for v in range(n):
            if v not in root:
                comps.setdefault(find(v), []).append(v)
        # Components smaller than MIN_LEAF (e.g. every coefficient of the terminal slacks g*, which
        # meets the rest of the problem through one trajectory coefficient only) are gathered into
        # one extra leaf: its block is block-diagonal, and eliminating it leaf-style keeps those
        # variables out of the dense root factorisation.
        leaves, small = [], []
        for comp in sorted(comps.values(), key=lambda c: c[0]):
            if len(comp) < self.MIN_LEAF:
                small.extend(comp)
            else:
                leaves.append(sorted(comp))
        if len(small) >= self.MIN_LEAF and gather_small:
            leaves.append(sorted(small))
        else:
            root.update(small)
        self.leaves = leaves
        self.n_leaf = len(leaves)
        root_vars = sorted(root) + [n]                 # t last
        order = [v for leaf in leaves for v in leaf] + root_vars
        self.order = np.array(order, dtype=np.int32)   # position -> variable (n = t)
        self.pos = np.empty(n + 1, dtype=np.int32)
        self.pos[self.order] = np.arange(n + 1)
        self.leaf_off = np.cumsum([0] + [len(l) for l in leaves]).astype(np.int32)
        self.root_off = int(self.leaf_off[-1])
        self.n_root = len(root_vars)                   # includes t
        leaf_of = -np.ones(n + 1, dtype=np.int32)
        for l, leaf in enumerate(leaves):
            leaf_of[leaf] = l
        self.leaf_of_var = leaf_of

        # per-row Jacobian structure (permuted positions), term -> entry index
        jr_ptr, jr_pos, t_jidx = [0], [], -np.ones((tpl.n_terms, 3), dtype=np.int32)
        row_leaf = -np.ones(m + 1, dtype=np.int32)
        for r in range(m + 1):
            vs = sorted(rows_vars[r]) if r < m else sorted(
                set(int(v) for t in range(tpl.row_ptr[m], tpl.row_ptr[m + 1])
                    for v in tpl.t_var[t] if v >= 0))
            ls = set(int(leaf_of[v]) for v in vs if leaf_of[v] >= 0)
            if len(ls) > 1 and r < m:
                raise ValueError('row %d couples two leaves; partition invalid' % r)
            if r < m and eq[r] and ls:
                raise ValueError('equality row %d touches a leaf variable' % r)
            row_leaf[r] = ls.pop() if (ls and r < m) else -1
            local = {v: k for k, v in enumerate(sorted(vs, key=lambda v: self.pos[v]))}
            base = len(jr_pos)
            jr_pos.extend(int(self.pos[v]) for v in sorted(vs, key=lambda v: self.pos[v]))
            for t in range(tpl.row_ptr[r], tpl.row_ptr[r + 1]):
                for k in range(3):
                    v = int(tpl.t_var[t, k])
                    if v >= 0:
                        t_jidx[t, k] = base + local[v]
            jr_ptr.append(len(jr_pos))
        for vs in obj_couplings:
            if len(set(int(leaf_of[v]) for v in vs if leaf_of[v] >= 0)) > 1:
                raise ValueError('objective couples two leaves; partition invalid')
        self.jr_ptr = np.array(jr_ptr, dtype=np.int32)
        self.jr_pos = np.array(jr_pos, dtype=np.int32)
        self.t_jidx = t_jidx
        self.row_leaf = row_leaf
        self.nnz_j = len(jr_pos)

        # column structure for deterministic J^T w gathers
        cols = [[] for _ in range(n)]
        for r in range(m + 1):
            for e in range(self.jr_ptr[r], self.jr_ptr[r + 1]):
                cols[self.jr_pos[e]].append((r, e))
        self.jc_ptr = np.cumsum([0] + [len(c) for c in cols]).astype(np.int32)
        self.jc_row = np.array([r for c in cols for (r, e) in c], dtype=np.int32)
        self.jc_ent = np.array([e for c in cols for (r, e) in c], dtype=np.int32)

        # root positions coupled to each leaf (rows of the B_l blocks); t always
        cpl = []
        for l in range(self.n_leaf):
            s = set()
            for r in range(m):
                if row_leaf[r] == l:
                    s.update(int(self.pos[v]) - self.root_off for v in rows_vars[r]
                             if leaf_of[v] < 0)
            s.add(self.n_root - 1)
            cpl.append(sorted(s))
        self.cpl_ptr = np.cumsum([0] + [len(c) for c in cpl]).astype(np.int32)
        self.cpl_idx = np.array([i for c in cpl for i in c], dtype=np.int32)
        # root-local index -> row in B_l (or -1), flattened [n_leaf, n_root]
        self.cpl_map = -np.ones((max(self.n_leaf, 1), self.n_root), dtype=np.int32)
        for l, c in enumerate(cpl):
            self.cpl_map[l, c] = np.arange(len(c))

    def summary(self):
        return dict(n_leaf=self.n_leaf, leaf_sizes=[len(l) for l in self.leaves],
                    n_root=self.n_root, n_eq=self.n_eq, nnz_j=self.nnz_j,
                    cpl=[int(b - a) for a, b in zip(self.cpl_ptr[:-1], self.cpl_ptr[1:])])
