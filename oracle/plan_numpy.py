"""ORACLE (test infrastructure): independent Python statement of the block-arrow partition the library
derives in csrc/omgx_plan.h (which variables form leaves, which the root).  The dense numpy interior
point (oracle/ipm_numpy.py) only needs the leaf membership (inertia-correction classes of cold starts);
tests compare the leaf sets with `omgx_plan_describe`.  Only tests/ and the other oracle modules import
this."""
import numpy as np


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
        # csrc/omgx_plan.h `wave_ok`: every panel fits one wave -- the solver then answers a root block of the wrong
        # inertia by raising the inertia correction of the root variables only
        nr = self.n_root + self.n_eq
        self.wave_ok = nr <= 40 and all(len(leaf) <= 40 and len(leaf) + len(c) - 1 <= 64 and len(c) + 1 <= 32
                                        for leaf, c in zip(self.leaves, cpl))
        self.cpl_map = -np.ones((max(self.n_leaf, 1), self.n_root), dtype=np.int32)
        for l, c in enumerate(cpl):
            self.cpl_map[l, c] = np.arange(len(c))

    def summary(self):
        return dict(n_leaf=self.n_leaf, leaf_sizes=[len(l) for l in self.leaves],
                    n_root=self.n_root, n_eq=self.n_eq, nnz_j=self.nnz_j,
                    cpl=[int(b - a) for a, b in zip(self.cpl_ptr[:-1], self.cpl_ptr[1:])])
