"""ORACLE (test infrastructure, never shipped or measured as product).

Numpy restatement of one ADMM iteration of omg-tools' FormationPoint2point,
written directly from the reference's formulas:
  z-update   `problems/admm.py:117-168` (f = -(l + rho x~), G = -A A'/rho,
             h = b + A f/rho, mu = G^{-1} h, z~ = -(A' mu + f)/rho, back-shift),
             with A from `problems/formation.py:46-65`;
  l-update   `admm.py:248-268`  l <- l + rho (x - z)   (untransformed);
  residuals  `admm.py:270-307`  (as executed: on the stored coefficients);
  exchange   `admm.py:468-475`.
The x-update NLP is solved by whatever `solve_x` callable the test passes (the
CPU port, or the HIP kernel under test).  Parity unpinned against IPOPT for the
x-update (see oracle/ipm_numpy.py); the closed-form updates are exact algebra.
"""
import numpy as np


def shiftfirstknot_T(knots, degree, n, t_shift):
    """Independent construction by least squares on the future piece [t_shift, 1]
    (cf. `spline_extra.py:220-255`)."""
    from oracle.nlp_numpy import eval_basis_row
    k2 = np.array(knots, float)
    k2[:degree + 1] = t_shift
    x = np.linspace(t_shift, 1.0, 400)[1:]
    B1 = np.array([eval_basis_row(knots, degree, u) for u in x])
    B2 = np.array([eval_basis_row(k2, degree, u) for u in x])
    T = np.linalg.lstsq(B2, B1, rcond=None)[0]
    T[np.abs(T) < 1e-10] = 0.
    T[degree + 1:] = np.eye(n)[degree + 1:]
    return T


def admm_iteration(state, lay, nbr, slot, rho, t0, A, solve_x):
    """state: dict of arrays x [B,n_var], p [B,n_par], x_i, z_i, l_i [B,ns],
    x_j, z_ij, l_ij [B,nn,ns]; returns (state, (pr, dr, cr))."""
    B, nn, ns, L, nd = state['x'].shape[0], lay.n_nghb, lay.ns, lay.L, lay.n_dim
    p = state['p']
    # parameters of the x-update: consensus state of the previous iteration
    p[:, lay.p_zi:lay.p_zi + ns] = state['z_i']
    p[:, lay.p_li:lay.p_li + ns] = state['l_i']
    p[:, lay.p_zji:lay.p_zji + nn * ns] = state['z_ji'].reshape(B, -1)
    p[:, lay.p_lji:lay.p_lji + nn * ns] = state['l_ji'].reshape(B, -1)
    p[:, lay.p_rho] = rho
    # x-update
    state['x'], status = solve_x(p, state['x'])
    c = state['x'][:, lay.x_spl:lay.x_spl + ns].reshape(B, nd, L)
    rel = p[:, lay.p_rel:lay.p_rel + nd]
    x_i = (c + rel[:, :, None]).reshape(B, ns)
    x_j = x_i[nbr]                                      # communicate #1
    # z-update (closed form) on forward-shifted coefficients
    knots, deg = lay.basis.knots, lay.basis.degree
    Tf = shiftfirstknot_T(knots, deg, L, t0)
    Tb = np.linalg.inv(Tf)
    blk = nd * (1 + nn)
    F, Bk = np.kron(np.eye(blk), Tf), np.kron(np.eye(blk), Tb)
    x_all = np.concatenate([x_i[:, None, :], x_j], axis=1).reshape(B, -1)
    l_all = np.concatenate([state['l_i'][:, None, :], state['l_ij']], axis=1).reshape(B, -1)
    z_prev = np.concatenate([state['z_i'][:, None, :], state['z_ij']], axis=1).reshape(B, -1)
    xt, lt = x_all @ F.T, l_all @ F.T
    f = -(lt + rho * xt)
    G = -(A @ A.T) / rho
    h = (f @ A.T) / rho
    mu = np.linalg.solve(G, h.T).T
    zt = -(mu @ A + f) / rho
    z_all = zt @ Bk.T
    # l-update, residuals
    l_all = l_all + rho * (x_all - z_all)
    # (as executed the reference measures the residuals on the untransformed coefficients: the transformed
    # structs of `admm.py:286-289` are dropped, see tests/test_golden_admm.py)
    pr = ((x_all - z_all) ** 2).sum()
    dr = rho * ((z_all - z_prev) ** 2).sum()
    cr = rho * pr + dr
    z_all, l_all = z_all.reshape(B, 1 + nn, ns), l_all.reshape(B, 1 + nn, ns)
    state.update(x_i=x_i, x_j=x_j, z_i=z_all[:, 0], z_ij=z_all[:, 1:], l_i=l_all[:, 0], l_ij=l_all[:, 1:])
    # communicate #2: what my neighbours think of me
    state['z_ji'] = np.stack([state['z_ij'][nbr[:, k], slot[:, k]] for k in range(nn)], axis=1)
    state['l_ji'] = np.stack([state['l_ij'][nbr[:, k], slot[:, k]] for k in range(nn)], axis=1)
    return state, (pr, dr, cr), status


def init_state(x0, p, lay, nbr):
    """`admm.py:360-370`: z_i = x_i, z_ji = x_i of the LOCAL agent, multipliers 0."""
    B, nn, ns, L, nd = x0.shape[0], lay.n_nghb, lay.ns, lay.L, lay.n_dim
    c = x0[:, lay.x_spl:lay.x_spl + ns].reshape(B, nd, L)
    x_i = (c + p[:, lay.p_rel:lay.p_rel + nd][:, :, None]).reshape(B, ns)
    z = np.zeros((B, nn, ns))
    return dict(x=x0.copy(), p=p.copy(), x_i=x_i, z_i=x_i.copy(), l_i=np.zeros((B, ns)),
                x_j=z.copy(), z_ij=z.copy(), l_ij=z.copy(),
                z_ji=np.repeat(x_i[:, None, :], nn, axis=1), l_ji=z.copy())
