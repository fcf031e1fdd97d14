"""Numeric side of the consensus iteration (no front-end objects): the coupling matrices and closed-form z-update
projectors of the formation and rendez-vous fleets (`problems/admm.py:144-162, 313-354`, `problems/formation.py`,
`problems/rendezvous.py:47-58`), neighbour tables (`vehicles/fleet.py:49-60`), the offsets the ADMM kernels touch inside
x and p, and the knot-crossing shift tables of a fleet (`problems/admm.py:477-491`).  `omgtools.admm` and the workload
bundles (`omgtools.workloads`) need nothing else of `formation.py` / `rendezvous.py`."""
import numpy as np

from .splines import shiftfirstknot_T, shiftoverknot_T


def coupling_matrix(L, n_dim, degree, n_nghb, P_term):
    """A of the z-update's equality constraints (`formation.py:46-65` seen through
    `admm.py:313-354`): z_i - z_ij = 0 for every neighbour, and the terminal
    derivative rows (d^o/dtau^o centre_i)(1) = 0, o = 1..degree.  Unknown vector
    [z_i | z_ij (neighbour by neighbour)], each block axis-major."""
    ns = L * n_dim
    rows = []
    for j in range(n_nghb):
        A = np.zeros((ns, (1 + n_nghb) * ns))
        A[:, :ns] = np.eye(ns)
        A[:, (1 + j) * ns:(2 + j) * ns] = -np.eye(ns)
        rows.append(A)
    for k in range(n_dim):
        for o in range(1, degree + 1):
            a = np.zeros((1, (1 + n_nghb) * ns))
            a[0, k * L:(k + 1) * L] = P_term[o - 1]
            rows.append(a)
    return np.vstack(rows)


def zupdate_matrices(basis, n_dim, n_nghb, t0):
    """(M, F): z_all = M (x_all + l_all/rho) with
    M = blkdiag(T_bwd) (I - A'(AA')^{-1}A) blkdiag(T_fwd)
    (`admm.py:144-162`: f=-(l+rho x), G=-AA'/rho, mu=G^{-1}h, z=-(A'mu+f)/rho with
    b = 0, then the backward knot transform) and F the transform the residuals are measured in.
    As executed the reference applies NO knot transform in the z-update and the residuals: the transformed
    structs `_transform_spline` returns are dropped (`admm.py:143-146, 286-289`, `dualmethod.py:154-157`).
    For M that makes no difference (the coupling rows are differences of whole splines and the terminal
    rows touch the last coefficients only: T_bwd Pi T_fwd = Pi); for the residuals it does, so F = I.
    Pinned by tests/test_golden_admm.py."""
    L, d = len(basis), basis.degree
    P_term = [basis.derivative(o)[1][-1, :] for o in range(1, d + 1)]
    A = coupling_matrix(L, n_dim, d, n_nghb, P_term)
    Tf, Tb = shiftfirstknot_T(basis, t0, inverse=True)
    nb = n_dim * (1 + n_nghb)
    F = np.kron(np.eye(nb), Tf)
    Bk = np.kron(np.eye(nb), Tb)
    Pi = np.eye(A.shape[1]) - A.T @ np.linalg.solve(A @ A.T, A)
    return Bk @ Pi @ F, np.eye(F.shape[0])


class FormationLayout(object):
    """Offsets of everything the ADMM kernels touch inside x and p."""

    def __init__(self, template, vehicle, problem, updater, n_nghb):
        t = template
        self.n_dim, self.L, self.degree = vehicle.n_dim, len(vehicle.basis), vehicle.degree
        self.ns, self.n_nghb = self.n_dim * self.L, n_nghb
        self.x_spl = t.entry_range(vehicle.label, 'splines_seg0', 'var')[0]
        par = lambda child, name: t.entry_range(child.label, name, 'par')[0]
        self.p_rel = par(vehicle, 'rel_pos_c')
        self.p_state0, self.p_input0 = par(vehicle, 'state0'), par(vehicle, 'input0')
        self.p_poseT = par(vehicle, 'poseT')
        self.p_T, self.p_t = par(problem, 'T'), par(problem, 't')
        self.p_zi, self.p_zji = par(updater, 'z_i'), par(updater, 'z_ji')
        self.p_li, self.p_lji = par(updater, 'l_i'), par(updater, 'l_ji')
        self.p_rho = par(updater, 'rho')
        self.basis = vehicle.basis


def circular_neighbors(n):
    """[next, previous] for every agent (`vehicles/fleet.py:49-60`; order of
    `distributedproblem.py:181-182`: next first)."""
    idx = np.arange(n)
    return np.stack([(idx + 1) % n, (idx - 1) % n], axis=1).astype(np.int32)


def reverse_slots(nbr):
    """slot[b, k] = position of b in the neighbour list of nbr[b, k]."""
    B, nn = nbr.shape
    slot = np.zeros((B, nn), dtype=np.int32)
    for b in range(B):
        for k in range(nn):
            j = nbr[b, k]
            slot[b, k] = int(np.nonzero(nbr[j] == b)[0][0])
    return slot


def shift_tables(father, tpl, lay, basis, consensus_is_spline=True):
    """Knot-crossing shift of an ADMM fleet (`problems/admm.py:477-491`): (entries, T matrices) for every spline
    variable of x, for the consensus blocks z_i, l_i, z_ji, l_ji inside p and for the side arrays z_ij / l_ij --
    the three arguments of `HipAdmmOps.shift`."""
    Tm = shiftoverknot_T(basis)
    ents, mats, off = [], [], 0
    for label, name, spl in father.shifted_entries(every_spline=True):
        lo, rows, cols = tpl.var_layout[(label, name)]
        Ts = shiftoverknot_T(spl['basis'])
        ents.append([lo, rows, cols, off]); mats.append(Ts.reshape(-1)); off += Ts.size
    shift_x = (np.array(ents, dtype=np.int32), np.concatenate(mats))
    L, nd, n_nghb = lay.L, lay.n_dim, lay.n_nghb
    if consensus_is_spline:
        p_ents = [[lay.p_zi, L, nd, 0], [lay.p_li, L, nd, 0]]
        for j in range(n_nghb):
            p_ents += [[lay.p_zji + j * lay.ns, L, nd, 0], [lay.p_lji + j * lay.ns, L, nd, 0]]
        shift_p = (np.array(p_ents, dtype=np.int32), Tm.reshape(-1).copy())
        shift_side = (np.array([[j * lay.ns, L, nd, 0] for j in range(n_nghb)], dtype=np.int32), Tm.reshape(-1).copy())
    else:                  # (the shared quantity is a plain vector: nothing of the consensus state is shifted)
        shift_p = (np.zeros((0, 4), dtype=np.int32), np.zeros(0))
        shift_side = (np.zeros((0, 4), dtype=np.int32), np.zeros(0))
    return shift_x, shift_p, shift_side


def consensus_matrix(ns, n_nghb):
    """A of the z-update's equality constraints (`rendezvous.py:47-58` seen through `admm.py:313-354`):
    z_i - z_ij = 0 for every neighbour; unknown vector [z_i | z_ij (neighbour by neighbour)]."""
    A = np.zeros((n_nghb * ns, (1 + n_nghb) * ns))
    for j in range(n_nghb):
        A[j * ns:(j + 1) * ns, :ns] = np.eye(ns)
        A[j * ns:(j + 1) * ns, (1 + j) * ns:(2 + j) * ns] = -np.eye(ns)
    return A


class RendezVousLayout(object):
    """Offsets of everything the ADMM kernels touch inside x and p (same fields as FormationLayout; the shared
    vector is n_dim blocks of L = 1 coefficient)."""

    def __init__(self, template, vehicle, problem, updater, n_nghb):
        t = template
        self.n_dim, self.L, self.degree = vehicle.n_dim, 1, 0
        self.ns, self.n_nghb = self.n_dim, n_nghb
        self.x_spl = t.entry_range(problem.label, 'conT0', 'var')[0]           # what omgx_admm_center reads
        self.x_traj = t.entry_range(vehicle.label, 'splines_seg0', 'var')[0]
        par = lambda child, name: t.entry_range(child.label, name, 'par')[0]
        self.p_rel = par(vehicle, 'rel_pos_c')
        self.p_state0, self.p_input0 = par(vehicle, 'state0'), par(vehicle, 'input0')
        self.p_poseT = par(vehicle, 'poseT')
        self.p_T, self.p_t = par(problem, 'T'), par(problem, 't')
        self.p_zi, self.p_zji = par(updater, 'z_i'), par(updater, 'z_ji')
        self.p_li, self.p_lji = par(updater, 'l_i'), par(updater, 'l_ji')
        self.p_rho = par(updater, 'rho')
        self.basis = vehicle.basis
        A = consensus_matrix(self.ns, n_nghb)
        self._M = np.eye(A.shape[1]) - A.T @ np.linalg.solve(A @ A.T, A)

    def zupdate(self, t0):
        """(M, F) of the closed-form z-update z_all = M (x_all + l_all / rho) (`admm.py:144-162`); nothing depends
        on the time: the shared vector is not a spline."""
        return self._M, np.eye(self._M.shape[0])
