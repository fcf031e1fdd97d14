"""Formation ADMM: x-update template, z/lambda/residual data model and the
batched iteration driver.

Behavioural spec: reference `problems/admm.py` (construct_upd_x 63-115,
construct_upd_z 117-168, construct_upd_l 248-268, construct_upd_res 270-307,
init_var_admm 360-370, communicate 468-475, init_step 477-491,
ADMMProblem.dual_update 584-628), `problems/formation.py` (construct 33-66),
`problems/dualmethod.py` (options 200-203, initialize/solve 209-224) and
`problems/distributedproblem.py` (interprete_constraints 105-169: the q_i / q_ij /
q_ji index sets; for a formation these are the `fleet_center` coefficient blocks).

Design: every agent of a fleet with the same vehicle type and
number of neighbours shares ONE x-update template; the per-agent consensus state
z_i, l_i, z_ji, l_ji lives *inside* the parameter matrix p [B, n_par] (they are
parameters of the x-update NLP), x_j / z_ij / l_ij in side arrays.  One iteration =
batched x-update (`omgx_batch_solve`) -> neighbour gather -> closed-form z-update
(constant projector, see `zupdate_matrices`) -> lambda update -> residuals ->
neighbour gather.  Across GPUs agents are sharded contiguously; only the boundary
agents' [x_i, z_ij, l_ij] cross ranks (halo) plus one all-reduce of the three
residual sums.
"""
import numpy as np

from .opti import OptiChild, OptiFather
from .problems import FixedTPoint2point
from .splines import shift_knot1_fwd, shiftfirstknot_T, shiftoverknot_T
from .symbolic import Poly


class ADMMUpdater(OptiChild):
    """Owner of the ADMM parameters of one agent's x-update (`admm.py:63-72`)."""

    def __init__(self):
        OptiChild.__init__(self, 'admm')

    def construct(self, center, n_nghb, t0):
        basis = center[0].basis
        L, n_dim = len(basis), len(center)
        ns = L * n_dim
        z_i = np.atleast_1d(self.define_parameter('z_i', ns))
        z_ji = np.atleast_1d(self.define_parameter('z_ji', n_nghb * ns))
        l_i = np.atleast_1d(self.define_parameter('l_i', ns))
        l_ji = np.atleast_1d(self.define_parameter('l_ji', n_nghb * ns))
        rho = self.define_parameter('rho')

        def fwd(vec):                      # only the future piece of each spline is penalised
            return shift_knot1_fwd(vec, basis, t0)
        obj = Poly()
        for k in range(n_dim):
            x = fwd(np.asarray(center[k].coeffs, dtype=object))
            pairs = [(z_i[k * L:(k + 1) * L], l_i[k * L:(k + 1) * L])]
            for j in range(n_nghb):
                o = j * ns + k * L
                pairs.append((z_ji[o:o + L], l_ji[o:o + L]))
            for z, l in pairs:
                z, l = fwd(z), fwd(l)
                for q in range(L):
                    diff = x[q] - z[q]
                    obj = obj + l[q] * diff + 0.5 * rho * diff * diff
        self.define_objective(obj)


def build_updx_template(vehicle, environment, n_nghb, options=None):
    """The x-update NLP of one formation agent as an `NLPTemplate`
    (children in the reference's order `[vehicle, problem, environment, admm] +
    obstacles`, `problems/dualmethod.py:52-53`)."""
    import omgtools.backend as be
    opts = {'verbose': 0}
    opts.update(options or {})
    problem = FixedTPoint2point(vehicle, environment, opts)
    updater = ADMMUpdater()
    father = OptiFather([vehicle, problem, environment, updater] + environment.obstacles)
    problem.father = father
    with father.table:
        rel_pos_c = np.atleast_1d(vehicle.define_parameter('rel_pos_c', vehicle.n_dim))
        problem.construct()
        center = vehicle.get_fleet_center(vehicle.splines[0], rel_pos_c, substitute=True)
        updater.construct(center, n_nghb, problem.t0)
        saved = be.create_nlp
        be.create_nlp = lambda tpl, opt, name='': (None, 0.)   # the batch solver is created by the caller
        try:
            father.construct_problem(opts)
        finally:
            be.create_nlp = saved
    father.init_transformations(problem.init_primal_transform, problem.init_dual_transform)
    return problem, updater, father


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
    M = blkdiag(T_bwd) (I - A'(AA')^{-1}A) blkdiag(T_fwd)  and F = blkdiag(T_fwd)
    (`admm.py:144-162`: f=-(l+rho x), G=-AA'/rho, mu=G^{-1}h, z=-(A'mu+f)/rho with
    b = 0, then the backward knot transform)."""
    L, d = len(basis), basis.degree
    P_term = [basis.derivative(o)[1][-1, :] for o in range(1, d + 1)]
    A = coupling_matrix(L, n_dim, d, n_nghb, P_term)
    Tf, Tb = shiftfirstknot_T(basis, t0, inverse=True)
    nb = n_dim * (1 + n_nghb)
    F = np.kron(np.eye(nb), Tf)
    Bk = np.kron(np.eye(nb), Tb)
    Pi = np.eye(A.shape[1]) - A.T @ np.linalg.solve(A @ A.T, A)
    return Bk @ Pi @ F, F


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
