"""Formation ADMM on the CPU: closed-form z-update against the reference's
formulas and against a least-squares solution of the same equality QP (K10), the
BatchADMM driver against the monolithic oracle iteration, and world_size-2 gloo
sharding with halo exchange against the single-process run."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _scenario(n, perturb=True):
    from omgtools.scenarios import formation_holonomic
    problem, updater, father, lay, P = formation_holonomic(n, with_obstacles=False)
    if perturb:                                  # break the formation so that consensus has work to do
        rng = np.random.default_rng(7)
        d = rng.normal(scale=0.15, size=(n, 2))
        P['p'][:, lay.p_state0:lay.p_state0 + 2] += d
        L = lay.L
        for b in range(n):
            s = P['p'][b, lay.p_state0:lay.p_state0 + 2]
            g = P['p'][b, lay.p_poseT:lay.p_poseT + 2]
            P['x0'][b, lay.x_spl:lay.x_spl + 2 * L] = np.c_[np.linspace(s[0], g[0], L),
                                                            np.linspace(s[1], g[1], L)].reshape(-1, order='F')
    return father.template, lay, P


def test_zupdate_closed_form_is_the_qp_solution():
    from omgtools.formation import zupdate_matrices, coupling_matrix
    from omgtools.splines import BSplineBasis
    K, d, nd, nn = 10, 3, 2, 2
    basis = BSplineBasis(np.r_[np.zeros(d), np.linspace(0, 1, K + 1), np.ones(d)], d)
    L = len(basis)
    rng = np.random.default_rng(0)
    x, l, rho, t0 = rng.normal(size=78), rng.normal(size=78), 1.7, 0.04
    M, F = zupdate_matrices(basis, nd, nn, t0)
    z = M @ (x + l / rho)
    A = coupling_matrix(L, nd, d, nn, [basis.derivative(o)[1][-1, :] for o in range(1, d + 1)])
    assert A.shape == (58, 78)
    zt = F @ z
    assert np.abs(A @ zt).max() < 1e-10                       # feasible for the coupling constraints
    # minimiser of  -l~'z~ + rho/2 |x~ - z~|^2  s.t. A z~ = 0: gradient is in the row space of A
    grad = -(F @ l) - rho * (F @ x - zt)
    coef = np.linalg.lstsq(A.T, grad, rcond=None)[0]
    assert np.abs(A.T @ coef - grad).max() < 1e-9


def test_driver_matches_oracle_iteration():
    from omgtools.admm import BatchADMM
    from omgtools.formation import reverse_slots, coupling_matrix
    from oracle import admm_numpy, port_binding
    from admm_numpy_ops import NumpyAdmmOps
    tpl, lay, P = _scenario(5)
    nbr = P['nbr']
    ops = NumpyAdmmOps(tpl, lay, P['p'], P['x0'], warm=False)     # cold x-updates like the oracle's solve_x below
    admm = BatchADMM(lay, nbr, ops, rho=1.0)
    admm.initialize()
    basis = lay.basis
    A = coupling_matrix(lay.L, 2, 3, 2, [basis.derivative(o)[1][-1, :] for o in range(1, 4)])
    st = admm_numpy.init_state(P['x0'], P['p'], lay, nbr)
    slot = reverse_slots(nbr)

    def solve_x(p, x):
        r = port_binding.solve(tpl, p, x, tol=1e-6, max_iter=300)
        return r['x'], r['status']
    for it in range(3):
        status, (pr, dr, cr) = admm.iterate(0.0)
        st, (pr2, dr2, cr2), _ = admm_numpy.admm_iteration(st, lay, nbr, slot, 1.0, 0.0, A, solve_x)
        assert np.all(status == 0)
        assert abs(pr - np.sqrt(pr2)) < 1e-8 and abs(dr - np.sqrt(dr2)) < 1e-8
        assert np.abs(ops.x - st['x']).max() < 1e-8
        assert np.abs(ops.z_ij - st['z_ij']).max() < 1e-9
    assert admm.residuals[-1][0] < admm.residuals[0][0]       # consensus is being reached


def _ring_groups(sizes):
    """Neighbour table of disjoint rings: agent i is coupled to its two ring neighbours."""
    rows, off = [], 0
    for m in sizes:
        rows += [[off + (k + 1) % m, off + (k - 1) % m] for k in range(m)]
        off += m
    return np.array(rows, dtype=np.int32)


def _worker(rank, world, port, q, n=6, sizes=None, kw=None):
    sys.path.insert(0, os.path.join(ROOT, 'omg-tools_amd'))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from omgtools.admm import BatchADMM
    from omgtools.distributed import shard_range, gather_solutions
    from admm_numpy_ops import NumpyAdmmOps
    tpl, lay, P = _scenario(n)
    nbr = P['nbr'] if sizes is None else _ring_groups(sizes)
    lo, hi = shard_range(n, rank, world)
    ops = NumpyAdmmOps(tpl, lay, P['p'][lo:hi], P['x0'][lo:hi])
    admm = BatchADMM(lay, nbr, ops, rank=rank, world=world, dist=dist, rho=1.0, **(kw or {}))
    admm.initialize()
    counts = []
    for _ in range(4 if kw else 3):
        l0, c0 = ops.launches, ops.collectives
        admm.iterate(0.0, sync=False)                         # nothing comes to the host inside the loop
        counts.append((ops.launches - l0, ops.collectives - c0, bool(ops.fused)))
    x_all = gather_solutions(ops.x, n, dist=dist)
    if rank == 0:
        q.put((admm.residuals, x_all, counts))
    dist.destroy_process_group()


def _sharded_vs_single(world, n=6, sizes=None, kw=None, port_off=0, exact=False):
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from omgtools.admm import BatchADMM
    from admm_numpy_ops import NumpyAdmmOps
    tpl, lay, P = _scenario(n)
    nbr = P['nbr'] if sizes is None else _ring_groups(sizes)
    ops = NumpyAdmmOps(tpl, lay, P['p'], P['x0'])
    ref = BatchADMM(lay, nbr, ops, rho=1.0, **(kw or {}))
    ref.initialize()
    for _ in range(4 if kw else 3):
        ref.iterate(0.0)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29600 + port_off + os.getpid() % 1000
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, n, sizes, kw)) for r in range(world)]
    for pr in procs:
        pr.start()
    try:
        residuals, x_all, counts = q.get(timeout=240)
    finally:
        for pr in procs:
            pr.join(timeout=60)
            if pr.is_alive():
                pr.terminate()
    assert all(pr.exitcode == 0 for pr in procs)
    assert np.allclose(np.array(residuals), np.array(ref.residuals), rtol=1e-9, atol=1e-12)
    assert np.abs(x_all - ops.x).max() < 1e-9
    if exact:       # every agent's numbers do not depend on the rank that holds it (only the fleet sums are added up in another order)
        assert np.array_equal(x_all, ops.x)
    if not kw:
        # the sharded iteration without acceleration: x-update, centre (+ published rows), z / lambda update (+ published
        # rows and residual sums), neighbour read-back = four launches around two all_gathers, nothing else
        assert all(c == (4, 2, True) for c in counts), counts
    return ref


def test_two_rank_halo_exchange_matches_single_process():
    from omgtools.admm import HaloPlan
    nbr = _scenario(6)[2]['nbr']
    halo = HaloPlan(nbr, 1, 2)
    assert (halo.lo, halo.hi) == (3, 6) and halo.needed == [0, 2] and halo.any_halo
    _sharded_vs_single(2)


def test_eight_ranks_of_64_agents_like_baseline_config_4():
    """BASELINE.json configs[3]: 512 agents in formation sharded over the 8 GPUs of a node, 64 per rank -- here 8 gloo
    ranks on the host.  Every rank has exactly two boundary agents whose rows cross ranks; one sharded iteration is
    four launches around two all_gathers, and the fleet's x equals the one-rank fleet's bit for bit."""
    from omgtools.admm import HaloPlan
    from omgtools.consensus import circular_neighbors
    nbr = circular_neighbors(512)
    for r in (0, 3, 7):
        h = HaloPlan(nbr, r, 8)
        assert (h.lo, h.hi) == (64 * r, 64 * r + 64) and len(h.needed) == 2 and len(h.publish_local) == 2 and h.any_halo
    _sharded_vs_single(8, n=512, port_off=3000, exact=True)


def test_rank_without_halo_needs_still_joins_the_collectives():
    """Agents 0..5 form one ring over ranks 0 and 1, agents 6..8 a ring of their own on rank 2: rank 2 needs
    nothing and publishes nothing, but the exchange is a collective of all three ranks (a rank that skips
    it leaves the others waiting; its residual sums also travel in it)."""
    from omgtools.admm import HaloPlan
    nbr = _ring_groups([6, 3])
    halos = [HaloPlan(nbr, r, 3) for r in range(3)]
    assert halos[2].needed == [] and len(halos[2].publish_local) == 0
    assert halos[0].needed and all(h.any_halo for h in halos)
    _sharded_vs_single(3, n=9, sizes=[6, 3], port_off=1000)


def test_nesterov_acceleration_sharded_matches_single_process():
    """`problems/admm.py:510-554` with reset: the previous z_ij, l_ij ride in the second exchange and every rank
    extrapolates the rows it received with the fleet-wide alpha / reset decision."""
    kw = dict(nesterov_acceleration=True, nesterov_reset=True, eta=0.999)
    ref = _sharded_vs_single(2, kw=kw, port_off=2000)
    assert ref.ops.alpha >= 1.0


def test_nesterov_statements_against_the_reference_formulas():
    """One accelerate() of the ops against the reference's statements written out on plain arrays."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from omgtools.admm import BatchADMM
    from admm_numpy_ops import NumpyAdmmOps
    tpl, lay, P = _scenario(5)
    ops = NumpyAdmmOps(tpl, lay, P['p'], P['x0'])
    admm = BatchADMM(lay, P['nbr'], ops, rho=1.0, nesterov_acceleration=True, nesterov_reset=False)
    admm.initialize()
    admm.iterate(0.0)                                           # alpha 1 -> (1 + sqrt 5) / 2, weight 0
    ns = lay.ns
    z_p, l_p = ops.p[:, lay.p_zi:lay.p_zi + ns].copy(), ops.p[:, lay.p_li:lay.p_li + ns].copy()
    zij_p, lij_p = ops.z_ij.copy(), ops.l_ij.copy()
    a1 = ops.alpha
    assert abs(a1 - 0.5 * (1 + np.sqrt(5.))) < 1e-15
    # second iteration: redo the plain update on a copy, then extrapolate by hand
    plain = NumpyAdmmOps(tpl, lay, ops.p.copy(), ops.x.copy())
    plain.z_ij, plain.l_ij = zij_p.copy(), lij_p.copy()
    plain.lam, plain.status, plain.dw = ops.lam.copy(), ops.status.copy(), ops.dw.copy()
    ref = BatchADMM(lay, P['nbr'], plain, rho=1.0)
    ref.iterate(0.0)
    admm.iterate(0.0)
    a2 = 0.5 * (1 + np.sqrt(1 + 4 * a1 ** 2))
    w = (a1 - 1) / a2
    z_new, l_new = plain.p[:, lay.p_zi:lay.p_zi + ns], plain.p[:, lay.p_li:lay.p_li + ns]
    assert np.allclose(ops.p[:, lay.p_zi:lay.p_zi + ns], z_new + w * (z_new - z_p), atol=1e-13)
    assert np.allclose(ops.p[:, lay.p_li:lay.p_li + ns], l_new + w * (l_new - l_p), atol=1e-13)
    assert np.allclose(ops.z_ij, plain.z_ij + w * (plain.z_ij - zij_p), atol=1e-13)
    assert np.allclose(ops.l_ij, plain.l_ij + w * (plain.l_ij - lij_p), atol=1e-13)
    assert abs(ops.alpha - a2) < 1e-15


def _exchange_worker(rank, world, port, q):
    """The device path's exchange (HipAdmmOps.exchange / bind: torch tensors, all_gather_into_tensor) on CPU
    tensors over gloo -- the index logic of the multi-GPU path without a GPU."""
    sys.path.insert(0, os.path.join(ROOT, 'omg-tools_amd'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from omgtools.admm import HaloPlan, HipAdmmOps
    nbr = _ring_groups([7, 3])                     # 10 agents over 3 ranks: uneven shards, one self-contained group
    halo = HaloPlan(nbr, rank, world)
    ops = HipAdmmOps.__new__(HipAdmmOps)           # only the exchange plumbing: no solver, no device
    ops.torch, ops.dev = torch, torch.device('cpu')
    ops.bind(halo, np.zeros_like(halo.nbr_local))
    Bl, w = halo.hi - halo.lo, 5
    rows = torch.arange(halo.lo, halo.hi, dtype=torch.float64)[:, None] * 100. + torch.arange(w, dtype=torch.float64)[None, :]
    extra = torch.tensor([1.0 + rank, 10.0 * (rank + 1), 0.5], dtype=torch.float64)
    out, summed = ops.exchange(rows, halo, dist, extra)
    # every row this rank's agents refer to is the row of that global agent
    want = torch.tensor([[g * 100. + k for k in range(w)] for g in list(range(halo.lo, halo.hi)) + halo.needed], dtype=torch.float64)
    ok = bool(torch.equal(out, want)) and bool(torch.allclose(summed, torch.tensor([6., 60., 1.5], dtype=torch.float64)))
    idx = torch.as_tensor(halo.nbr_local, dtype=torch.int64)
    ok = ok and bool(torch.equal(out[idx][:, :, 0] / 100., torch.as_tensor(nbr[halo.lo:halo.hi], dtype=torch.float64)))
    out2, none = ops.exchange(rows, halo, dist, None)
    ok = ok and none is None and bool(torch.equal(out2, want))
    q.put((rank, ok))
    dist.destroy_process_group()


def test_device_path_exchange_indexing_over_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29600 + 3000 + os.getpid() % 1000
    procs = [ctx.Process(target=_exchange_worker, args=(r, 3, port, q)) for r in range(3)]
    for pr in procs:
        pr.start()
    got = dict(q.get(timeout=120) for _ in range(3))
    for pr in procs:
        pr.join(timeout=60)
    assert all(pr.exitcode == 0 for pr in procs) and got == {0: True, 1: True, 2: True}


def test_formation_mpc_protocol_on_the_host():
    """The receding-horizon protocol of the formation loop (`problems/dualmethod.py:200-224`, `admm.py:477-491`) as
    `FormationMPC` drives it -- prediction, moving obstacle, knot-crossing shift of x and of the consensus state, one ADMM
    iteration per update -- on the numpy backend: the x-updates do real work (the consensus moves with the fleet),
    every one converges, the fleet advances in formation and the primal residual stays small across a crossing."""
    import omgtools.backend as be
    from omgtools import scenarios
    from omgtools.admm import BatchADMM, FormationMPC
    from admm_numpy_ops import NumpyAdmmOps
    saved = be.create_nlp
    be.create_nlp = lambda tpl, opt, name='': (None, 0.)
    try:
        problem, updater, father, lay, P = scenarios.formation_holonomic(6)
    finally:
        be.create_nlp = saved
    tpl = father.template
    ops = NumpyAdmmOps(tpl, lay, P['p'], P['x0'], tol=1e-6)
    admm = BatchADMM(lay, P['nbr'], ops, rho=1.0)
    moving = []
    for obs in problem.environment.obstacles:
        ox, ov, oa = (tpl.entry_range(obs.label, nm, 'par') for nm in ('x', 'v', 'a'))
        if np.any(P['p'][:, ov[0]:ov[1]] != 0.):
            moving.append((ox[0], ov[0], oa[0], ox[1] - ox[0]))
    assert len(moving) == 1
    mpc = FormationMPC(admm, father, tpl, lay, problem.vehicles[0], obstacles=moving, update_time=0.1, init_iter=5,
                       knot_time=problem.knot_time)
    mpc.initialize()
    x_obst0 = ops.p[0, moving[0][0]]
    start = ops.p[:, lay.p_state0:lay.p_state0 + 2].copy()
    crossings = 0
    for k in range(12):
        status, crossed = mpc.step()
        crossings += crossed
        assert np.all(np.asarray(status) == 0), k
    assert crossings == 1
    assert abs(ops.p[0, moving[0][0]] - (x_obst0 - 0.15 * 1.2)) < 1e-12          # the circle moved on
    moved = ops.p[:, lay.p_state0:lay.p_state0 + 2] - start
    assert np.all(moved[:, 1] > 0.2)                                            # 1.2 s towards the goal (+y)
    assert np.abs(moved - moved.mean(axis=0)).max() < 0.05                        # ... in formation
    res = admm.residuals
    # one iteration per update keeps the consensus converging while the fleet moves (and through the crossing)
    assert len(res) == 5 + 12 and res[-1][0] < res[4][0] and max(r[0] for r in res[5:]) < 2.5 * res[4][0], [r[0] for r in res]


# ---- interconnection = 'full': one all_reduce per iteration ------------------------------------------------------
def _full_scenario(n, interconnection):
    from omgtools.scenarios import formation_holonomic
    problem, updater, father, lay, P = formation_holonomic(n, with_obstacles=False, interconnection=interconnection)
    rng = np.random.default_rng(11)
    d = rng.normal(scale=0.15, size=(n, 2))
    P['p'][:, lay.p_state0:lay.p_state0 + 2] += d
    L = lay.L
    for b in range(n):
        s_, g_ = P['p'][b, lay.p_state0:lay.p_state0 + 2], P['p'][b, lay.p_poseT:lay.p_poseT + 2]
        P['x0'][b, lay.x_spl:lay.x_spl + 2 * L] = np.c_[np.linspace(s_[0], g_[0], L), np.linspace(s_[1], g_[1], L)].reshape(-1, order='F')
    return father.template, lay, P


def test_full_interconnection_collapses_to_one_all_reduce():
    """`vehicles/fleet.py:55-56` interconnection = 'full': the general iteration (`BatchADMM`, x-update template with N - 1
    neighbour blocks, z-update projector of order N n_shared) and the fused form (`FullConsensusADMM`: the usual template,
    one sum over the fleet per iteration) produce the same iterates and residuals."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from omgtools.admm import BatchADMM, FullConsensusADMM
    from admm_numpy_ops import NumpyAdmmOps
    n = 4
    tpl_g, lay_g, P_g = _full_scenario(n, 'full')
    assert lay_g.n_nghb == 3 and P_g['nbr'].shape == (4, 3)
    ops_g = NumpyAdmmOps(tpl_g, lay_g, P_g['p'], P_g['x0'])
    gen = BatchADMM(lay_g, P_g['nbr'], ops_g, rho=1.0)
    gen.initialize()
    tpl_f, lay_f, P_f = _full_scenario(n, 'circular')          # the usual template (two neighbour blocks)
    ops_f = NumpyAdmmOps(tpl_f, lay_f, P_f['p'], P_f['x0'])
    fus = FullConsensusADMM(lay_f, ops_f, n, rho=1.0)
    fus.initialize()
    for it in range(5):
        st_g, _ = gen.iterate(0.0)
        st_f = fus.iterate(0.0)
        assert np.all(st_g == 0) and np.all(st_f == 0)
        xg = ops_g.x[:, lay_g.x_spl:lay_g.x_spl + lay_g.ns]
        xf = ops_f.x[:, lay_f.x_spl:lay_f.x_spl + lay_f.ns]
        assert np.abs(xg - xf).max() < 2e-6, (it, np.abs(xg - xf).max())       # (x-updates at 1e-6)
        # every copy of the general iteration holds the fused iteration's c
        z_i = ops_g.p[:, lay_g.p_zi:lay_g.p_zi + lay_g.ns]
        assert np.abs(z_i - fus.c[None]).max() < 2e-6 and np.abs(ops_g.z_ij - fus.c[None, None]).max() < 2e-6
    assert np.allclose(np.array(fus.residuals), np.array(gen.residuals), rtol=2e-4, atol=1e-7)
    assert fus.residuals[-1][0] < 0.5 * fus.residuals[0][0]


def _full_worker(rank, world, port, q, n):
    sys.path.insert(0, os.path.join(ROOT, 'omg-tools_amd')); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from omgtools.admm import FullConsensusADMM
    from omgtools.distributed import shard_range, gather_solutions
    from admm_numpy_ops import NumpyAdmmOps
    tpl, lay, P = _full_scenario(n, 'circular')
    lo, hi = shard_range(n, rank, world)
    ops = NumpyAdmmOps(tpl, lay, P['p'][lo:hi], P['x0'][lo:hi])
    admm = FullConsensusADMM(lay, ops, n, rank=rank, world=world, dist=dist, rho=1.0)
    admm.initialize()
    for _ in range(4):
        admm.iterate(0.0)
    res = admm.residuals
    x_all = gather_solutions(ops.x, n, dist=dist)
    if rank == 0:
        q.put((res, x_all, admm.collectives))
    dist.destroy_process_group()


def test_full_interconnection_sharded_is_one_collective_per_iteration():
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from omgtools.admm import FullConsensusADMM
    from admm_numpy_ops import NumpyAdmmOps
    n, world = 8, 2
    tpl, lay, P = _full_scenario(n, 'circular')
    ops = NumpyAdmmOps(tpl, lay, P['p'], P['x0'])
    ref = FullConsensusADMM(lay, ops, n, rho=1.0)
    ref.initialize()
    for _ in range(4):
        ref.iterate(0.0)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29600 + 4000 + os.getpid() % 1000
    procs = [ctx.Process(target=_full_worker, args=(r, world, port, q, n)) for r in range(world)]
    for pr in procs:
        pr.start()
    try:
        res, x_all, collectives = q.get(timeout=240)
    finally:
        for pr in procs:
            pr.join(timeout=60)
            if pr.is_alive():
                pr.terminate()
    assert all(pr.exitcode == 0 for pr in procs)
    assert collectives == 4                                    # one all_reduce of n_shared + 3 doubles per iteration
    assert np.abs(x_all - ops.x).max() < 1e-7
    assert np.allclose(np.array(res), np.array(ref.residuals), rtol=1e-6, atol=1e-9)
