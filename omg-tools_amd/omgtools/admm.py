"""`BatchADMM`: the formation ADMM iteration over a (sharded) fleet.

One iteration (`problems/admm.py:584-611`): x-update (batched NLP solve) ->
exchange x -> z-update, lambda update, residuals -> exchange z/lambda; the
kernels are reached through an `ops` object: `HipAdmmOps` (device tensors +
include/omgx.h) on the GPU, or the numpy oracle in the CPU distributed test.

Sharding: agents are split contiguously over the ranks; a rank
needs the consensus rows of the few remote agents its own agents are neighbours
with (2 for the circular topology).  Every rank publishes the rows other ranks
need.  Two collectives per iteration: one `all_gather` of the published x_i rows after the
x-update, and one `all_gather` of the published [z_ij | l_ij] rows in which every rank's three
residual sums ride along as one more row (reference: plain attribute reads `admm.py:468-475` and
a Python sum `admm.py:601-603`).  Nothing is copied to the host inside the loop: the residual
history stays on the device until somebody asks for it.
"""
import ctypes as C

import os

import numpy as np

from .distributed import shard_range
from .consensus import zupdate_matrices, reverse_slots


class AdmmLayoutC(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('n_dim', 'L', 'n_nghb', 'x_spl', 'p_rel',
                                         'p_zi', 'p_zji', 'p_li', 'p_lji')]


class HaloPlan(object):
    """Which remote agents a rank needs and who publishes them."""

    def __init__(self, nbr_global, rank, world):
        N = nbr_global.shape[0]
        self.rank, self.world = rank, world
        ranges = [shard_range(N, r, world) for r in range(world)]
        self.lo, self.hi = ranges[rank]
        owner = np.zeros(N, dtype=np.int64)
        for r, (a, b) in enumerate(ranges):
            owner[a:b] = r
        needed = []
        for r, (a, b) in enumerate(ranges):
            nb = np.unique(nbr_global[a:b].ravel())
            needed.append([int(g) for g in nb if not (a <= g < b)])
        # publish[r] = agents owned by r that any other rank needs (sorted)
        publish = [sorted(set(g for q in range(world) if q != r for g in needed[q] if owner[g] == r))
                   for r in range(world)]
        self.max_pub = max([len(pp) for pp in publish] + [1])
        # the same on every rank: either all ranks enter the collective of an exchange or none does (a rank
        # whose own agents only have local neighbours must still take part when another rank needs rows)
        self.any_halo = any(len(nd) > 0 for nd in needed)
        self.publish_local = np.array([g - self.lo for g in publish[rank]], dtype=np.int64)
        self.needed = needed[rank]
        # where each needed agent sits in the all_gather result [world, max_pub, width]
        self.src = np.array([[owner[g], publish[owner[g]].index(g)] for g in self.needed],
                            dtype=np.int64).reshape(-1, 2)
        Bl = self.hi - self.lo
        lookup = {g: Bl + i for i, g in enumerate(self.needed)}
        lookup.update({g: g - self.lo for g in range(self.lo, self.hi)})
        self.nbr_local = np.vectorize(lookup.__getitem__)(nbr_global[self.lo:self.hi]).astype(np.int32)
        # the copy-free exchange (`BatchADMM.iterate`, include/omgx.h omgx_admm_*_ex): a buffer holds the local rows on top
        # and the all_gather result -- `rows` rows per rank -- straight behind them; a remote neighbour is addressed where
        # its owner's published row lands.  rows_x = max_pub for the x_i rows, rows_zl = max_pub + 1 for [z_ij | l_ij]
        # (the extra row of every rank carries its residual sums).
        self.rows_x, self.rows_zl = self.max_pub, self.max_pub + 1
        self.pub_slot = np.full(Bl, -1, dtype=np.int32)
        self.pub_slot[self.publish_local] = np.arange(len(self.publish_local), dtype=np.int32)

        def gathered(rows):
            lk = {g: g - self.lo for g in range(self.lo, self.hi)}
            lk.update({g: Bl + int(owner[g]) * rows + publish[owner[g]].index(g) for g in self.needed})
            return np.vectorize(lk.__getitem__)(nbr_global[self.lo:self.hi]).astype(np.int32)
        self.nbr_x, self.nbr_zl = gathered(self.rows_x), gathered(self.rows_zl)


class BatchADMM(object):

    def __init__(self, layout, nbr_global, ops, rank=0, world=1, dist=None, rho=1.0, horizon_time=10.,
                 nesterov_acceleration=False, nesterov_reset=False, eta=0.999, AMA=False):
        self.lay, self.ops, self.dist = layout, ops, dist
        self.halo = HaloPlan(np.asarray(nbr_global), rank, world)
        self.slot = reverse_slots(np.asarray(nbr_global))[self.halo.lo:self.halo.hi].astype(np.int32)
        self.rho, self.T = float(rho), float(horizon_time)
        # `problems/admm.py:568-571` (options of ADMMProblem)
        self.nesterov, self.nesterov_reset, self.eta, self.AMA = bool(nesterov_acceleration), bool(nesterov_reset), float(eta), bool(AMA)
        self._mcache = {}
        self.iteration = 0
        self._res = []                     # per iteration: backend array [3] = (pr^2 sum, dr^2 sum, cr sum), not fetched yet
        self._res_host, self.history_cap = [], 100000      # fetched rows (the newest `history_cap` are kept)
        if hasattr(ops, 'bind'):
            ops.bind(self.halo, self.slot)

    # -- exchange ---------------------------------------------------------------------
    def exchanging(self):
        return self.halo.world > 1 and self.halo.any_halo

    def extend(self, local, extra=None):
        """[B_local, w] -> [B_local + halo, w] (backend array type in, same type out).  `extra` ([k] values)
        rides along and comes back summed over the ranks."""
        if not self.exchanging():
            return (local, extra) if extra is not None else local
        out, summed = self.ops.exchange(local, self.halo, self.dist, extra)
        return (out, summed) if extra is not None else out

    def matrices(self, t_rel):
        key = round(t_rel / self.T, 12)
        if key not in self._mcache:
            if hasattr(self.lay, 'zupdate'):          # (rendezvous.RendezVousLayout: a plain consensus projector)
                M, F = self.lay.zupdate(key)
            else:
                M, F = zupdate_matrices(self.lay.basis, self.lay.n_dim, self.lay.n_nghb, key)
            self._mcache[key] = self.ops.resident(M, F) if hasattr(self.ops, 'resident') else (M, F)
        return self._mcache[key]

    # -- iteration ----------------------------------------------------------------------
    def initialize(self):
        self.ops.init_consensus(self.lay)

    def _mark(self, tag):
        """Phase stamp for a caller that asked for a timeline (`ops.timeline` = a list: bench.py --workload formation
        reports x-update / exchange / update times per iteration): an event on the stream, nothing waits for it."""
        if getattr(self.ops, 'timeline', None) is not None:
            self.ops.mark(tag)

    def iterate(self, t_rel=0.0, sync=True):
        """One ADMM iteration (`admm.py:584-611`).  sync=False returns (status, None) and leaves the
        residuals on the device (`residuals` fetches the whole history at once)."""
        ops, lay = self.ops, self.lay
        self._mark('begin')
        ops.set_time(lay, t_rel, self.rho)
        status = ops.solve()                                   # x-update
        self._mark('x_update')
        if self.exchanging() and not self.nesterov and getattr(ops, 'fused', False):
            # sharded fleet, no copies around the collectives: three launches and two all_gathers per iteration, every
            # kernel reads and writes the exchange buffers in place (include/omgx.h omgx_admm_*_ex)
            ops.center(lay)                                    # x_i rows + the rows other ranks need -> send buffer (left there by the x-update's epilogue: no launch)
            self._mark('centre')
            ops.gather_x(self.dist)                            # collective #1, straight behind the local rows
            self._mark('collective_x')
            M, F = self.matrices(t_rel)
            ops.update_fused(lay, M, F, self.rho)              # [z_ij | l_ij] in place; published rows and this rank's sums -> send buffer
            self._mark('z_l_update')
            ops.gather_zl(self.dist)                           # collective #2
            self._mark('collective_zl')
            sums = ops.communicate_fused(lay)                  # z_ji, l_ji from the gathered rows; the fleet's residual sums
            self._mark('read_back')
            self._res.append(sums)
            self.iteration += 1
            if not sync:
                return status, None
            s3 = ops.to_host(sums)
            return status, (float(np.sqrt(s3[0])), float(np.sqrt(s3[1])), float(s3[2]))
        x_i = ops.center(lay)
        self._mark('centre')
        x_ext = self.extend(x_i)                               # collective #1
        self._mark('collective_x')
        M, F = self.matrices(t_rel)
        if self.nesterov:
            ops.save_previous(lay)                             # z_p, l_p of `admm.py:409-410, 450-451`
        res = ops.update(lay, x_ext, self.halo.nbr_local, M, F, self.rho)
        sums = ops.residual_sums(res)                          # [3], backend array, no host copy
        self._mark('z_l_update')
        # with acceleration the previous z_ij, l_ij travel too: the extrapolation is elementwise with fleet-wide
        # scalars, so every rank applies it to the rows it received instead of waiting for a third collective
        if not self.nesterov and not self.exchanging() and getattr(ops, '_slot', None) is not None:
            # nothing travels and nothing is extrapolated: the neighbours' rows are read where update left them
            if self.dist is not None and self.halo.world > 1:
                sums = ops.allreduce(sums, self.dist)
            ops.communicate_local(lay)
            self._mark('read_back')
            self._res.append(sums)
            self.iteration += 1
            if not sync:
                return status, None
            s3 = ops.to_host(sums)
            return status, (float(np.sqrt(s3[0])), float(np.sqrt(s3[1])), float(s3[2]))
        buf = ops.zl_flat(with_prev=self.nesterov)
        if self.exchanging():
            buf, sums = self.extend(buf, sums)                 # collective #2: [z_ij | l_ij] rows + the residual sums
        elif self.dist is not None and self.halo.world > 1:
            sums = ops.allreduce(sums, self.dist)              # (no halo anywhere: disjoint groups still share the sums)
        zl_ext = ops.accelerate(lay, sums, buf, self.eta, self.nesterov_reset, self.AMA) if self.nesterov else buf
        ops.communicate(lay, self.halo.nbr_local, self.slot, zl_ext)
        self._mark('read_back')
        self._res.append(sums)
        self.iteration += 1
        if not sync:
            return status, None
        s3 = ops.to_host(sums)
        return status, (float(np.sqrt(s3[0])), float(np.sqrt(s3[1])), float(s3[2]))

    @property
    def residuals(self):
        """[(primal, dual, combined)] per iteration (`admm.py:601-605, 624-627`); one host copy."""
        if not self._res:
            return list(self._res_host)
        # only the rows that were not fetched yet travel to the host; the device-side handles are dropped afterwards
        arr = self.ops.to_host_stack(self._res)
        self._res_host += [(float(np.sqrt(r[0])), float(np.sqrt(r[1])), float(r[2])) for r in arr]
        self._res = []
        if len(self._res_host) > self.history_cap:
            del self._res_host[:len(self._res_host) - self.history_cap]
        return list(self._res_host)


class FullConsensusADMM(object):
    """Formation ADMM of a fleet with `interconnection='full'` (`vehicles/fleet.py:55-56`: every vehicle is every other's
    neighbour) with ONE collective per iteration -- BASELINE.json's "all-reduce of z / lambda" (SURVEY.md §8e).

    With every pair coupled, the copies z_i, z_ij of one agent's z-update (`problems/admm.py:407-445`) are all equal to one
    vector, c_i = Pi (1/N) sum_k (x_k + l_ik / rho) (Pi: the projector of the terminal rows), and -- whatever the initial
    guess -- from the first z-update on the c_i of all agents are the same c: the multipliers an agent keeps for agent k's
    copy, l_ik = rho sum_t (x_k - c)_t, do not depend on i.  The N - 1 neighbour blocks of the x-update objective
    (`admm.py:63-107`) then all hold (c, l_i) and add up to N [l_i'(x - c) + rho / 2 |x - c|^2].  So the iteration needs
    the fleet only through sum_k (x_k + l_k / rho): an all_reduce(sum) of n_shared doubles, with the three residual sums of
    the previous iteration riding along (n_shared + 3 doubles per iteration, nothing else crosses ranks).

    Runs on the x-update template of ANY neighbour count n (the usual one has 2): the 1 + n blocks get the weights
    N rho / (1 + n) and N l_i / (1 + n).  `ops`: NumpyAdmmOps / HipAdmmOps (x-update, centre; the rest is array
    arithmetic on the backend's arrays).  tests/test_admm_cpu.py: equal to the general iteration (`BatchADMM` on the
    template with N - 1 neighbour blocks) on a four-vehicle fleet, and 2 gloo ranks == 1 rank."""

    def __init__(self, layout, ops, n_agents, rank=0, world=1, dist=None, rho=1.0):
        self.lay, self.ops, self.N, self.rho = layout, ops, int(n_agents), float(rho)
        self.rank, self.world, self.dist = rank, world, dist
        basis, d, L, nd = layout.basis, layout.degree, layout.L, layout.n_dim
        # projector of the terminal rows (d^o/dtau^o centre)(1) = 0, o = 1..degree, per dimension (`formation.py:46-65`)
        P_term = np.array([basis.derivative(o)[1][-1, :] for o in range(1, d + 1)]).reshape(d, L)
        Pi1 = np.eye(L) - P_term.T @ np.linalg.solve(P_term @ P_term.T, P_term) if d > 0 else np.eye(L)
        self.Pi = ops.asarray(np.kron(np.eye(nd), Pi1))
        self.iteration = 0
        self.collectives = 0
        self._pending = None          # local residual partials of the last iteration (they travel with the next all_reduce)
        self._res = []

    def initialize(self):
        """`admm.py:360-370`: every copy starts at its agent's own centre, multipliers zero."""
        ops, lay = self.ops, self.lay
        ops.init_consensus(lay)
        x_i = ops.center(lay)
        self.z_prev = x_i * 1.0          # what the fleet's copies of agent k hold
        self.c = None
        self.l = x_i * 0.0

    def _set_consensus(self):
        """(c, l_i) into the 1 + n blocks of the x-update's parameters, with the weights that make them stand for N copies."""
        ops, lay = self.ops, self.lay
        ns, nn = lay.ns, lay.n_nghb
        w = self.N / float(1 + nn)
        z = self.z_prev if self.c is None else self.c
        ops.p[:, lay.p_zi:lay.p_zi + ns] = z
        ops.p[:, lay.p_li:lay.p_li + ns] = w * self.l
        for j in range(nn):
            ops.p[:, lay.p_zji + j * ns:lay.p_zji + (j + 1) * ns] = z
            ops.p[:, lay.p_lji + j * ns:lay.p_lji + (j + 1) * ns] = w * self.l
        ops.p[:, lay.p_rho] = w * self.rho

    def iterate(self, t_rel=0.0):
        ops, lay, rho, N = self.ops, self.lay, self.rho, self.N
        ops.set_time(lay, t_rel, rho)
        self._set_consensus()
        status = ops.solve()                                      # x-update
        x_i = ops.center(lay)
        ns = lay.ns
        buf = ops.asarray(np.zeros(ns + 3))
        buf[:ns] = (x_i + self.l / rho).sum(0)
        if self._pending is not None:
            buf[ns:] = self._pending
        if self.dist is not None and self.world > 1:
            buf = ops.allreduce(buf, self.dist)                   # the one collective of the iteration
            self.collectives += 1
        if self._pending is not None:
            self._res.append(buf[ns:] * 1.0)
        c = self.Pi @ (buf[:ns] / N)
        self.l = self.l + rho * (x_i - c)
        s_p = ((x_i - c) ** 2).sum()
        # what the general iteration reports (`admm.py:493-508` summed over the fleet): every agent counts all N copies.
        # (Before the first z-update an agent's own copy holds its centre and its copies of the others hold zero,
        # `admm.py:360-370`; afterwards every copy holds the previous c.)
        if self.iteration == 0:
            dual = ((c - self.z_prev) ** 2).sum() + x_i.shape[0] * (N - 1) * (c ** 2).sum()
        else:
            dual = N * ((c - self.z_prev) ** 2).sum()
        self._pending = ops.asarray(np.zeros(3))
        self._pending[0], self._pending[1], self._pending[2] = N * s_p, rho * dual, rho * N * s_p + rho * dual
        self.c = c
        self.z_prev = x_i * 0.0 + c
        self.iteration += 1
        return status

    @property
    def residuals(self):
        """[(primal, dual, combined)] per iteration; the sums of the last iteration are still local: one more all_reduce."""
        rows = list(self._res)
        if self._pending is not None:
            last = self._pending * 1.0
            if self.dist is not None and self.world > 1:
                last = self.ops.allreduce(last, self.dist)
            rows.append(last)
        arr = self.ops.to_host_stack(rows) if rows else np.zeros((0, 3))
        return [(float(np.sqrt(r[0])), float(np.sqrt(r[1])), float(r[2])) for r in arr]


class FormationMPC(object):
    """The receding-horizon protocol of the reference's ADMM problems with nothing leaving the device
    (`problems/dualmethod.py:200-224`: `init_iter` iterations at the start time, afterwards `max_iter_per_update`
    iteration(s) per update, the time advanced by `update_time`; `problems/admm.py:477-491`: on a knot crossing the
    warm start of x and the whole consensus state are shifted; ideal prediction `vehicles/vehicle.py:323-326`: the
    initial conditions of the next x-update are the current plan at the new time).  One `step()` =
      prediction launch (state0, input0, t into p) -> moving obstacles advanced -> shift on a crossing -> ADMM iteration(s).
    ops: `HipAdmmOps`; obstacles: [(p_x, p_v, p_a, n_dim)] of the obstacles that move."""

    def __init__(self, admm, father, tpl, lay, vehicle, obstacles=(), update_time=0.1, init_iter=5, iters_per_update=1,
                 knot_time=None, consensus_is_spline=True):
        from .consensus import shift_tables
        self.admm, self.ops, self.lay, self.tpl = admm, admm.ops, lay, tpl
        self.T, self.update_time = admm.T, float(update_time)
        # the plan the prediction reads: the vehicle's own splines (the consensus quantity may be something else: RendezVous)
        self.basis, self.n_spl = vehicle.basis, vehicle.n_spl
        self.o_spl = tpl.entry_range(vehicle.label, 'splines_seg0', 'var')[0]
        self.knot_time = float(knot_time) if knot_time is not None else self.T / (len(self.basis.knots) - 2 * self.basis.degree - 1)
        self.init_iter, self.iters_per_update = int(init_iter), int(iters_per_update)
        self.shift = shift_tables(father, tpl, lay, self.basis, consensus_is_spline)
        self.obst = list(obstacles)
        self.time = 0.0

    def initialize(self):
        self.admm.initialize()
        for _ in range(self.init_iter):
            self.admm.iterate(0.0, sync=False)
        # the z-update matrices of every time an update can happen at (the multiples of update_time modulo knot_time: one
        # period) go to the device now -- what the reference's exporter generates ahead of time as updz.so -- so that no
        # step computes and uploads one (formation bench: 0.44 -> 0.39 ms per update)
        if hasattr(self.ops, 'stage_shift'):            # (and the shift tables of the knot crossings)
            self.ops.stage_shift(*self.shift)
        from .backend import admm_table_keys
        try:
            for t_rel in admm_table_keys(self.knot_time, self.update_time, 256):
                self.admm.matrices(t_rel)
        except ValueError:                              # incommensurable times: matrices as they come
            pass

    def step(self):
        lay, ops = self.lay, self.ops
        t_prev, t_now = self.time, self.time + self.update_time
        from .splines import since_knot
        rel_prev = since_knot(t_prev, self.knot_time)
        tau = (rel_prev + self.update_time) / self.T
        crossed = int(np.round(t_prev / self.knot_time, 6)) < int(np.round(t_now / self.knot_time, 6))
        t_rel = since_knot(t_now, self.knot_time)
        ops.predict(self.o_spl, self.n_spl, self.basis, tau, 1.0 / self.T, [lay.p_state0, lay.p_input0], lay.p_t, t_rel)
        dt = self.update_time
        for ox, ov, oa, nd in self.obst:        # x <- x + v dt + a dt^2 / 2, v <- v + a dt (`environment/obstacle.py:246-264`)
            px, pv, pa = ops.p[:, ox:ox + nd], ops.p[:, ov:ov + nd], ops.p[:, oa:oa + nd]
            px += dt * pv + (0.5 * dt * dt) * pa
            pv += dt * pa
        if crossed:
            ops.shift(*self.shift)
        self.time = t_now
        status = None
        for _ in range(self.iters_per_update):
            status, _ = self.admm.iterate(t_rel, sync=False)
        return status, crossed


class HipAdmmOps(object):
    """Device-resident state + the HIP kernels of include/omgx.h (torch tensors
    only as the allocator / collective carrier)."""

    def __init__(self, solver, template, layout, p, x0, device):
        import torch
        self.torch, self.dev = torch, device
        self.solver, self.tpl = solver, template
        f64 = dict(dtype=torch.float64, device=device)
        B = solver.n_agents
        self.B, self.ns, self.nn = B, layout.ns, layout.n_nghb
        self.p = torch.as_tensor(np.ascontiguousarray(p), **f64)
        self.x = torch.as_tensor(np.ascontiguousarray(x0), **f64)
        self.x_new = torch.empty_like(self.x)
        self.lb = torch.as_tensor(template.lb, **f64)
        self.ub = torch.as_tensor(template.ub, **f64)
        # consecutive x-updates are neighbouring problems: primal-dual warm start from the previous
        # one (status 1 everywhere = the first solve is cold)
        self.lam = torch.zeros((B, template.n_con), **f64)
        self.status = torch.ones(B, dtype=torch.int32, device=device)
        solver.set_options(warm_start=1)
        self.iters = torch.empty(B, dtype=torch.int32, device=device)
        self.x_i = torch.empty((B, self.ns), **f64)
        self.z_ij = torch.zeros((B, self.nn, self.ns), **f64)
        self.l_ij = torch.zeros((B, self.nn, self.ns), **f64)
        self.res = torch.empty((B, 3), **f64)
        self.layc = AdmmLayoutC(layout.n_dim, layout.L, layout.n_nghb, layout.x_spl, layout.p_rel,
                                layout.p_zi, layout.p_zji, layout.p_li, layout.p_lji)
        lib = solver.lib
        lib.omgx_admm_center.argtypes = [C.c_void_p, C.POINTER(AdmmLayoutC)] + [C.c_void_p] * 3
        lib.omgx_admm_update.argtypes = [C.c_void_p, C.POINTER(AdmmLayoutC)] + [C.c_void_p] * 4 + \
            [C.c_double] + [C.c_void_p] * 4
        lib.omgx_admm_update_sums.argtypes = [C.c_void_p, C.POINTER(AdmmLayoutC)] + [C.c_void_p] * 4 + \
            [C.c_double] + [C.c_void_p] * 5
        lib.omgx_admm_communicate.argtypes = [C.c_void_p, C.POINTER(AdmmLayoutC)] + [C.c_void_p] * 5
        lib.omgx_admm_center_ex.argtypes = [C.c_void_p, C.POINTER(AdmmLayoutC)] + [C.c_void_p] * 4 + [C.c_int32, C.c_void_p]
        lib.omgx_admm_update_ex.argtypes = [C.c_void_p, C.POINTER(AdmmLayoutC)] + [C.c_void_p] * 4 + [C.c_double] + \
            [C.c_void_p] * 3 + [C.c_int32] + [C.c_void_p] * 4 + [C.c_int32]
        lib.omgx_admm_communicate_ex.argtypes = [C.c_void_p, C.POINTER(AdmmLayoutC)] + [C.c_void_p] * 4 + [C.c_int32] + \
            [C.c_void_p] * 2 + [C.c_int32, C.c_int32, C.c_void_p]
        if not os.environ.get('OMGX_NO_FUSED_CENTER'):       # (developer knob: the centre step as its own launch)
            lib.omgx_batch_set_center.argtypes = [C.c_void_p, C.POINTER(AdmmLayoutC), C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
        self.zl_stride = self.nn * self.ns             # doubles between the z_ij (l_ij) rows of consecutive agents
        self.fused = False
        self.center_fused = self._xi_fresh = False
        self.launches = self.collectives = 0           # kernel launches / collectives issued (tests count them per iteration)
        self._sum_blocks, self._sum_k, self._sums = [], 0, None     # fleet residual sums, one row per update (history)
        solver.set_stream(torch.cuda.current_stream().cuda_stream)
        self.zl = torch.zeros((B, 2 * self.nn * self.ns), **f64)      # [z_ij | l_ij] packed for the exchange
        self._nbr = self._slot = None
        # Nesterov state (`admm.py:510-554`), all on the device
        self.alpha = torch.ones((), **f64)
        self.c_res_p = None
        self._prev = None
        self.fuse_center()

    def fuse_center(self):
        """The centre step rides on the x-update (`omgx_batch_set_center`): every solve leaves x_i -- and, sharded, the rows
        other ranks need in x_send -- behind; `center` then has nothing to launch.  Called again when the exchange
        buffers are (re)bound.  A fleet that publishes a row twice keeps the separate launch."""
        if os.environ.get('OMGX_NO_FUSED_CENTER'):
            return
        pub = np.ascontiguousarray(self.halo.publish_local, dtype=np.int32) if self.fused else np.zeros(0, dtype=np.int32)
        rc = self.solver.lib.omgx_batch_set_center(
            self.solver._h, C.byref(self.layc), self.x_i.data_ptr(), pub.ctypes.data if len(pub) else None, len(pub),
            self.x_send.data_ptr() if len(pub) else None)
        self.center_fused, self._xi_fresh = (rc == 0), False

    # -- phase timeline (bench.py): `timeline = []` switches it on, `phase_times()` reads it after a synchronisation ----
    timeline = None

    def mark(self, tag):
        ev = self.torch.cuda.Event(enable_timing=True)
        ev.record()
        self.timeline.append((tag, ev))

    def phase_times(self):
        """Mean milliseconds per iteration between consecutive marks, by the tag of the mark that ends the span."""
        tot, cnt, prev = {}, {}, None
        for tag, ev in self.timeline:
            if tag != 'begin' and prev is not None:
                tot[tag] = tot.get(tag, 0.0) + prev.elapsed_time(ev)
                cnt[tag] = cnt.get(tag, 0) + 1
            prev = ev
        return dict((k, tot[k] / cnt[k]) for k in tot)

    def bind(self, halo, slot):
        """Index tensors that never change: uploaded once."""
        t = self.torch
        self._nbr = t.as_tensor(np.ascontiguousarray(halo.nbr_local), dtype=t.int32, device=self.dev)
        self._slot = t.as_tensor(np.ascontiguousarray(slot), dtype=t.int32, device=self.dev)
        self._pub = t.as_tensor(halo.publish_local, dtype=t.int64, device=self.dev)
        self._src = t.as_tensor(halo.src, dtype=t.int64, device=self.dev)
        # (the fused path lets the update launch put this rank's three residual sums into one more row of the second
        # exchange buffer: a fleet whose shared vector has a single number and one neighbour -- row width 2 -- takes the
        # general exchange)
        if halo.world > 1 and halo.any_halo and hasattr(self, 'z_ij') and 2 * self.nn * self.ns >= 3:
            self._bind_fused(halo)

    def _bind_fused(self, halo):
        """Exchange buffers of the sharded iteration: x_i, z_ij, l_ij become views of them (row strides change: every
        kernel call below passes `zl_stride`)."""
        t = self.torch
        f64 = dict(dtype=t.float64, device=self.dev)
        B, ns, w = self.B, self.ns, self.nn * self.ns
        self.halo = halo
        self.x_all = t.zeros((B + halo.world * halo.rows_x, ns), **f64)
        self.x_send = t.zeros((max(halo.rows_x, 1), ns), **f64)
        self.zl_all = t.zeros((B + halo.world * halo.rows_zl, 2 * w), **f64)
        self.zl_send = t.zeros((halo.rows_zl, 2 * w), **f64)
        self.x_i = self.x_all[:B]
        z_old, l_old = self.z_ij, self.l_ij
        self.z_ij = self.zl_all[:B, :w].view(B, self.nn, ns)
        self.l_ij = self.zl_all[:B, w:].view(B, self.nn, ns)
        self.z_ij.copy_(z_old)
        self.l_ij.copy_(l_old)
        self.zl_stride = 2 * w
        self._nbr_x = t.as_tensor(np.ascontiguousarray(halo.nbr_x), dtype=t.int32, device=self.dev)
        self._nbr_zl = t.as_tensor(np.ascontiguousarray(halo.nbr_zl), dtype=t.int32, device=self.dev)
        self._pub_rows = t.as_tensor(np.ascontiguousarray(halo.publish_local), dtype=t.int32, device=self.dev)
        self._pub_slot = t.as_tensor(np.ascontiguousarray(halo.pub_slot), dtype=t.int32, device=self.dev)
        self.fused = True
        self.fuse_center()

    def asarray(self, a):
        return self.torch.as_tensor(np.ascontiguousarray(a), dtype=self.torch.float64, device=self.dev)

    def resident(self, M, F):
        t = self.torch
        return (t.as_tensor(np.ascontiguousarray(M), dtype=t.float64, device=self.dev),
                t.as_tensor(np.ascontiguousarray(F), dtype=t.float64, device=self.dev))

    def _chk(self, rc, what):
        if rc != 0:
            raise RuntimeError('%s failed: %s' % (what, self.solver.lib.omgx_last_error().decode()))

    def init_consensus(self, lay):
        """`admm.py:360-370`: z_i = x_i, z_ji = x_i of the local agent, multipliers zero."""
        x_i = self.center(lay)
        p = self.p
        p[:, lay.p_zi:lay.p_zi + self.ns] = x_i
        p[:, lay.p_li:lay.p_li + self.ns] = 0.
        p[:, lay.p_zji:lay.p_zji + self.nn * self.ns] = x_i.repeat(1, self.nn)
        p[:, lay.p_lji:lay.p_lji + self.nn * self.ns] = 0.
        self.z_ij.zero_()
        self.l_ij.zero_()

    def set_time(self, lay, t_rel, rho):
        # (both stay the same over the iterations of one update: written only when they change)
        if getattr(self, '_t_rho', None) != (float(t_rel), float(rho)):
            self.p[:, lay.p_t] = t_rel
            self.p[:, lay.p_rho] = rho
            self._t_rho = (float(t_rel), float(rho))

    def predict(self, o_spl, n_spl, basis, tau, inv_T, p_offs, p_t, t_rel):
        """Ideal prediction (`vehicles/vehicle.py:323-326`): p[p_offs[o] + k] <- o-th time derivative of spline k of the
        current plan at tau, p[p_t] <- t_rel; one launch (`omgx_batch_predict_ex`)."""
        self.solver.predict_ex(self.x, self.p, o_spl, n_spl, basis.degree, basis.knots, tau, inv_T, p_offs, p_t, t_rel)
        if getattr(self, '_t_rho', None) is not None:       # (the launch wrote t: set_time has nothing left to do)
            self._t_rho = (float(t_rel), self._t_rho[1])

    def solve(self):
        self.solver.solve_device(self.p, self.x, self.lb, self.ub, self.x_new, self.lam,
                                 self.status, self.iters, bounds_shared=True)
        self.x, self.x_new = self.x_new, self.x
        self.launches += 1
        self._xi_fresh = self.center_fused
        return self.status

    def center(self, lay):
        if self._xi_fresh:                          # (written by the solve that just ran)
            return self.x_i
        n_pub = len(self.halo.publish_local) if self.fused else 0
        self._chk(self.solver.lib.omgx_admm_center_ex(
            self.solver._h, C.byref(self.layc), self.x.data_ptr(), self.p.data_ptr(), self.x_i.data_ptr(),
            self._pub_rows.data_ptr() if n_pub else None, n_pub, self.x_send.data_ptr() if n_pub else None), 'omgx_admm_center_ex')
        self.launches += 1
        return self.x_i

    # -- the sharded iteration without copies (BatchADMM.iterate) -------------------------------------------------
    def gather_x(self, dist):
        dist.all_gather_into_tensor(self.x_all[self.B:], self.x_send[:self.halo.rows_x])
        self.collectives += 1

    def gather_zl(self, dist):
        dist.all_gather_into_tensor(self.zl_all[self.B:], self.zl_send)
        self.collectives += 1

    def _history_row(self):
        t = self.torch
        if self._sum_k % 1024 == 0:
            self._sum_blocks.append(t.zeros((1024, 3), dtype=t.float64, device=self.dev))
            self._sum_blocks = self._sum_blocks[-1:]      # (rows of older blocks stay alive through the views BatchADMM holds until they are fetched)
        row = self._sum_blocks[-1][self._sum_k % 1024]
        self._sum_k += 1
        return row

    def update_fused(self, lay, M, F, rho):
        if not self.torch.is_tensor(M):
            M, F = self.resident(M, F)
        w = self.nn * self.ns
        self._chk(self.solver.lib.omgx_admm_update_ex(
            self.solver._h, C.byref(self.layc), self.x_all.data_ptr(), self._nbr_x.data_ptr(), M.data_ptr(), F.data_ptr(),
            float(rho), self.p.data_ptr(), self.z_ij.data_ptr(), self.l_ij.data_ptr(), self.zl_stride, self.res.data_ptr(),
            self.zl_send[self.halo.rows_zl - 1].data_ptr(), self._pub_slot.data_ptr(), self.zl_send.data_ptr(), 2 * w),
            'omgx_admm_update_ex')
        self._keep = (M, F)
        self.launches += 1

    def communicate_fused(self, lay):
        w, el = self.nn * self.ns, 8
        halo = self.halo
        sums = self._history_row()
        base = self.zl_all.data_ptr()
        self._chk(self.solver.lib.omgx_admm_communicate_ex(
            self.solver._h, C.byref(self.layc), self._nbr_zl.data_ptr(), self._slot.data_ptr(), base, base + w * el,
            2 * w, self.p.data_ptr(), base + (self.B + halo.rows_zl - 1) * 2 * w * el, halo.world, halo.rows_zl * 2 * w,
            sums.data_ptr()), 'omgx_admm_communicate_ex')
        self.launches += 1
        return sums

    def update(self, lay, x_ext, nbr_local, M, F, rho):
        t = self.torch
        if self._nbr is None:                      # (used without BatchADMM.bind: upload now)
            self._nbr = t.as_tensor(np.ascontiguousarray(nbr_local), dtype=t.int32, device=self.dev)
        if not t.is_tensor(M):
            M, F = self.resident(M, F)
        x_ext = x_ext.contiguous()
        # the fleet sums of the residuals come out of the same launch (the workgroup that finishes last adds them up
        # in a fixed order); every update gets its own row, BatchADMM keeps them as the residual history
        self._sums = self._history_row()
        self._chk(self.solver.lib.omgx_admm_update_ex(
            self.solver._h, C.byref(self.layc), x_ext.data_ptr(), self._nbr.data_ptr(), M.data_ptr(),
            F.data_ptr(), float(rho), self.p.data_ptr(), self.z_ij.data_ptr(), self.l_ij.data_ptr(), self.zl_stride,
            self.res.data_ptr(), self._sums.data_ptr(), None, None, 0), 'omgx_admm_update_ex')
        self._keep = (M, F, x_ext)
        self.launches += 1
        return self.res

    def residual_sums(self, res):
        return self._sums if res is self.res and self._sums is not None else res.sum(dim=0)

    def zl_flat(self, with_prev=False):
        B, w = self.B, self.nn * self.ns
        if with_prev and self.zl.shape[1] != 4 * w:
            self.zl = self.torch.zeros((B, 4 * w), dtype=self.torch.float64, device=self.dev)
        self.zl[:, :w] = self.z_ij.view(B, -1)
        self.zl[:, w:2 * w] = self.l_ij.view(B, -1)
        if with_prev:
            self.zl[:, 2 * w:3 * w] = self._prev[1].view(B, -1)
            self.zl[:, 3 * w:] = self._prev[3].view(B, -1)
        return self.zl

    def z_ij_flat(self):
        return self.z_ij.view(self.B, -1)

    def l_ij_flat(self):
        return self.l_ij.view(self.B, -1)

    def communicate(self, lay, nbr_local, slot, zl_ext):
        t = self.torch
        if self._slot is None:
            self._nbr = t.as_tensor(np.ascontiguousarray(nbr_local), dtype=t.int32, device=self.dev)
            self._slot = t.as_tensor(np.ascontiguousarray(slot), dtype=t.int32, device=self.dev)
        w = self.nn * self.ns
        # (the rows are read where they are: z_ij at column 0, l_ij at column w of the exchanged rows)
        zl_ext = zl_ext if zl_ext.stride(1) == 1 else zl_ext.contiguous()
        self._chk(self.solver.lib.omgx_admm_communicate_ex(
            self.solver._h, C.byref(self.layc), self._nbr.data_ptr(), self._slot.data_ptr(), zl_ext.data_ptr(),
            zl_ext.data_ptr() + w * zl_ext.element_size(), zl_ext.stride(0), self.p.data_ptr(), None, 0, 0, None),
            'omgx_admm_communicate_ex')
        self._keep2 = zl_ext
        self.launches += 1

    def communicate_local(self, lay):
        """communicate when every neighbour is a local agent and nothing is extrapolated: z_ij, l_ij as they are."""
        self._chk(self.solver.lib.omgx_admm_communicate_ex(
            self.solver._h, C.byref(self.layc), self._nbr.data_ptr(), self._slot.data_ptr(), self.z_ij.data_ptr(),
            self.l_ij.data_ptr(), self.zl_stride, self.p.data_ptr(), None, 0, 0, None), 'omgx_admm_communicate_ex')
        self.launches += 1

    # -- Nesterov acceleration (`admm.py:510-554`), branch-free on the device --------------------
    def save_previous(self, lay):
        ns = self.ns
        self._prev = (self.p[:, lay.p_zi:lay.p_zi + ns].clone(), self.z_ij.clone(),
                      self.p[:, lay.p_li:lay.p_li + ns].clone(), self.l_ij.clone())

    def accelerate(self, lay, sums, ext, eta, reset, AMA):
        """ext [B + halo, 4w] = [z_ij | l_ij | z_ij_p | l_ij_p] -> accelerated [z_ij | l_ij] of the same rows; the
        local rows and z_i, l_i are updated in place."""
        t, ns, B, w = self.torch, self.ns, self.B, self.nn * self.ns
        c_res = sums[2]
        if self.c_res_p is None:
            self.c_res_p = c_res / eta
        z_i, l_i = self.p[:, lay.p_zi:lay.p_zi + ns], self.p[:, lay.p_li:lay.p_li + ns]
        z_i_p, l_i_p = self._prev[0], self._prev[2]
        alpha_p = self.alpha
        alpha_n = 0.5 * (1. + t.sqrt(1. + 4. * alpha_p ** 2))
        wl = (alpha_p - 1.) / alpha_n
        wz = t.zeros_like(wl) if AMA else wl
        good = (c_res <= eta * self.c_res_p) if reset else t.ones((), dtype=t.bool, device=self.dev)
        z_i.copy_(t.where(good, z_i + wz * (z_i - z_i_p), z_i_p))
        l_i.copy_(t.where(good, l_i + wl * (l_i - l_i_p), l_i_p))
        z, l, z_p, l_p = ext[:, :w], ext[:, w:2 * w], ext[:, 2 * w:3 * w], ext[:, 3 * w:]
        out = t.cat([t.where(good, z + wz * (z - z_p), z_p), t.where(good, l + wl * (l - l_p), l_p)], dim=1)
        self.z_ij.copy_(out[:B, :w].reshape(B, self.nn, ns))
        self.l_ij.copy_(out[:B, w:].reshape(B, self.nn, ns))
        self.alpha = t.where(good, alpha_n, t.ones_like(alpha_n))
        self.c_res_p = t.where(good, c_res, self.c_res_p / eta)
        return out

    def to_host(self, a):
        return a.cpu().numpy()

    def to_host_stack(self, lst):
        return self.torch.stack(lst).cpu().numpy()

    def allreduce(self, sums, dist):
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        return sums

    # -- host bookkeeping of the drop-in problem class (formation.FormationPoint2point) -----------
    def upload_params(self, p_host, cols):
        t = self.torch
        idx = t.as_tensor(np.asarray(cols), dtype=t.int64, device=self.dev)
        self._t_rho = None
        self._xi_fresh = False
        self.p[:, idx] = t.as_tensor(np.ascontiguousarray(p_host[:, cols]), dtype=t.float64, device=self.dev)

    def upload_x(self, x_host):
        self._xi_fresh = False
        self.x.copy_(self.torch.as_tensor(np.ascontiguousarray(x_host), dtype=self.torch.float64, device=self.dev))

    def download_x(self):
        return self.x.cpu().numpy()

    def shift(self, shift_x, shift_p, shift_side):
        """Knot crossing: x <- T x for every spline variable, and the same shift of the consensus
        state z_i, l_i, z_ji, l_ji (inside p) and z_ij, l_ij (`admm.py:477-491`)."""
        lib, h = self.solver.lib, self.solver._h
        self._xi_fresh = False
        lib.omgx_shift_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                        C.c_int32, C.c_void_p, C.c_int32]
        for data, stride, (ents, mats) in ((self.x, self.x.shape[1], shift_x), (self.p, self.p.shape[1], shift_p),
                                           (self.z_ij, self.zl_stride, shift_side),
                                           (self.l_ij, self.zl_stride, shift_side)):
            ents = np.ascontiguousarray(ents, dtype=np.int32)
            mats = np.ascontiguousarray(mats, dtype=np.float64)
            if len(ents) == 0:
                continue
            self._chk(lib.omgx_shift_rows(h, data.data_ptr(), int(stride), self.B, None, ents.ctypes.data,
                                          len(ents), mats.ctypes.data, mats.size), 'omgx_shift_rows')

    def stage_shift(self, shift_x, shift_p, shift_side):
        """The shift tables of `shift` uploaded ahead of the loop (n_rows = 0: nothing launched)."""
        lib, h = self.solver.lib, self.solver._h
        lib.omgx_shift_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                        C.c_int32, C.c_void_p, C.c_int32]
        for stride, (ents, mats) in ((self.x.shape[1], shift_x), (self.p.shape[1], shift_p), (self.zl_stride, shift_side)):
            ents = np.ascontiguousarray(ents, dtype=np.int32)
            mats = np.ascontiguousarray(mats, dtype=np.float64)
            if len(ents):
                self._chk(lib.omgx_shift_rows(h, None, int(stride), 0, None, ents.ctypes.data, len(ents), mats.ctypes.data, mats.size),
                          'omgx_shift_rows')

    def exchange(self, local, halo, dist, extra=None):
        """One all_gather: every rank sends the rows other ranks need (+ one row carrying `extra`, whose sum
        over the ranks is returned).  -> ([B_local + halo, w], summed extra or None)."""
        t = self.torch
        wl = local.shape[1]
        # (the row that carries `extra` needs extra.numel() columns: a RendezVous fleet with one shared number and one
        # neighbour exchanges rows of two)
        w = max(wl, extra.numel()) if extra is not None else wl
        rows = halo.max_pub + (1 if extra is not None else 0)
        send = t.zeros((rows, w), dtype=local.dtype, device=local.device)
        if len(halo.publish_local):
            send[:len(halo.publish_local), :wl] = local[self._pub]
        if extra is not None:
            send[halo.max_pub, :extra.numel()] = extra
        # (concatenated form [world * rows, w]: accepted by every backend; viewed as [world, rows, w] afterwards)
        flat = t.empty((halo.world * rows, w), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(flat, send)
        allp = flat.view(halo.world, rows, w)
        summed = allp[:, halo.max_pub, :extra.numel()].sum(dim=0) if extra is not None else None
        out = t.cat([local, allp[self._src[:, 0], self._src[:, 1], :wl]], dim=0) if len(halo.needed) else local
        return out, summed
