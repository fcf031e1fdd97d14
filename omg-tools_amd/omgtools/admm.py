"""`BatchADMM`: the formation ADMM iteration over a (sharded) fleet.

One iteration (`problems/admm.py:584-611`): x-update (batched NLP solve) ->
exchange x -> z-update, lambda update, residuals -> exchange z/lambda; the
kernels are reached through an `ops` object: `HipAdmmOps` (device tensors +
include/omgx.h) on the GPU, or the numpy oracle in the CPU distributed test.

Sharding: agents are split contiguously over the ranks; a rank
needs the consensus rows of the few remote agents its own agents are neighbours
with (2 for the circular topology).  Every rank publishes the rows other ranks
need, one `all_gather` per exchange, plus one `all_reduce` of the three residual
sums per iteration (reference: plain attribute reads `admm.py:468-475` and a Python
sum `admm.py:601-603`).
"""
import ctypes as C

import numpy as np

from .distributed import shard_range
from .formation import zupdate_matrices, reverse_slots


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
        self.publish_local = np.array([g - self.lo for g in publish[rank]], dtype=np.int64)
        self.needed = needed[rank]
        # where each needed agent sits in the all_gather result [world, max_pub, width]
        self.src = np.array([[owner[g], publish[owner[g]].index(g)] for g in self.needed],
                            dtype=np.int64).reshape(-1, 2)
        Bl = self.hi - self.lo
        lookup = {g: Bl + i for i, g in enumerate(self.needed)}
        lookup.update({g: g - self.lo for g in range(self.lo, self.hi)})
        self.nbr_local = np.vectorize(lookup.__getitem__)(nbr_global[self.lo:self.hi]).astype(np.int32)


class BatchADMM(object):

    def __init__(self, layout, nbr_global, ops, rank=0, world=1, dist=None, rho=1.0, horizon_time=10.):
        self.lay, self.ops, self.dist = layout, ops, dist
        self.halo = HaloPlan(np.asarray(nbr_global), rank, world)
        self.slot = reverse_slots(np.asarray(nbr_global))[self.halo.lo:self.halo.hi].astype(np.int32)
        self.rho, self.T = float(rho), float(horizon_time)
        self._mcache = {}
        self.iteration = 0
        self.residuals = []

    # -- exchange ---------------------------------------------------------------------
    def extend(self, local):
        """[B_local, w] -> [B_local + halo, w] (backend array type in, same type out)."""
        if self.halo.world == 1 or not self.halo.needed:
            return local
        return self.ops.exchange(local, self.halo, self.dist)

    def matrices(self, t_rel):
        key = round(t_rel / self.T, 12)
        if key not in self._mcache:
            self._mcache[key] = zupdate_matrices(self.lay.basis, self.lay.n_dim, self.lay.n_nghb, key)
        return self._mcache[key]

    # -- iteration ----------------------------------------------------------------------
    def initialize(self):
        self.ops.init_consensus(self.lay)

    def iterate(self, t_rel=0.0):
        ops, lay = self.ops, self.lay
        ops.set_time(lay, t_rel, self.rho)
        status = ops.solve()                                   # x-update
        x_i = ops.center(lay)
        x_ext = self.extend(x_i)                               # communicate #1
        M, F = self.matrices(t_rel)
        res = ops.update(lay, x_ext, self.halo.nbr_local, M, F, self.rho)
        z_ext, l_ext = self.extend(ops.z_ij_flat()), self.extend(ops.l_ij_flat())
        ops.communicate(lay, self.halo.nbr_local, self.slot, z_ext, l_ext)   # communicate #2
        sums = ops.reduce_residuals(res, self.dist if self.halo.world > 1 else None)
        pr, dr, cr = float(np.sqrt(sums[0])), float(np.sqrt(sums[1])), float(sums[2])
        self.residuals.append((pr, dr, cr))
        self.iteration += 1
        return status, (pr, dr, cr)


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
        lib.omgx_admm_communicate.argtypes = [C.c_void_p, C.POINTER(AdmmLayoutC)] + [C.c_void_p] * 5
        solver.set_stream(torch.cuda.current_stream().cuda_stream)

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
        self.p[:, lay.p_t] = t_rel
        self.p[:, lay.p_rho] = rho

    def solve(self):
        self.solver.solve_device(self.p, self.x, self.lb, self.ub, self.x_new, self.lam,
                                 self.status, self.iters, bounds_shared=True)
        self.x, self.x_new = self.x_new, self.x
        return self.status

    def center(self, lay):
        self._chk(self.solver.lib.omgx_admm_center(self.solver._h, C.byref(self.layc), self.x.data_ptr(),
                                                   self.p.data_ptr(), self.x_i.data_ptr()), 'omgx_admm_center')
        return self.x_i

    def update(self, lay, x_ext, nbr_local, M, F, rho):
        t = self.torch
        nbr = t.as_tensor(np.ascontiguousarray(nbr_local), dtype=t.int32, device=self.dev)
        Md = t.as_tensor(np.ascontiguousarray(M), dtype=t.float64, device=self.dev)
        Fd = t.as_tensor(np.ascontiguousarray(F), dtype=t.float64, device=self.dev)
        x_ext = x_ext.contiguous()
        self._chk(self.solver.lib.omgx_admm_update(
            self.solver._h, C.byref(self.layc), x_ext.data_ptr(), nbr.data_ptr(), Md.data_ptr(),
            Fd.data_ptr(), float(rho), self.p.data_ptr(), self.z_ij.data_ptr(), self.l_ij.data_ptr(),
            self.res.data_ptr()), 'omgx_admm_update')
        self._keep = (nbr, Md, Fd, x_ext)
        return self.res

    def z_ij_flat(self):
        return self.z_ij.view(self.B, -1)

    def l_ij_flat(self):
        return self.l_ij.view(self.B, -1)

    def communicate(self, lay, nbr_local, slot, z_ext, l_ext):
        t = self.torch
        nbr = t.as_tensor(np.ascontiguousarray(nbr_local), dtype=t.int32, device=self.dev)
        sl = t.as_tensor(np.ascontiguousarray(slot), dtype=t.int32, device=self.dev)
        z_ext, l_ext = z_ext.contiguous(), l_ext.contiguous()
        self._chk(self.solver.lib.omgx_admm_communicate(
            self.solver._h, C.byref(self.layc), nbr.data_ptr(), sl.data_ptr(), z_ext.data_ptr(),
            l_ext.data_ptr(), self.p.data_ptr()), 'omgx_admm_communicate')
        self._keep2 = (nbr, sl, z_ext, l_ext)

    # -- host bookkeeping of the drop-in problem class (formation.FormationPoint2point) -----------
    def upload_params(self, p_host, cols):
        t = self.torch
        idx = t.as_tensor(np.asarray(cols), dtype=t.int64, device=self.dev)
        self.p[:, idx] = t.as_tensor(np.ascontiguousarray(p_host[:, cols]), dtype=t.float64, device=self.dev)

    def upload_x(self, x_host):
        self.x.copy_(self.torch.as_tensor(np.ascontiguousarray(x_host), dtype=self.torch.float64, device=self.dev))

    def download_x(self):
        return self.x.cpu().numpy()

    def shift(self, shift_x, shift_p, shift_side):
        """Knot crossing: x <- T x for every spline variable, and the same shift of the consensus
        state z_i, l_i, z_ji, l_ji (inside p) and z_ij, l_ij (`admm.py:477-491`)."""
        lib, h = self.solver.lib, self.solver._h
        lib.omgx_shift_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                        C.c_int32, C.c_void_p, C.c_int32]
        for data, stride, (ents, mats) in ((self.x, self.x.shape[1], shift_x), (self.p, self.p.shape[1], shift_p),
                                           (self.z_ij, self.nn * self.ns, shift_side),
                                           (self.l_ij, self.nn * self.ns, shift_side)):
            ents = np.ascontiguousarray(ents, dtype=np.int32)
            mats = np.ascontiguousarray(mats, dtype=np.float64)
            self._chk(lib.omgx_shift_rows(h, data.data_ptr(), int(stride), self.B, None, ents.ctypes.data,
                                          len(ents), mats.ctypes.data, mats.size), 'omgx_shift_rows')

    def exchange(self, local, halo, dist):
        t = self.torch
        w = local.shape[1]
        send = t.zeros((halo.max_pub, w), dtype=local.dtype, device=local.device)
        if len(halo.publish_local):
            send[:len(halo.publish_local)] = local[t.as_tensor(halo.publish_local, device=local.device)]
        gathered = [t.empty_like(send) for _ in range(halo.world)]
        dist.all_gather(gathered, send)
        allp = t.stack(gathered)
        src = t.as_tensor(halo.src, device=local.device)
        return t.cat([local, allp[src[:, 0], src[:, 1]]], dim=0)

    def reduce_residuals(self, res, dist):
        sums = res.sum(dim=0)
        if dist is not None:
            dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        return sums.cpu().numpy()
