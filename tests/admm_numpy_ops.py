"""Test infrastructure: numpy `ops` backend for omgtools.admm.BatchADMM (same
interface as HipAdmmOps) used by the CPU distributed test; x-update by the oracle
CPU port."""
import numpy as np


class NumpyAdmmOps(object):
    def __init__(self, template, layout, p, x0, tol=1e-6, warm=True):
        self.warm = warm
        from oracle import port_binding
        self.port, self.tpl, self.tol = port_binding, template, tol
        self.p, self.x = np.array(p, float), np.array(x0, float)
        self.B, self.ns, self.nn = self.p.shape[0], layout.ns, layout.n_nghb
        self.z_ij = np.zeros((self.B, self.nn, self.ns))
        self.l_ij = np.zeros((self.B, self.nn, self.ns))
        self.fused, self.launches, self.collectives = False, 0, 0

    # -- the copy-free exchange of the sharded iteration (same buffers and indices as HipAdmmOps._bind_fused; the kernels
    # of include/omgx.h omgx_admm_*_ex restated on numpy views of the torch CPU tensors the collectives work on) --------
    def bind(self, halo, slot):
        if not (halo.world > 1 and halo.any_halo) or not getattr(self, 'use_fused', True):
            return
        import torch
        B, ns, w = self.B, self.ns, self.nn * self.ns
        self.halo, self._slot = halo, np.asarray(slot)
        self.t_x_all = torch.zeros((B + halo.world * halo.rows_x, ns), dtype=torch.float64)
        self.t_x_send = torch.zeros((max(halo.rows_x, 1), ns), dtype=torch.float64)
        self.t_zl_all = torch.zeros((B + halo.world * halo.rows_zl, 2 * w), dtype=torch.float64)
        self.t_zl_send = torch.zeros((halo.rows_zl, 2 * w), dtype=torch.float64)
        self.x_all, self.x_send = self.t_x_all.numpy(), self.t_x_send.numpy()
        self.zl_all, self.zl_send = self.t_zl_all.numpy(), self.t_zl_send.numpy()
        z_old, l_old = self.z_ij, self.l_ij
        self.z_ij = self.zl_all[:B, :w].reshape(B, self.nn, ns)         # views: updated in place from here on
        self.l_ij = self.zl_all[:B, w:].reshape(B, self.nn, ns)
        assert np.shares_memory(self.z_ij, self.zl_all) and np.shares_memory(self.l_ij, self.zl_all)
        self.z_ij[...], self.l_ij[...] = z_old, l_old
        self.fused = True

    def gather_x(self, dist):
        dist.all_gather_into_tensor(self.t_x_all[self.B:], self.t_x_send[:self.halo.rows_x])
        self.collectives += 1

    def gather_zl(self, dist):
        dist.all_gather_into_tensor(self.t_zl_all[self.B:], self.t_zl_send)
        self.collectives += 1

    def update_fused(self, lay, M, F, rho):
        halo, w = self.halo, self.nn * self.ns
        res = self.update(lay, self.x_all, halo.nbr_x, M, F, rho)
        pub = halo.publish_local
        self.zl_send[:len(pub), :w] = self.z_ij[pub].reshape(len(pub), w)
        self.zl_send[:len(pub), w:] = self.l_ij[pub].reshape(len(pub), w)
        self.zl_send[halo.rows_zl - 1, :3] = res.sum(axis=0)

    def communicate_fused(self, lay):
        halo, ns, nn = self.halo, self.ns, self.nn
        w = nn * ns
        z_ext, l_ext = self.zl_all[:, :w].reshape(-1, nn, ns), self.zl_all[:, w:].reshape(-1, nn, ns)
        nbr, slot = halo.nbr_zl, self._slot
        for k in range(nn):
            self.p[:, lay.p_zji + k * ns:lay.p_zji + (k + 1) * ns] = z_ext[nbr[:, k], slot[:, k]]
            self.p[:, lay.p_lji + k * ns:lay.p_lji + (k + 1) * ns] = l_ext[nbr[:, k], slot[:, k]]
        rows = self.zl_all[self.B:].reshape(halo.world, halo.rows_zl, 2 * w)
        self.launches += 1
        return rows[:, halo.rows_zl - 1, :3].sum(axis=0)

    def init_consensus(self, lay):
        x_i = self.center(lay)
        self.p[:, lay.p_zi:lay.p_zi + self.ns] = x_i
        self.p[:, lay.p_li:lay.p_li + self.ns] = 0.
        self.p[:, lay.p_zji:lay.p_zji + self.nn * self.ns] = np.tile(x_i, (1, self.nn))
        self.p[:, lay.p_lji:lay.p_lji + self.nn * self.ns] = 0.

    def set_time(self, lay, t_rel, rho):
        self.p[:, lay.p_t] = t_rel
        self.p[:, lay.p_rho] = rho

    def predict(self, o_spl, n_spl, basis, tau, inv_T, p_offs, p_t, t_rel):
        """Ideal prediction (`vehicles/vehicle.py:323-326`) with the front end's own spline algebra."""
        L = len(basis)
        c = self.x[:, o_spl:o_spl + n_spl * L].reshape(self.B, n_spl, L)
        for o, off in enumerate(p_offs):
            if o == 0:
                E = basis.eval_basis([tau])[0]
            else:
                dbasis, Po = basis.derivative(o)
                E = dbasis.eval_basis([tau])[0] @ Po * inv_T ** o
            self.p[:, off:off + n_spl] = c @ np.asarray(E).reshape(-1)
        self.p[:, p_t] = t_rel

    def solve(self):
        # consecutive x-updates are neighbouring problems: primal-dual warm start from the previous
        # one (first call: status 1 everywhere = cold), like HipAdmmOps
        if not hasattr(self, 'lam'):
            self.lam = np.zeros((self.B, self.tpl.n_con))
            self.status = np.ones(self.B, dtype=np.int32)
            self.dw = np.zeros(self.B)
        if not self.warm:
            self.status[:] = 1
        r = self.port.solve(self.tpl, self.p, self.x, tol=self.tol, max_iter=300, warm_start=1,
                            lam_g0=self.lam, status0=self.status, dw_state=self.dw,
                            warm_z_cap=0.0 if self.tol < 1e-4 else 0.01, max_soc=0 if self.tol < 1e-4 else 1)      # (as FormationPoint2point sets it for its 1e-6 x-updates)
        self.x, self.lam, self.status = r['x'], r['lam_g'], r['status']
        self.launches += 1
        return r['status']

    def center(self, lay):
        c = self.x[:, lay.x_spl:lay.x_spl + self.ns].reshape(self.B, lay.n_dim, lay.L)
        x_i = (c + self.p[:, lay.p_rel:lay.p_rel + lay.n_dim][:, :, None]).reshape(self.B, self.ns)
        self.launches += 1
        if self.fused:
            self.x_all[:self.B] = x_i
            pub = self.halo.publish_local
            self.x_send[:len(pub)] = x_i[pub]
            return self.x_all[:self.B]
        return x_i

    def update(self, lay, x_ext, nbr, M, F, rho):
        ns, nn, p = self.ns, self.nn, self.p
        x_all = np.concatenate([x_ext[:self.B, None, :], x_ext[nbr]], axis=1).reshape(self.B, -1)
        l_all = np.concatenate([p[:, None, lay.p_li:lay.p_li + ns], self.l_ij], axis=1).reshape(self.B, -1)
        z_prev = np.concatenate([p[:, None, lay.p_zi:lay.p_zi + ns], self.z_ij], axis=1).reshape(self.B, -1)
        z_all = (x_all + l_all / rho) @ M.T
        l_all = l_all + rho * (x_all - z_all)
        pr = (((x_all - z_all) @ F.T) ** 2).sum(axis=1)
        dr = rho * (((z_all - z_prev) @ F.T) ** 2).sum(axis=1)
        z_all, l_all = z_all.reshape(self.B, 1 + nn, ns), l_all.reshape(self.B, 1 + nn, ns)
        p[:, lay.p_zi:lay.p_zi + ns], p[:, lay.p_li:lay.p_li + ns] = z_all[:, 0], l_all[:, 0]
        self.z_ij[...], self.l_ij[...] = z_all[:, 1:], l_all[:, 1:]        # (in place: they may be views of the exchange buffer)
        self.launches += 1
        return np.stack([pr, dr, rho * pr + dr], axis=1)

    def upload_params(self, p_host, cols):
        self.p[:, cols] = p_host[:, cols]

    def upload_x(self, x_host):
        self.x = np.array(x_host, float)

    def download_x(self):
        return self.x.copy()

    def shift(self, shift_x, shift_p, shift_side):
        def apply(arr, ents, mats):
            for lo, rows, cols, off in ents:
                Tm = mats[off:off + rows * rows].reshape(rows, rows)
                blk = arr[:, lo:lo + rows * cols].reshape(arr.shape[0], cols, rows)
                arr[:, lo:lo + rows * cols] = (blk @ Tm.T).reshape(arr.shape[0], -1)
        apply(self.x, *shift_x)
        apply(self.p, *shift_p)
        for side in (self.z_ij, self.l_ij):
            flat = side.reshape(self.B, -1).copy()
            apply(flat, *shift_side)
            side[...] = flat.reshape(side.shape)              # (in place: z_ij, l_ij may be views of the exchange buffer)

    def z_ij_flat(self):
        return self.z_ij.reshape(self.B, -1)

    def l_ij_flat(self):
        return self.l_ij.reshape(self.B, -1)

    def zl_flat(self, with_prev=False):
        parts = [self.z_ij, self.l_ij] + ([self._prev[1], self._prev[3]] if with_prev else [])
        return np.concatenate([a.reshape(self.B, -1) for a in parts], axis=1)

    def residual_sums(self, res):
        return res.sum(axis=0)

    def communicate(self, lay, nbr, slot, zl_ext):
        ns, nn = self.ns, self.nn
        w = nn * ns
        z_ext, l_ext = zl_ext[:, :w].reshape(-1, nn, ns), zl_ext[:, w:].reshape(-1, nn, ns)
        for k in range(nn):
            self.p[:, lay.p_zji + k * ns:lay.p_zji + (k + 1) * ns] = z_ext[nbr[:, k], slot[:, k]]
            self.p[:, lay.p_lji + k * ns:lay.p_lji + (k + 1) * ns] = l_ext[nbr[:, k], slot[:, k]]
        self.launches += 1

    def exchange(self, local, halo, dist, extra=None):
        import torch
        wl = local.shape[1]
        w = max(wl, len(extra)) if extra is not None else wl
        rows = halo.max_pub + (1 if extra is not None else 0)
        send = torch.zeros((rows, w), dtype=torch.float64)
        if len(halo.publish_local):
            send[:len(halo.publish_local), :wl] = torch.from_numpy(local[halo.publish_local])
        if extra is not None:
            send[halo.max_pub, :len(extra)] = torch.from_numpy(np.asarray(extra, float))
        gathered = [torch.empty_like(send) for _ in range(halo.world)]
        dist.all_gather(gathered, send)
        allp = torch.stack(gathered).numpy()
        summed = allp[:, halo.max_pub, :len(extra)].sum(axis=0) if extra is not None else None
        out = np.concatenate([local, allp[halo.src[:, 0], halo.src[:, 1], :wl]], axis=0) if len(halo.needed) else local
        return out, summed

    def asarray(self, a):
        return np.array(a, float)

    def allreduce(self, sums, dist):
        import torch
        s = torch.from_numpy(np.asarray(sums, float).copy())
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        return s.numpy()

    def to_host(self, a):
        return np.asarray(a)

    def to_host_stack(self, lst):
        return np.stack([np.asarray(a) for a in lst])

    # -- Nesterov acceleration: the reference's statements (`admm.py:510-554`) ---------------------
    def save_previous(self, lay):
        ns = self.ns
        self._prev = (self.p[:, lay.p_zi:lay.p_zi + ns].copy(), self.z_ij.copy(),
                      self.p[:, lay.p_li:lay.p_li + ns].copy(), self.l_ij.copy())

    def accelerate(self, lay, sums, ext, eta, reset, AMA):
        ns, nn, B = self.ns, self.nn, self.B
        w = nn * ns
        c_res = float(sums[2])
        if not hasattr(self, 'c_res_p'):
            self.c_res_p = (1. / eta) * c_res
        if not hasattr(self, 'alpha'):
            self.alpha = 1.
        z_i, l_i = self.p[:, lay.p_zi:lay.p_zi + ns], self.p[:, lay.p_li:lay.p_li + ns]
        z_i_p, l_i_p = self._prev[0], self._prev[2]
        z_ij, l_ij, z_ij_p, l_ij_p = (ext[:, k * w:(k + 1) * w].copy() for k in range(4))
        if (not reset) or c_res <= eta * self.c_res_p:
            alpha_p = self.alpha
            self.alpha = 0.5 * (1. + np.sqrt(1 + 4. * alpha_p ** 2))
            if not AMA:
                z_i[:] = z_i + ((alpha_p - 1) / self.alpha) * (z_i - z_i_p)
                z_ij = z_ij + ((alpha_p - 1) / self.alpha) * (z_ij - z_ij_p)
            l_i[:] = l_i + ((alpha_p - 1) / self.alpha) * (l_i - l_i_p)
            l_ij = l_ij + ((alpha_p - 1) / self.alpha) * (l_ij - l_ij_p)
            self.c_res_p = c_res
        else:
            self.alpha = 1.
            z_i[:], l_i[:] = z_i_p, l_i_p
            z_ij, l_ij = z_ij_p, l_ij_p
            self.c_res_p = (1. / eta) * self.c_res_p
        self.z_ij[...] = z_ij[:B].reshape(B, nn, ns)
        self.l_ij[...] = l_ij[:B].reshape(B, nn, ns)
        return np.concatenate([z_ij, l_ij], axis=1)
