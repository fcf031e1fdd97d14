"""`BatchP2P`: B independent point-to-point agents driven through the
receding-horizon loop with everything resident on the device.

The reference loops `Deployer.update` -> `problem.predict / solve / store`
(`execution/deployer.py:43-79`) for one agent in Python.  Here one MPC step of the
whole batch is: (1) ideal prediction -- the initial condition of the next solve is
the current plan evaluated `update_time` ahead (`vehicles/vehicle.py:323-326`,
C++ `Vehicle::predict` Vehicle.cpp:61-80); (2) horizon bookkeeping `t = time since
the last knot crossing` and, on a crossing, the warm-start shift of every `seg`
spline variable (`problems/point2point.py:187-198`, `basics/optilayer.py:470-490`)
plus an index shift of the multipliers; (3) `omgx_batch_solve` with a primal-dual
warm start.  The glue is a handful of tiny tensor ops on [B, *] device arrays
(torch is only the allocator/stream here); no data leaves HBM between steps.

Passing a host solver object as `ops` (anything with the `solve(template, p, x0, ...)` signature of
the tests' oracle port binding) runs the same protocol on host arrays instead: that is how the
parity tests and the CPU baseline of bench.py replay the loop.  This package itself never imports
the oracle; the default `ops='hip'` is the only product path.
"""
import os

import numpy as np

from .splines import shiftoverknot_T, since_knot


# solver options of a knot-crossing step (BatchP2P `cross_options`)
CROSS_OPTIONS = {}


def dual_shift_perm(father, extrapolate=True):
    """perm[r] = row whose multiplier warm-starts row r after the horizon moved by
    one knot interval (-1: none).  Spline-valued constraint entries are indexed by
    B-spline coefficients: moving the horizon by one interval drops the first
    `mult` coefficients (mult = multiplicity of the interior knots of that entry's
    basis); scalar rows keep their multiplier.  The rows that enter at the end of the horizon
    have no predecessor: with `extrapolate` they start from the multiplier of the last row that has
    one (for the terminal-slack rows that is the right magnitude: the objective weight of the last
    coefficients), otherwise from zero (-1) -- a zero multiplier on a row that is active at the
    solution removes that row's curvature from the first Newton system, and the first step of the
    crossing solve is cut to ~1e-6 by the fraction-to-boundary rule."""
    tpl = father.template
    perm = np.arange(tpl.n_con, dtype=np.int64)
    for label, child in father.children.items():
        for cname in child._constraints:
            key = (label, child._add_label(cname))
            off, rows, _ = tpl.con_layout[key]
            if cname not in child._splines_dual:
                continue
            basis = child._splines_dual[cname]['basis']
            interior = basis.knots[(basis.knots > basis.knots[0]) & (basis.knots < basis.knots[-1])]
            if len(interior) == 0:
                continue
            mult = int(np.sum(interior == interior[0]))
            idx = np.arange(rows) + mult
            perm[off:off + rows] = off + np.minimum(idx, rows - 1) if extrapolate else np.where(idx < rows, off + idx, -1)
    return perm


class BatchP2P(object):

    def __init__(self, problem, P, ops='hip', device=None, options=None, update_time=0.1,
                 max_iter_step=None, shift_every_spline=True, straggler_first=True, cross_options=None):
        # max_iter_step: iteration cap of a receding-horizon step (default: the cold-solve cap; an
        # agent that hits it keeps its last strictly feasible iterate and restarts cold next step).
        # shift_every_spline: on a knot crossing shift every spline variable like the generated
        # C++ does (`export/export.py:414-439`); False = the Python rule, names containing 'seg'
        # only (`optilayer.py:482`), which leaves the stale leading coefficient of g* / eps_* (it
        # carries no cost just before the crossing, so the barrier parks it far from its bound).
        # cross_options: solver options of the step right after a knot crossing (the shifted plan sits on the boundary of
        # the rows that enter the horizon and its multipliers are index-shifted: a wider push into the interior and the
        # barrier parameter of the shifted point itself)
        self.cross_options = dict(cross_options) if cross_options is not None else dict(CROSS_OPTIONS)
        self.problem = problem
        father = problem.father
        self.tpl = tpl = father.template
        if len(problem.vehicles) != 1:
            raise NotImplementedError('BatchP2P drives single-vehicle problems (one agent = one vehicle); '
                                      'a problem with %d vehicles needs the prediction of each of them'
                                      % len(problem.vehicles))
        veh = problem.vehicles[0]
        self.veh, self.basis = veh, veh.basis
        self.L, self.n_dim = len(veh.basis), veh.n_dim
        self.T = float(problem.options['horizon_time'])
        self.knot_time = float(problem.knot_time)
        self.update_time = float(update_time)
        self.B = P['p'].shape[0]
        self.o_spl = tpl.entry_range(veh.label, 'splines_seg0', 'var')[0]
        # initial conditions the prediction writes: time derivatives 0, 1, (2) of the plan at the time of the
        # next solve (`vehicles/holonomic.py:88-89,153-159`: state0, input0; `vehicles/quadrotor.py:76-85,110-114`:
        # spl0, dspl0, ddspl0)
        names = ('spl0', 'dspl0', 'ddspl0') if (veh.label, 'spl0') in tpl.par_layout else ('state0', 'input0')
        self.p_offs = [tpl.entry_range(veh.label, nm, 'par')[0] for nm in names]
        self.n_spl = veh.n_spl
        self.o_state0, self.o_input0 = self.p_offs[0], self.p_offs[1]
        self.o_t = tpl.entry_range(problem.label, 't', 'par')[0]
        # obstacle motion model between two solves (`environment/obstacle.py:246-264` without bouncing):
        # the parameters x, v, a of every obstacle are the values AT the time of the solve
        # (`obstacle.py:142-155` reads signals[...][:, -1]; the template extrapolates them back by t)
        self.obst = []
        for obs in problem.environment.obstacles:
            ox = tpl.entry_range(obs.label, 'x', 'par')
            ov = tpl.entry_range(obs.label, 'v', 'par')
            oa = tpl.entry_range(obs.label, 'a', 'par')
            if np.any(P['p'][:, ov[0]:ov[1]] != 0.) or np.any(P['p'][:, oa[0]:oa[1]] != 0.):     # static obstacles: nothing to do
                self.obst.append((ox[0], ov[0], oa[0], ox[1] - ox[0]))
        # (a problem loaded from a bundle, `omgtools.workloads`, brings the multiplier map of a knot crossing with it)
        self.perm = father.dual_perm if hasattr(father, 'dual_perm') else dual_shift_perm(father)
        ents, mats, off = [], [], 0
        for label, name, spl in father.shifted_entries(every_spline=shift_every_spline):
            lo, rows, cols = tpl.var_layout[(label, name)]
            Tm = shiftoverknot_T(spl['basis'])
            ents.append([lo, rows, cols, off])
            mats.append(Tm.reshape(-1))
            off += Tm.size
        self.shift_entries = np.array(ents, dtype=np.int32)
        self.shift_mats = np.concatenate(mats)
        self._shift_dense = [(e, m.reshape(e[1], e[1])) for e, m in zip(ents, mats)]
        self.time = 0.0
        self.under_way, self.stop_tol = None, 1e-3        # (stop_at_arrival)
        # (warm_mu_factor 0.1: a step starts at the barrier parameter the previous solve of the agent ended with, tol / 10,
        # unless the shifted point is far off that central path -- a tenth of its average complementarity then: at tol 1e-3
        # the same iterates as factor 0, at 1e-6 0.03 % instead of 0.4 % of the steps end at the iteration cap)
        self.opts = dict(tol=1e-3, max_iter=300, warm_mu_factor=0.1, warm_z_floor=0.1, warm_z_cap=0.0)
        self.opts.update(options or {})
        self.max_iter_cold = self.opts['max_iter']
        from .backend import DEFAULT_OPTIONS
        self._base_extra = dict((k, self.opts.get(k, DEFAULT_OPTIONS[k])) for k in self.cross_options)   # (what a non-crossing step resets them to)
        self.max_iter_step = int(max_iter_step) if max_iter_step else self.max_iter_cold
        self.kind = 'hip' if ops == 'hip' else 'host'
        self.straggler_first = bool(straggler_first)
        if ops == 'hip':
            import torch
            from .backend import BatchSolver
            self.torch = torch
            self.dev = device if device is not None else torch.device('cuda', 0)
            f64 = dict(dtype=torch.float64, device=self.dev)
            self.solver = BatchSolver(tpl, self.B, device=self.dev.index or 0, options=self.opts)
            self.solver.set_stream(torch.cuda.current_stream().cuda_stream)
            self.p = torch.as_tensor(np.ascontiguousarray(P['p']), **f64)
            self.x = torch.as_tensor(np.ascontiguousarray(P['x0']), **f64)
            self.x_new = torch.empty_like(self.x)
            self.lam = torch.zeros((self.B, tpl.n_con), **f64)
            self.lb, self.ub = torch.as_tensor(tpl.lb, **f64), torch.as_tensor(tpl.ub, **f64)
            self.status = torch.zeros(self.B, dtype=torch.int32, device=self.dev)
            self.iters = torch.zeros(self.B, dtype=torch.int32, device=self.dev)
            pm = np.maximum(self.perm, 0)
            self._perm_idx = torch.as_tensor(pm, dtype=torch.int64, device=self.dev)
            self._perm_ok = torch.as_tensor((self.perm >= 0).astype(np.float64), **f64)
            self._mask = torch.ones(self.B, dtype=torch.uint8, device=self.dev)
            self._order = torch.arange(self.B, dtype=torch.int32, device=self.dev)
        else:
            if isinstance(ops, str):
                raise ValueError("ops must be 'hip' or an injected host solver object (tests / CPU baseline)")
            self.port = ops                          # injected by tests / bench.py's cpu_baseline leg
            self.n_threads = 1
            self.pool = None                         # optional: a host solver with `solve(p, x, lam, status, iters, dw,
            #                                          step=...)` that also runs the step glue per agent (bench.py)
            self.dw = np.zeros(self.B)            # inertia correction carried between warm solves
            self.p, self.x = np.ascontiguousarray(P['p'], dtype=float).copy(), np.ascontiguousarray(P['x0'], dtype=float).copy()
            self.lam = np.zeros((self.B, tpl.n_con))
            self.status = np.zeros(self.B, dtype=np.int32)
            self.iters = np.zeros(self.B, dtype=np.int32)

    # -- solves ------------------------------------------------------------------------
    def _solve(self, warm, events=None, step_desc=None, ordered=False, extra=None):
        extra = extra or {}
        if self.kind == 'hip':
            self.solver.set_options(warm_start=int(warm),
                                    max_iter=self.max_iter_step if warm else self.max_iter_cold, **dict(self._base_extra, **extra))
            if not warm:
                self.lam.zero_()
                self._x_init = self.x.clone()
            if warm and self.straggler_first and not ordered:
                # agents that needed most iterations last time are launched first
                self.solver.order_by_iters(self.iters, self._order)
            if events is not None:                 # timing events of the caller (bench.py): on the solve kernel's own dispatch
                self.solver.set_launch_events(events[0], events[1])
            self.solver.solve_device(self.p, self.x, self.lb, self.ub, self.x_new, self.lam,
                                     self.status, self.iters, bounds_shared=True)
            self.x, self.x_new = self.x_new, self.x
        elif self.pool is not None:
            if self.under_way is not None:
                raise NotImplementedError('stop_at_arrival: not with a host pool (its workers run the step glue themselves)')
            if not warm:
                self.lam[:] = 0.
            self.pool.solve(self.p, self.x, self.lam, self.status, self.iters, self.dw, step=step_desc,
                            **dict(self.opts, warm_start=int(warm), max_iter=self.max_iter_step if warm else self.max_iter_cold, **extra))
        else:
            kw = dict(self.opts, max_iter=self.max_iter_step if warm else self.max_iter_cold, **extra)
            if self.under_way is not None:
                # the stop rule as the solve kernel applies it: the criterion on p ends an agent's loop for good; the agents
                # under way are solved as a batch of their own (independent problems: the same results), the others keep
                # their plan, their multipliers and their status, iters = 0
                self.under_way &= ~self.arrived(self.stop_tol)
                idx = np.flatnonzero(self.under_way)
                self.iters = np.zeros(self.B, dtype=np.int32)
                if len(idx):
                    dw = self.dw[idx].copy()
                    r = self.port.solve(self.tpl, self.p[idx], self.x[idx], lam_g0=self.lam[idx] if warm else None,
                                        status0=self.status[idx] if warm else None, warm_start=int(warm),
                                        n_threads=self.n_threads, dw_state=dw, **kw)
                    self.x[idx], self.lam[idx], self.status[idx], self.iters[idx], self.dw[idx] = r['x'], r['lam_g'], r['status'], r['iters'], dw
                return
            r = self.port.solve(self.tpl, self.p, self.x, lam_g0=self.lam if warm else None,
                                status0=self.status if warm else None, warm_start=int(warm),
                                n_threads=self.n_threads, dw_state=self.dw, **kw)
            self.x, self.lam, self.status, self.iters = r['x'], r['lam_g'], r['status'], r['iters']

    def solve_cold(self, bends=(1.0, -1.0, 2.5, -2.5), fused=True):
        """Cold solve from the reference's initial guess (`get_init_spline_value`: coefficients on the straight
        line).  Agents that do not converge from it (phase I stalls: e.g. 5 % of the Quadrotor class with its five
        moving obstacles) are solved again from the same guess bent sideways by `bends[0]`, then `bends[1]` ... metres
        at mid-course -- a different side of the obstacles; the reference has no such retry (its user would
        re-initialise by hand), `bends=()` switches it off.  fused: the restarts run inside the launch of the first
        attempt (`omgx_batch_set_restarts`), else as separate passes over the failed agents (`restart_failed`) --
        same guesses, same results.  Returns the largest number of restarts an agent needed."""
        if self.kind != 'hip' or not bends:
            self._solve(False)
            return 0
        if not fused:
            self._solve(False)
            return self.restart_failed(bends)
        t = self.torch
        alts = t.stack([self._bent(self.x, s) for s in bends]).contiguous()
        attempts = t.zeros(self.B, dtype=t.int32, device=self.dev)
        self.solver.set_restarts(alts, attempts)
        try:
            self._solve(False)
        finally:
            self.solver.set_restarts(None)
        return int(attempts.max().item())

    def _bent(self, x_first, s):
        """The initial guess x_first with its spline coefficients moved sideways (perpendicular to start -> goal in
        the x-y plane) by s * sin^2(pi * k / (L - 1)) metres."""
        t = self.torch
        L, ns = self.L, self.n_spl
        c = x_first[:, self.o_spl:self.o_spl + ns * L].reshape(self.B, ns, L)
        d = c[:, :, -1] - c[:, :, 0]
        d = d / d.norm(dim=1, keepdim=True).clamp_min(1e-12)
        nrm = t.zeros_like(d)
        nrm[:, 0], nrm[:, 1] = -d[:, 1], d[:, 0]
        nrm = nrm / nrm.norm(dim=1, keepdim=True).clamp_min(1e-12)
        bump = t.sin(t.linspace(0., 1., L, dtype=t.float64, device=self.dev) * np.pi) ** 2
        alt = x_first.clone()
        alt[:, self.o_spl:self.o_spl + ns * L] += (float(s) * nrm[:, :, None] * bump[None, None, :]).reshape(self.B, -1)
        return alt

    def restart_failed(self, bends=(1.0, -1.0, 2.5, -2.5)):
        """Restart passes of a cold solve (see solve_cold); returns how many were needed."""
        passes = 0
        if self.kind != 'hip' or not bends:
            return passes
        for s in bends:
            if bool((self.status == 0).all()):
                break
            alt = self._bent(self._x_init, s)      # (_x_init: the guess the first pass started from)
            # the solved agents sit in self.x (the output of the last pass): they are skipped and keep it
            self.solver.set_options(warm_start=0, max_iter=self.max_iter_cold)
            self.solver.solve_device(self.p, alt, self.lb, self.ub, self.x, self.lam, self.status, self.iters,
                                     bounds_shared=True, only_failed=True)
            passes += 1
        return passes

    # -- one receding-horizon step ---------------------------------------------------------
    def step(self, events=None, before_solve=None):
        """before_solve(self): called between the glue of the step (prediction, obstacles, shift: p and x are what the
        solve will read) and the solve -- where a caller with host buffers uploads its parameters (bench.py's pipelined
        host-boundary leg)."""
        B, L, nd = self.B, self.L, self.n_dim
        t_prev = self.time
        t_now = t_prev + self.update_time
        rel_prev = since_knot(t_prev, self.knot_time)
        # (1) ideal prediction on the current plan, (2) horizon bookkeeping
        tau = (rel_prev + self.update_time) / self.T
        crossed = int(np.round(t_prev / self.knot_time, 6)) < int(np.round(t_now / self.knot_time, 6))
        t_rel = since_knot(t_now, self.knot_time)
        if self.kind == 'host' and self.pool is not None:
            self.time = t_now
            self._solve(True, step_desc=self._pool_step(tau, t_rel, crossed), extra=self.cross_options if crossed else None)
            return crossed
        if self.kind == 'hip':
            # (asked for before the prediction: its launch then carries the ordering as one more workgroup)
            if self.straggler_first:
                self.solver.order_by_iters(self.iters, self._order)
            # one kernel: the initial conditions from the plan at tau and the new t, all written into p
            self.solver.predict_ex(self.x, self.p, self.o_spl, self.n_spl, self.basis.degree, self.basis.knots, tau,
                                   1.0 / self.T, self.p_offs, self.o_t, t_rel)
        else:
            c = self.x[:, self.o_spl:self.o_spl + self.n_spl * L].reshape(B, self.n_spl, L)
            for o, E in enumerate(self._eval_rows(tau)):
                self.p[:, self.p_offs[o]:self.p_offs[o] + self.n_spl] = c @ E
            self.p[:, self.o_t] = t_rel
        # obstacles move on: x <- x + v dt + a dt^2 / 2, v <- v + a dt (a no-op for static obstacles)
        dt = self.update_time
        for ox, ov, oa, nd_o in self.obst:
            px, pv, pa = self.p[:, ox:ox + nd_o], self.p[:, ov:ov + nd_o], self.p[:, oa:oa + nd_o]
            px += dt * pv + (0.5 * dt * dt) * pa
            pv += dt * pa
        if crossed:
            self._shift()
        self.time = t_now
        if before_solve is not None:
            before_solve(self)
        # (3) warm-started solve
        self._solve(True, events, ordered=self.kind == 'hip' and self.straggler_first, extra=self.cross_options if crossed else None)
        return crossed

    def rollout(self, n_steps, iters_log=None, status_log=None):
        """`n_steps` receding-horizon steps of every agent in ONE launch (`omgx_batch_rollout`): per agent the statements of
        `step` -- prediction, obstacles, knot-crossing shift, warm-started solve -- in the same order with the same numbers,
        without the barrier between the steps of different agents (they are independent problems: each vehicle of the
        reference runs its own `Deployer.update` loop).  For simulation / evaluation runs with ideal prediction; a deployment
        that feeds measured states back steps with `step`.  Returns the number of knot crossings."""
        if self.kind != 'hip':
            raise NotImplementedError('rollout is a device launch')
        tau, t_rel, crossed = [], [], []
        t = self.time
        for _ in range(int(n_steps)):                      # (the clock of `step`, statement for statement)
            t_prev, t_now = t, t + self.update_time
            rel_prev = since_knot(t_prev, self.knot_time)
            tau.append((rel_prev + self.update_time) / self.T)
            crossed.append(int(np.round(t_prev / self.knot_time, 6)) < int(np.round(t_now / self.knot_time, 6)))
            t_rel.append(since_knot(t_now, self.knot_time))
            t = t_now
        self.solver.set_options(warm_start=1, max_iter=self.max_iter_step, **self._base_extra)
        if self.straggler_first:
            self.solver.order_by_iters(self.iters, self._order)
        self.solver.rollout(self.p, self.x, self.lb, self.ub, self.lam, self.status, self.iters, tau, t_rel, crossed,
                            self.o_spl, self.n_spl, self.basis.degree, self.basis.knots, 1.0 / self.T, self.p_offs, self.o_t,
                            obstacles=self.obst, dt=self.update_time, shift_entries=self.shift_entries, shift_T=self.shift_mats,
                            lam_perm=self.perm, cross_options=self.cross_options or None, iters_log=iters_log, status_log=status_log)
        self.time = t
        return int(sum(crossed))

    def _eval_rows(self, tau):
        """[E_0, E_1, ...]: c @ E_o = o-th time derivative of the plan at tau."""
        rows = [self.basis.eval_basis([tau])[0]]
        for o in range(1, len(self.p_offs)):
            dbasis, Po = self.basis.derivative(o)
            rows.append(dbasis.eval_basis([tau])[0] @ Po / self.T ** o)
        return rows

    def _pool_step(self, tau, t_rel, crossed):
        """Constants of this step for a pool that runs the glue per agent in its workers (field names of the
        `StepDesc` the pool defines)."""
        desc = self.pool.step_desc()
        rows = [np.ascontiguousarray(r) for r in self._eval_rows(tau)]
        if len(rows) != 2 or self.n_spl != self.n_dim:
            raise NotImplementedError('the pool step glue carries state0 / input0 only')
        E, Ed = rows
        obst = np.ascontiguousarray(np.array(self.obst, dtype=np.int32).reshape(-1, 4))
        perm = np.ascontiguousarray(self.perm, dtype=np.int64)
        ents = np.ascontiguousarray(self.shift_entries, dtype=np.int32)
        mats = np.ascontiguousarray(self.shift_mats, dtype=np.float64)
        desc._keep = (E, Ed, obst, perm, ents, mats)
        desc.o_spl, desc.n_dim, desc.L = self.o_spl, self.n_dim, self.L
        desc.o_state0, desc.o_input0, desc.o_t = self.o_state0, self.o_input0, self.o_t
        desc.t_rel, desc.dt = t_rel, self.update_time
        desc.E, desc.Ed = E.ctypes.data, Ed.ctypes.data
        desc.n_obst, desc.obst = len(self.obst), obst.ctypes.data
        desc.crossed, desc.n_shift = int(crossed), len(ents)
        desc.shift_entries, desc.shift_mats, desc.perm = ents.ctypes.data, mats.ctypes.data, perm.ctypes.data
        return desc

    def _shift(self):
        if self.kind == 'hip':
            self.solver.shift(self.x, self._mask, self.shift_entries, self.shift_mats, device=True)
            self.lam = self.lam.index_select(1, self._perm_idx) * self._perm_ok
        else:
            for (lo, rows, cols, _), Tm in self._shift_dense:
                blk = self.x[:, lo:lo + rows * cols].reshape(self.B, cols, rows)
                self.x[:, lo:lo + rows * cols] = (blk @ Tm.T).reshape(self.B, -1)
            self.lam = np.where(self.perm >= 0, self.lam[:, np.maximum(self.perm, 0)], 0.0)

    # -- the reference's stop criterion ----------------------------------------------------------
    def arrived(self, stop_tol=1e-3):
        """Per agent: the reference's `stop_criterium` (`problems/point2point.py:98-102` -> `vehicles/holonomic.py:145-151`,
        `holonomic3d.py`: |state - poseT| <= stop_tol and |input| <= stop_tol, Euclidean norms, `stop_tol` = 1e-3 by default,
        `vehicles/vehicle.py:72`) on the state the last prediction wrote into p -- the state the vehicle is in at the time of
        the current update.  Boolean tensor (device loop) / array (host loop); the reference's `Simulator.run` ends a vehicle's
        loop at the first update for which this holds (`execution/simulator.py:39-62`).  Point-mass classes (state0 / input0 / poseT)."""
        tpl, veh = self.tpl, self.veh
        if (veh.label, 'poseT') not in tpl.par_layout or (veh.label, 'state0') not in tpl.par_layout:
            raise NotImplementedError('arrived(): the class has no state0 / input0 / poseT parameters')
        nd = self.n_dim
        o_pose = tpl.entry_range(veh.label, 'poseT', 'par')[0]
        st, inp, pose = (self.p[:, o:o + nd] for o in (self.o_state0, self.o_input0, o_pose))
        if self.kind == 'hip':
            return ((st - pose).norm(dim=1) <= stop_tol) & (inp.norm(dim=1) <= stop_tol)
        return (np.linalg.norm(st - pose, axis=1) <= stop_tol) & (np.linalg.norm(inp, axis=1) <= stop_tol)

    def stop_at_arrival(self, stop_tol=1e-3, on=True):
        """End every agent's loop where the reference's does: from now on an agent for which `arrived(stop_tol)` holds at an update
        is not solved at that update or any later one (`execution/simulator.py:39-62` leaves its `while` loop; a fleet's
        vehicles arrive at different updates) -- it keeps its plan, its multipliers and its status, `iters` reads 0.  Device loop:
        the rule is the solve kernel's (`omgx_batch_set_stop`, no launch of its own); `under_way` [B] (int32 tensor / bool array)
        holds who is still running.  The agents under way are solved exactly as without the rule.  `rollout` applies it too: an agent's
        loop inside the launch ends at the step its state meets the criterion (its plan stays as it is at that step)."""
        if not on:
            self.under_way = None
            if self.kind == 'hip':
                self.solver.set_stop(under_way=None)
            return
        tpl, veh = self.tpl, self.veh
        if (veh.label, 'poseT') not in tpl.par_layout or (veh.label, 'state0') not in tpl.par_layout:
            raise NotImplementedError('stop_at_arrival(): the class has no state0 / input0 / poseT parameters')
        self.stop_tol = float(stop_tol)
        if self.kind == 'hip':
            self.under_way = self.torch.ones(self.B, dtype=self.torch.int32, device=self.dev)
            self.solver.set_stop(self.o_state0, self.o_input0, tpl.entry_range(veh.label, 'poseT', 'par')[0], self.n_dim, self.stop_tol,
                                 self.under_way)
        else:
            self.under_way = np.ones(self.B, dtype=bool)

    # -- convenience -----------------------------------------------------------------------
    def host(self, name):
        a = getattr(self, name)
        return a.cpu().numpy() if self.kind == 'hip' else np.asarray(a)


def split_bounds(B, n_streams, slots=None):
    """Contiguous sub-batches [(lo, hi)] of a batch of B agents.  Without `slots`: sizes that differ by at most one.  With the number
    of workgroups the chip holds at once (`slots`): sizes in units of a quarter of it, the larger sub-batches first, the last one
    takes the remainder -- 1024 agents on 512 slots in three sub-batches: 384 / 384 / 256 instead of 342 / 341 / 341.  Measured on
    the benchmark batch (round 5, four runs each on two boxes): 2.29-2.31 M solves/s against 2.20-2.26 M for the even split, and
    against 2.19-2.22 M for 512 / 256 / 256 and 2.21-2.25 M for 416 / 416 / 192: launches whose sizes are multiples of a quarter
    of the resident workgroups leave fewer of them idle when two sub-batches share the chip."""
    from .distributed import shard_range
    even = [shard_range(B, s, n_streams) for s in range(n_streams)]
    unit = (slots or 0) // 4
    if unit < 1 or B < unit * n_streams:
        return even
    units = -(-B // unit)                                   # ceil: the last unit may be a partial one
    per = [units // n_streams + (1 if s < units % n_streams else 0) for s in range(n_streams)]
    sizes = [u * unit for u in per]
    sizes[-1] -= sum(sizes) - B
    if min(sizes) < 1:
        return even
    lo, out = 0, []
    for n in sizes:
        out.append((lo, lo + n))
        lo += n
    return out


_SUB_BATCH_STREAMS = {}


def sub_batch_streams(dev, n):
    """The HIP streams the sub-batches of a device run on: HIGH-PRIORITY streams, created once per device, shared by every
    `StreamedP2P` on it and put to use (an event record) right away.  The HIP runtime maps streams onto `GPU_MAX_HW_QUEUES`
    (four) hardware queues per priority level when they are first used -- a new queue while the level has fewer than four, else
    the least-used one -- so whether three default-priority streams got three queues depended on what the process had created
    before (the null stream, every library handle's own stream, a process group's): measured on the benchmark batch, round 6
    (`profiles/r06_stream_placement.txt`), 1.2 M instead of 2.2 M solves/s when a library handle was created ahead of the first
    `StreamedP2P`, or when a second `StreamedP2P` followed the first in a process -- two sub-batches on one queue run their
    launches one after the other.  Nothing else in a process uses the high-priority level: its first three streams get a
    queue each, whatever came before (2.15-2.20 M in all four creation orders tried, with and without a process group)."""
    import torch
    key = (dev.type, dev.index or 0)
    have = _SUB_BATCH_STREAMS.setdefault(key, [])
    while len(have) < n:
        st = torch.cuda.Stream(device=dev, priority=int(os.environ.get('OMGX_STREAM_PRIORITY', '-1')))
        if not os.environ.get('OMGX_NO_STREAM_TOUCH'):      # (developer knob: the order-dependent behaviour of rounds 5 / 6)
            torch.cuda.Event().record(st)
        have.append(st)
    return have[:n]


class StreamedP2P(object):
    """The batch as `n_streams` sub-batches, each a `BatchP2P` with its own library handle on its own HIP stream.  The problems of
    a point-to-point batch are independent, so nothing orders the steps of one sub-batch against those of another: while one waits
    for a straggler of its step, the next step of the other fills the idle workgroup slots (1024 agents, two streams: 2.07 M
    solves/s against 1.74 M on one stream; more streams lose again -- every handle launches a full grid of persistent workgroups).
    Per agent the same launches in the same order as `BatchP2P`: the same bits (tests/test_gpu_rollout.py).  Since round 5 this is
    what `receding_horizon_batch` hands out for a batch of at least two rounds of resident workgroups: the per-step product path.
    `step` returns whether the step crossed a knot; `x, p, lam, status, iters` join the streams and concatenate the sub-batches
    (`gather`); `parts[k]` / `streams[k]` give the sub-batches to callers that attach events, statistics or copies per stream."""

    def __init__(self, problem, P, n_streams=2, device=None, slots=None, **kw):
        import torch
        self.torch = torch
        B = P['p'].shape[0]
        if n_streams < 1 or n_streams > B:
            raise ValueError('%d agents do not split into %d sub-batches' % (B, n_streams))
        self.bounds = split_bounds(B, n_streams, slots)
        self.dev = device if device is not None else torch.device('cuda', 0)
        self.streams = sub_batch_streams(self.dev, n_streams) if not os.environ.get('OMGX_NO_STREAM_TOUCH') else \
            [torch.cuda.Stream(device=self.dev) for _ in range(n_streams)]
        self.parts = []
        # (the caller's stream may still be writing what the sub-batches read, and the other way round at the end)
        ready = torch.cuda.current_stream(self.dev).record_event()
        for (lo, hi), st in zip(self.bounds, self.streams):
            Ps = dict(P, p=P['p'][lo:hi], x0=P['x0'][lo:hi])
            st.wait_event(ready)
            with torch.cuda.stream(st):
                self.parts.append(BatchP2P(problem, Ps, ops='hip', device=self.dev, **kw))
        self.B = B
        self.kind = 'hip'
        self.tpl, self.problem = self.parts[0].tpl, problem

    def _each(self, fn):
        out = []
        for part, st in zip(self.parts, self.streams):
            with self.torch.cuda.stream(st):
                out.append(fn(part))
        return out

    def solve_cold(self, **kw):
        return max(self._each(lambda m: m.solve_cold(**kw)))

    def restart_failed(self, *a, **kw):
        return max(self._each(lambda m: m.restart_failed(*a, **kw)))

    def step(self, events=None, before_solve=None):
        """events: one (start, stop) pair per sub-batch (stamped on that sub-batch's solve kernel)."""
        evs = events if events is not None else [None] * len(self.parts)
        hooks = before_solve if isinstance(before_solve, (list, tuple)) else [before_solve] * len(self.parts)
        out = []
        for part, st, ev, hk in zip(self.parts, self.streams, evs, hooks):
            with self.torch.cuda.stream(st):
                out.append(part.step(events=ev, before_solve=hk))
        return any(out)

    def stop_at_arrival(self, stop_tol=1e-3, on=True):
        for m in self.parts:
            m.stop_at_arrival(stop_tol, on)

    @property
    def under_way(self):
        if self.parts[0].under_way is None:
            return None
        return self.gather('under_way')

    @property
    def time(self):
        return self.parts[0].time

    @time.setter
    def time(self, t):
        for m in self.parts:
            m.time = t

    def synchronize(self):
        for st in self.streams:
            st.synchronize()

    def join(self):
        """The caller's current stream waits for everything enqueued on the sub-batches' streams (no host sync)."""
        cur = self.torch.cuda.current_stream(self.dev)
        for st in self.streams:
            cur.wait_stream(st)

    def fork(self):
        """The sub-batches' streams wait for what the caller's current stream has enqueued so far (e.g. a reset of x / p)."""
        cur = self.torch.cuda.current_stream(self.dev)
        for st in self.streams:
            st.wait_stream(cur)

    def gather(self, name):
        self.join()
        return self.torch.cat([getattr(m, name) for m in self.parts])

    def load(self, x=None, p=None):
        """x / p of the whole batch -> the sub-batches (device tensors [B, *]), ordered behind the caller's stream."""
        self.fork()
        for (lo, hi), m, st in zip(self.bounds, self.parts, self.streams):
            with self.torch.cuda.stream(st):
                if x is not None:
                    m.x.copy_(x[lo:hi])
                if p is not None:
                    m.p.copy_(p[lo:hi])

    x = property(lambda self: self.gather('x'))
    p = property(lambda self: self.gather('p'))
    lam = property(lambda self: self.gather('lam'))
    status = property(lambda self: self.gather('status'))
    iters = property(lambda self: self.gather('iters'))

    def host(self, name):
        return self.gather(name).cpu().numpy()

    def close(self):
        self.synchronize()
        for m in self.parts:
            m.solver.close()


# Sub-batches of the per-step product path: three, on streams with a hardware queue each (`sub_batch_streams`).  Measured on the
# 1024-agent batch with the streams in their own (high-priority) queue pool, round 6: two 2.09 M, three 2.20-2.25 M, four
# 2.22-2.25 M (p50 step latency 0.75 / 0.66-0.91 / 0.97-1.01 ms), five 1.42 M solves/s -- the pool has four queues, a fifth stream
# shares one and its launches queue behind another sub-batch's.
PRODUCT_PATH_STREAMS = 3


def product_path_streams(process_group=None, hw_queues=None):
    """Sub-batches of the per-step product path: `PRODUCT_PATH_STREAMS`.  (While the sub-batch streams came from the runtime's
    common pool of four hardware queues the count depended on what else the process had created -- a `torch.distributed` process
    group's stream, a library handle ahead of the first `StreamedP2P` -- and this function picked two or four under a process
    group; with the streams in a queue pool of their own, `sub_batch_streams`, three is measured the same with and without a group
    and with four or eight hardware queues: 2.20 M in each case.  The arguments are kept for callers of that version.)"""
    return PRODUCT_PATH_STREAMS


def receding_horizon_batch(problem, P, device=None, n_streams='auto', **kw):
    """The per-step product path for a batch of independent agents: a `BatchP2P`, or -- when the batch is at least two rounds
    of resident workgroups (1024 agents of config 2 on 512 slots) -- the same batch as `PRODUCT_PATH_STREAMS` stream-ordered
    sub-batch launches per step (`StreamedP2P`): a step of the whole batch is quantised in rounds of the resident workgroups and
    one straggler costs the batch a whole extra round; with the sub-batches on their own streams the next step
    of one fills the slots the stragglers of the others leave idle.  Per agent the same launches, the same bits."""
    import torch
    dev = device if device is not None else torch.device('cuda', 0)
    B = P['p'].shape[0]
    if n_streams == 'auto' or n_streams <= 1:
        whole = BatchP2P(problem, P, ops='hip', device=dev, **kw)
        # (the launch grid of the handle = the workgroups the chip holds at once, capped at the batch)
        slots = whole.solver.workspace()['n_slabs']
        if n_streams != 'auto' or B < 2 * slots:
            return whole
        whole.solver.close()
        return StreamedP2P(problem, P, n_streams=product_path_streams(), device=dev, slots=slots, **kw)
    return StreamedP2P(problem, P, n_streams=n_streams, device=dev, **kw)
