# This file is derived from OMG-tools (meco-group/omg-tools, `omgtools/problems/formation.py`, `admm.py`, `dualmethod.py`, `distributedproblem.py` (API surface)).
#
# OMG-tools -- Optimal Motion Generation-tools
# Copyright (C) 2016 Ruben Van Parys & Tim Mercy, KU Leuven.
# All rights reserved.
#
# OMG-tools is free software; you can redistribute it and/or
# modify it under the terms of the GNU Lesser General Public
# License as published by the Free Software Foundation; either
# version 3 of the License, or (at your option) any later version.
# This software is distributed in the hope that it will be useful,
# but WITHOUT ANY WARRANTY; without even the implied warranty of
# MERCHANTABILITY or FITNESS FOR A PARTICULAR PURPOSE. See the GNU
# Lesser General Public License for more details.
#
# You should have received a copy of the GNU Lesser General Public
# License along with this program; if not, write to the Free Software
# Foundation, Inc., 51 Franklin Street, Fifth Floor, Boston, MA 02110-1301 USA
#
# Modifications: the public classes, option names, method order and messages of the files named
# above are kept so that scripts written for OMG-tools run unchanged where the original package is
# not installed (benchmark and test tiers of this repository); the CasADi expression layer underneath
# is replaced by explicit polynomials (symbolic.py) and the solver call by the HIP path (backend.py).
# Where the original package IS installed, use omgx_shim instead: it runs the original classes themselves.

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
from .splines import shift_knot1_fwd, shiftfirstknot_T, shiftoverknot_T, since_knot
from .consensus import (coupling_matrix, zupdate_matrices, FormationLayout, circular_neighbors, reverse_slots,      # noqa: F401
                        shift_tables)
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

        def fwd(vec):                      # only the future piece of the spline is penalised
            return shift_knot1_fwd(vec, basis, t0)
        # As executed, the reference transforms x only: `self._transform_spline([x_i, z_i, l_i], tf, self.q_i)`
        # (`admm.py:87-88`) changes the dict of x in place, but for the struct-valued z and l it builds and
        # returns NEW structs (`dualmethod.py:154-157`) that the caller drops -- z and l enter the objective as
        # stored.  Pinned by tests/test_golden_admm.py against the reference's own graphs.
        obj = Poly()
        for k in range(n_dim):
            x = fwd(np.asarray(center[k].coeffs, dtype=object))
            pairs = [(z_i[k * L:(k + 1) * L], l_i[k * L:(k + 1) * L])]
            for j in range(n_nghb):
                o = j * ns + k * L
                pairs.append((z_ji[o:o + L], l_ji[o:o + L]))
            for z, l in pairs:
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


class FormationPoint2point(object):
    """Drop-in for the reference's `FormationPoint2point` (`problems/formation.py:26-72` on top of
    `ADMMProblem` `problems/admm.py:556-628`, `DualProblem` `problems/dualmethod.py:190-244`,
    `DistributedProblem` `problems/distributedproblem.py:26-103`): same constructor, options
    (`rho`, `init_iter`, `max_iter_per_update`, `horizon_time`, ...) and the methods `Simulator` /
    `Deployer` call (`init, reinitialize, initialize, predict, solve, store, simulate,
    stop_criterium, final`).

    One sub-problem per vehicle with its own copy of the environment keeps the reference's host
    bookkeeping (parameters from predictions, trajectory storage); all agents share ONE x-update
    template, and a dual update = one `BatchADMM.iterate` over the whole fleet on the device
    (`omgx_batch_solve` + `omgx_admm_*`).  `ops` selects the kernel backend: 'hip' (the product
    path) or an object with the `HipAdmmOps` interface (tests inject the numpy oracle)."""

    def __init__(self, fleet, environment, options=None, ops='hip'):
        from .vehicles import get_fleet_vehicles
        self.fleet, self.vehicles = get_fleet_vehicles(fleet)
        self.environment = environment
        self.options = {'verbose': 2, 'horizon_time': 10., 'rho': 2., 'init_iter': 5,
                        'max_iter_per_update': 1, 'max_iter': None, 'solver': 'ipopt',
                        'nesterov_acceleration': False, 'eta': 0.999, 'nesterov_reset': False, 'AMA': False,
                        'solver_options': {'ipopt': {'ipopt.tol': 1e-3}}}
        for key, value in (options or {}).items():
            if key == 'solver_options':
                for k2, v2 in value.items():
                    self.options['solver_options'].setdefault(k2, {}).update(v2)
            else:
                self.options[key] = value
        self._ops_kind = ops
        self.iteration, self.update_times = 0, []
        self.residuals = {'primal': [], 'dual': [], 'combined': []}
        self.start_time = 0.

    # -- construction ---------------------------------------------------------------------
    def init(self):
        from .admm import BatchADMM
        N = len(self.vehicles)
        index = {veh: l for l, veh in enumerate(self.vehicles)}
        nbr = [[index[g] for g in self.fleet.get_neighbors(veh)] for veh in self.vehicles]
        n_nghb = len(nbr[0])
        if any(len(r) != n_nghb for r in nbr):
            raise ValueError('every vehicle needs the same number of neighbours (one shared x-update template)')
        self.nbr = np.array(nbr, dtype=np.int32)
        sub_opts = {k: v for k, v in self.options.items()
                    if k not in ('rho', 'init_iter', 'max_iter_per_update', 'max_iter', 'nesterov_acceleration',
                                 'eta', 'nesterov_reset', 'AMA')}
        sub_opts['verbose'] = 0
        self.subs = []
        for veh in self.vehicles:
            problem, updater, father = self._build_template(veh, self.environment.copy(), n_nghb, sub_opts)
            veh._values['rel_pos_c'] = np.asarray(veh.rel_pos_c, dtype=float).reshape(-1, 1)
            self.subs.append((problem, updater, father))
        tpl = self.subs[0][2].template
        for _, _, father in self.subs[1:]:
            t = father.template
            if (t.n_var, t.n_con, t.n_par, t.n_terms) != (tpl.n_var, tpl.n_con, tpl.n_par, tpl.n_terms):
                raise ValueError('vehicles of one formation must share the x-update structure')
        self.tpl = tpl
        self.lay = lay = self._make_layout(tpl, self.vehicles[0], self.subs[0][0], self.subs[0][1], n_nghb)
        self.knot_time = self.subs[0][0].knot_time
        # parameter columns owned by the device-side consensus state (never overwritten from the host)
        keep = np.ones(tpl.n_par, dtype=bool)
        for off, size in ((lay.p_zi, lay.ns), (lay.p_li, lay.ns), (lay.p_zji, n_nghb * lay.ns),
                          (lay.p_lji, n_nghb * lay.ns)):
            keep[off:off + size] = False
        self.host_cols = np.nonzero(keep)[0]
        self._shift_x, self._shift_p, self._shift_side = shift_tables(self.subs[0][2], tpl, lay, self.vehicles[0].basis,
                                                                       self._consensus_is_spline)
        # device state
        p0, x0 = self._host_parameters(0.), self._host_variables()
        if self._ops_kind == 'hip':
            import torch
            from .admm import HipAdmmOps
            from .backend import BatchSolver, options_from_problem
            opts = options_from_problem(self.options)
            opts['tol'] = self.xupdate_tol()
            # (x-updates at 1e-6: the plain multiplier floor of a warm start, no slack cap -- over 24 updates of the host
            # study one x-update ends at the iteration cap with it, two with the cap; at 1e-3, the bench, the cap saves
            # three iterations per x-update: include/omgx.h warm_z_cap)
            opts['warm_z_cap'] = 0.0
            # (and plain backtracking: at 1e-6 a second-order correction changes which x-updates end in the rounding noise of the
            # merit function -- 92 instead of 33 of 660 x-updates of examples/formation_holonomic.py on the host build)
            opts['max_soc'] = 0
            self.solver = BatchSolver(tpl, N, options=opts)
            self.ops = HipAdmmOps(self.solver, tpl, lay, p0, x0, torch.device('cuda', 0))
        else:
            self.ops = self._ops_kind(tpl, lay, p0, x0, self.xupdate_tol())
        self.admm = BatchADMM(lay, self.nbr, self.ops, rho=self.options['rho'],
                              horizon_time=self.options['horizon_time'],
                              **{k: self.options[k] for k in ('nesterov_acceleration', 'eta', 'nesterov_reset', 'AMA')})
        return 0.

    # -- what a derived problem class (rendezvous.RendezVous) replaces ------------------------------------
    _consensus_is_spline = True

    def _build_template(self, vehicle, environment, n_nghb, options):
        return build_updx_template(vehicle, environment, n_nghb, options)

    def _make_layout(self, tpl, vehicle, problem, updater, n_nghb):
        return FormationLayout(tpl, vehicle, problem, updater, n_nghb)

    def xupdate_tol(self):
        """Tolerance of the x-update solves: `ipopt.tol` x 1e-3.  IPOPT's last (superlinear) step
        usually lands orders of magnitude below its tolerance while this solver stops right at it, and
        ADMM with one iteration per update is sensitive to inexact x-updates: on
        `examples/formation_holonomic.py` (rho = 1) the formation error in the passage between the
        obstacles is 0.41 m with x-updates at 1e-4, 0.11 m at 1e-6; at 1e-3 the consensus even keeps
        a residual velocity just above the vehicles' `stop_tol` and the run never ends."""
        from .backend import options_from_problem
        return 1e-3 * options_from_problem(self.options).get('tol', 1e-3)

    device_predictions = 0

    def _device_prediction(self, current_time, update_time, crossing):
        """Whether this update's initial conditions can be predicted on the device: every vehicle predicts ideally, the update
        moves the clock forward (not a start-up iteration), and no vehicle was handed another state, input or target since the
        last update (option 'device_prediction': True by default, False = always pack on the host)."""
        if not self.options.get('device_prediction', True) or not hasattr(self.ops, 'predict'):
            return False
        if not (current_time > self._time_prev + 1e-9) or self.iteration < self.options['init_iter']:
            return False
        if not all(v.options.get('ideal_prediction', False) for v in self.vehicles):
            return False
        if not hasattr(self, '_shared_cols'):
            tpl, lay = self.tpl, self.lay
            per_agent = np.zeros(tpl.n_par, dtype=bool)
            per_agent[self.host_cols] = True
            own = np.zeros(tpl.n_par, dtype=bool)
            veh = self.vehicles[0]
            for (label, name), (off, r, c) in tpl.par_layout.items():
                if label == veh.label or (label, name) == (self.subs[0][1].label, 'rho'):
                    own[off:off + r * c] = True                      # a vehicle's own entries (state0, input0, poseT, rel_pos_c) and rho
            own[lay.p_t] = True
            self._shared_cols = np.nonzero(per_agent & ~own)[0]
            self._o_plan = tpl.entry_range(veh.label, 'splines_seg0', 'var')[0]
            self._targets = None
        # a new target (`set_terminal_conditions`) or a state / input handed in from outside since the last update
        # (`set_initial_conditions`, `overrule_state`, `overrule_input`: every such call bumps the vehicle's `_ic_version`):
        # back to the host for this update -- the resident plan does not know about it
        targets = np.array([np.asarray(v.poseT, float) for v in self.vehicles])
        handed_in = [getattr(v, '_ic_version', 0) for v in self.vehicles]
        if (self._targets is None or targets.shape != self._targets.shape or not np.array_equal(targets, self._targets)
                or handed_in != getattr(self, '_handed_in', None)):
            self._targets, self._handed_in = targets, handed_in
            return False
        return True

    def _host_parameters(self, current_time):
        return np.stack([father.set_parameters(current_time).cat for _, _, father in self.subs])

    def _host_variables(self):
        return np.stack([np.asarray(father.get_variables()).reshape(-1) for _, _, father in self.subs])

    def reinitialize(self, father=None):
        for problem, _, fa in self.subs:
            problem.reinitialize(father=fa)
        self.ops.upload_x(self._host_variables())
        self.ops.upload_params(self._host_parameters(0.), self.host_cols)
        self.admm.initialize()

    def initialize(self, current_time):
        self.start_time = current_time
        for problem, _, _ in self.subs:
            problem.initialize(current_time)
        self._time_prev = 0.
        for _ in range(self.options['init_iter']):
            self.solve(current_time, 0.0)

    # -- the hot call --------------------------------------------------------------------------
    def solve(self, current_time, update_time):
        current_time -= self.start_time
        for _ in range(self.options['max_iter_per_update']):
            self.dual_update(current_time, update_time)

    def dual_update(self, current_time, update_time):
        import time
        t0 = time.time()
        # knot crossing: shift the warm start of x and the whole consensus state (`admm.py:477-491`)
        crossing = int(np.round(self._time_prev / self.knot_time, 6)) < int(np.round(current_time / self.knot_time, 6))
        t_rel = since_knot(current_time, self.knot_time)
        if self._device_prediction(current_time, update_time, crossing):
            # Nobody disturbs the vehicles (`ideal_prediction`, `vehicles/vehicle.py:323-326`): the initial conditions of this
            # update are the current plan `update_time` ahead -- one launch on the resident plan (`FormationMPC.step`) instead
            # of packing every vehicle's parameter vector on the host.  What the fleet shares (obstacle motion, T) comes from
            # ONE sub-problem's parameters and is written to all rows; rel_pos_c, poseT and rho do not change between updates.
            rel_prev = since_knot(self._time_prev, self.knot_time)
            tau = (rel_prev + (current_time - self._time_prev)) / float(self.options['horizon_time'])
            veh = self.vehicles[0]
            self.ops.predict(self._o_plan, veh.n_spl, veh.basis, tau, 1.0 / float(self.options['horizon_time']),
                             [self.lay.p_state0, self.lay.p_input0], self.lay.p_t, t_rel)
            row = self.subs[0][2].set_parameters(current_time).cat
            self.ops.upload_params(np.tile(np.asarray(row, float)[None], (len(self.subs), 1)), self._shared_cols)
            if crossing:
                self.ops.shift(self._shift_x, self._shift_p, self._shift_side)
            self.device_predictions += 1
        else:
            if crossing:
                self.ops.shift(self._shift_x, self._shift_p, self._shift_side)
            self.ops.upload_params(self._host_parameters(current_time), self.host_cols)
        self._time_prev = current_time
        status, (pr, dr, cr) = self.admm.iterate(t_rel)
        x = self.ops.download_x()
        for l, (_, _, father) in enumerate(self.subs):
            father.set_variables(x[l])
        self.iteration += 1
        t_upd = time.time() - t0
        if self.options['verbose'] >= 2:
            if (self.iteration - 1) % 20 == 0:
                print('----|------|----------|----------|----------')
                print('%3s | %4s | %8s | %8s | %8s ' % ('It', 't', 'prim res', 'dual res', 't upd'))
                print('----|------|----------|----------|----------')
            print('%3d | %4.1f | %.2e | %.2e | %.2e ' % (self.iteration, current_time, pr, dr, t_upd))
        for key, val in (('primal', pr), ('dual', dr), ('combined', cr)):
            self.residuals[key].append(val)
        self.update_times.append(t_upd)
        self.last_status = np.asarray(status.cpu() if hasattr(status, 'cpu') else status)

    # -- deployment (per-vehicle bookkeeping of the sub-problems) -----------------------------------
    def predict(self, current_time, predict_time, sample_time, states=None, inputs=None,
                dinputs=None, delay=0, enforce_states=False, enforce_inputs=False):
        n = len(self.vehicles)
        per = lambda v: [None] * n if v is None else v
        states, inputs, dinputs = per(states), per(inputs), per(dinputs)
        if current_time == self.start_time:
            enforce_states = True
        for k, vehicle in enumerate(self.vehicles):
            vehicle.predict(current_time, predict_time, sample_time, states[k], inputs[k], dinputs[k],
                            delay, enforce_states, enforce_inputs)

    def store(self, current_time, update_time, sample_time):
        for problem, _, _ in self.subs:
            problem.store(current_time, update_time, sample_time)

    def simulate(self, current_time, simulation_time, sample_time):
        horizon_time = self.options['horizon_time']
        rel = since_knot(current_time - self.start_time, self.knot_time)
        simulation_time = min(simulation_time, horizon_time - rel)
        for vehicle in self.vehicles:
            vehicle.simulate(simulation_time, sample_time)
        self.environment.simulate(simulation_time, sample_time)
        for problem, _, _ in self.subs:
            problem.environment.simulate(simulation_time, sample_time)

    def stop_criterium(self, current_time, update_time):
        if self.options['max_iter'] and self.iteration > self.options['max_iter']:
            return True
        return all(vehicle.check_terminal_conditions() for vehicle in self.vehicles)

    def final(self):
        if self.options['verbose'] >= 1:
            print('\nWe reached our target!')
            print('%-18s %6g ms' % ('Max update time:', max(self.update_times) * 1000.))
            print('%-18s %6g ms' % ('Av update time:', sum(self.update_times) * 1000. / len(self.update_times)))

    def get_interaction_error(self):
        """Average deviation of the agents' fleet centres from their mean (`formation.py:79-100`)."""
        x_i = np.asarray(self.ops.center(self.lay).cpu() if hasattr(self.ops.center(self.lay), 'cpu')
                         else self.ops.center(self.lay))
        return float(np.abs(x_i - x_i.mean(axis=0)).mean())

    def plot(self, *args, **kwargs):
        pass
