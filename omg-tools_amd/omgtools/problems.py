# This file is derived from OMG-tools (meco-group/omg-tools, `omgtools/problems/problem.py`, `point2point.py`).
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

"""`Problem`, `Point2point` (fixed-T) -- the receding-horizon solve.

Behavioural spec: reference `problems/problem.py` (options 54-74, init 85-91,
solve 103-136, predict 138-163, reset_init_guess 165-181, simulate 187-192) and
`problems/point2point.py` (factory 28-35, construct 53-62, init constraints
64-70, FixedTPoint2point 126-266).  `solve()` keeps the reference's call shape
`self.problem(x0=, p=, lbg=, ubg=) -> {'x','lam_g'}` + `stats()['return_status']`;
behind it sits the HIP batch solver (backend.py) instead of CasADi/IPOPT.
"""
import time

import numpy as np

from .opti import OptiChild, OptiFather, inf
from .plotting import PlotLayer
from .vehicles import get_fleet_vehicles
from .splines import definite_integral, evalspline, shiftoverknot_T


class Problem(OptiChild, PlotLayer):

    def __init__(self, fleet, environment, options=None, label='problem'):
        OptiChild.__init__(self, label)
        PlotLayer.__init__(self)
        self.fleet, self.vehicles = get_fleet_vehicles(fleet)
        self.environment = environment
        self.set_default_options()
        self.set_options(options or {})
        self.iteration = 0
        self.update_times = []
        children = [vehicle for vehicle in self.vehicles]
        children += [obstacle for obstacle in self.environment.obstacles]
        children += [self, self.environment]
        self.father = OptiFather(children)

    def set_default_options(self):
        self.options = {'verbose': 2, 'solver': 'ipopt'}
        # the option names of the reference are kept; the HIP interior-point
        # solver maps 'ipopt.tol' / 'ipopt.max_iter' onto its own settings
        self.options['solver_options'] = {'ipopt': {
            'ipopt.tol': 1e-3, 'ipopt.warm_start_init_point': 'yes',
            'ipopt.print_level': 0, 'print_time': 0,
            'ipopt.fixed_variable_treatment': 'make_constraint'}}
        self.options['codegen'] = {'build': None, 'flags': '-O0'}

    def set_options(self, options):
        if 'solver_options' in options:
            for key, value in options['solver_options'].items():
                self.options['solver_options'].setdefault(key, {}).update(value)
        if 'codegen' in options:
            self.options['codegen'].update(options['codegen'])
        for key in options:
            if key not in ['solver_options', 'codegen']:
                self.options[key] = options[key]

    # -- construction -------------------------------------------------------------
    def construct(self):
        self.environment.init()
        for vehicle in self.vehicles:
            vehicle.init()

    def init(self):
        self.father.reset()
        with self.father.table:
            self.construct()
            self.problem, buildtime = self.father.construct_problem(self.options)
        self.father.init_transformations(self.init_primal_transform,
                                         self.init_dual_transform)
        return buildtime

    def init_primal_transform(self, basis):
        return None

    init_dual_transform = init_primal_transform

    def init_step(self, current_time, update_time):
        pass

    def initialize(self, current_time):
        pass

    def reinitialize(self, father=None):
        father = self.father if father is None else father
        father.init_variables()
        father.init_parameters()

    # -- the hot call ----------------------------------------------------------------
    def solve(self, current_time, update_time):
        current_time -= self.start_time
        self.init_step(current_time, update_time)
        var = self.father.get_variables()
        par = self.father.set_parameters(current_time)
        lb, ub = self.father.update_bounds(current_time)
        t0 = time.time()
        result = self.problem(x0=var, p=par, lbg=lb, ubg=ub)
        t_upd = time.time() - t0
        self.father.set_variables(result['x'])
        self.father.set_dual_variables(result['lam_g'])
        stats = self.problem.stats()
        if stats['return_status'] != 'Solve_Succeeded':
            if stats['return_status'] == 'Maximum_CpuTime_Exceeded':
                if current_time != 0.0:
                    print('Maximum solving time exceeded, resetting initial guess')
                    self.reset_init_guess()
                    print(stats['return_status'])
            else:
                print(stats['return_status'])
        if self.options['verbose'] >= 2:
            self.iteration += 1
            if (self.iteration - 1) % 20 == 0:
                print("----|------------|------------")
                print("%3s | %10s | %10s " % ("It", "t upd", "time"))
                print("----|------------|------------")
            print("%3d | %.4e | %.4e " % (self.iteration, t_upd, current_time))
        self.update_times.append(t_upd)

    def predict(self, current_time, predict_time, sample_time, states=None, inputs=None,
                dinputs=None, delay=0, enforce_states=False, enforce_inputs=False):
        n = len(self.vehicles)

        def per_vehicle(values):
            if values is None:
                return [None] * n
            if n == 1 and (not isinstance(values, list) or isinstance(values[0], float)):
                return [values]
            return values
        states, inputs, dinputs = per_vehicle(states), per_vehicle(inputs), per_vehicle(dinputs)
        if current_time == self.start_time:
            enforce_states = True
        for k, vehicle in enumerate(self.vehicles):
            vehicle.predict(current_time, predict_time, sample_time, states[k], inputs[k],
                            dinputs[k], delay, enforce_states, enforce_inputs)

    def reset_init_guess(self, init_guess=None):
        if init_guess is None:
            init_guess = [vehicle.get_init_spline_value() for vehicle in self.vehicles]
        elif not isinstance(init_guess, list):
            init_guess = [init_guess]
        for k, vehicle in enumerate(self.vehicles):
            guess = init_guess[k] if isinstance(init_guess[k], list) else [init_guess[k]]
            if len(guess) != vehicle.n_seg:
                raise ValueError('Each spline segment of the vehicle should receive an '
                                 'initial guess.')
            for l in range(vehicle.n_seg):
                if guess[l].shape[1] != vehicle.n_spl:
                    raise ValueError('Each vehicle spline should receive an initial guess.')
                self.father.set_variables(guess[l], child=vehicle, name='splines_seg' + str(l))

    def simulate(self, current_time, simulation_time, sample_time):
        for vehicle in self.vehicles:
            vehicle.simulate(simulation_time, sample_time)
        self.environment.simulate(simulation_time, sample_time)
        self.fleet.update_plots()
        self.update_plots()

    def stop_criterium(self, current_time, update_time):
        return False

    def final(self):
        pass

    def store(self, current_time, update_time, sample_time):
        pass


class Point2point(object):
    """Factory selecting fixed-T vs free-T (`point2point.py:28-35`)."""

    def __new__(cls, fleet, environment, options=None, freeT=False):
        if freeT:
            return FreeTPoint2point(fleet, environment, options)
        return FixedTPoint2point(fleet, environment, options)


class Point2pointProblem(Problem):

    def __init__(self, fleet, environment, options):
        Problem.__init__(self, fleet, environment, options, label='p2p')
        self.init_time = None
        self.start_time = 0.

    def set_default_options(self):
        Problem.set_default_options(self)
        self.options['inter_vehicle_avoidance'] = False

    def define_time(self):
        """T, t and t0 = t/T (`point2point.py:53-55`); FreeT overrides."""
        self.T, self.t = self.define_parameter('T'), self.define_parameter('t')
        self.t0 = self.t / self.T

    def construct(self):
        self.define_time()
        Problem.construct(self)
        for vehicle in self.vehicles:
            splines = vehicle.define_splines(n_seg=1)
            vehicle.define_trajectory_constraints(splines[0], self.T)
            self.environment.define_collision_constraints(vehicle, splines, self.T)
        if len(self.vehicles) > 1 and self.options['inter_vehicle_avoidance']:
            self.environment.define_intervehicle_collision_constraints(self.vehicles, self.T)

    def define_init_constraints(self):
        for vehicle in self.vehicles:
            for spline, condition in vehicle.get_initial_constraints(vehicle.splines[0], self.T):
                self.define_constraint(evalspline(spline, self.t0) - condition, 0., 0.)

    def initialize(self, current_time):
        self.start_time = current_time

    def reinitialize(self, father=None):
        father = self.father if father is None else father
        Problem.reinitialize(self)
        for vehicle in self.vehicles:
            init = vehicle.get_init_spline_value()
            for k in range(vehicle.n_seg):
                father.set_variables(init[k], vehicle, 'splines_seg' + str(k))

    def set_init_time(self, time):
        self.init_time = time

    def reset_init_time(self):
        self.init_time = None

    def stop_criterium(self, current_time, update_time):
        return all(vehicle.check_terminal_conditions() for vehicle in self.vehicles)

    def final(self):
        self.reset_init_time()
        obj = self.compute_objective()
        if self.options['verbose'] >= 1:
            print('\nWe reached our target!')
            print('%-18s %6g' % ('Objective:', obj))
            print('%-18s %6g ms' % ('Max update time:', max(self.update_times) * 1000.))
            print('%-18s %6g ms' % ('Av update time:',
                                    sum(self.update_times) * 1000. / len(self.update_times)))

    def compute_objective(self):
        raise NotImplementedError('Please implement this method!')


class FixedTPoint2point(Point2pointProblem):

    def __init__(self, fleet, environment, options):
        Point2pointProblem.__init__(self, fleet, environment, options)
        self.objective = 0.
        if self.vehicles[0].knot_intervals is None:
            raise ValueError('A constant knot interval should be used for a fixed T '
                             'point2point problem.')
        self.knot_time = (int(self.options['horizon_time'] * 1000.) /
                          self.vehicles[0].knot_intervals) / 1000.

    def set_default_options(self):
        Point2pointProblem.set_default_options(self)
        self.options['horizon_time'] = 10.
        self.options['hard_term_con'] = False
        self.options['no_term_con_der'] = False

    def construct(self):
        Point2pointProblem.construct(self)
        self.define_init_constraints()
        self.define_terminal_constraints()

    def define_terminal_constraints(self):
        objective = 0.
        self.term_con_len = []
        for vehicle in self.vehicles:
            term_con, term_con_der = vehicle.get_terminal_constraints(vehicle.splines[0])
            if self.options.get('no_term_con_der'):
                term_con_der = []
            self.term_con_len.append(len(term_con))
            for k, (spline, condition) in enumerate(term_con):
                g = self.define_spline_variable('g' + str(k), 1, basis=spline.basis)[0]
                objective = objective + definite_integral(g, self.t0, 1.)
                self.define_constraint(spline - condition - g, -inf, 0.)
                self.define_constraint(-spline + condition - g, -inf, 0.)
                if self.options['hard_term_con']:
                    self.define_constraint(spline(1.) - condition, 0., 0.)
            for spline, condition in term_con_der:
                self.define_constraint(spline(1.) - condition, 0., 0.)
        self.define_objective(objective)

    def set_parameters(self, current_time):
        parameters = {self: {}}
        if self.init_time is None:
            parameters[self]['t'] = np.round(current_time, 6) % self.knot_time
        else:
            parameters[self]['t'] = self.init_time
        parameters[self]['T'] = self.options['horizon_time']
        return parameters

    # -- deployment --------------------------------------------------------------------
    def init_step(self, current_time, update_time):
        if not hasattr(self, 'current_time_prev'):
            self.current_time_prev = 0
        interval_prev = int(np.round(self.current_time_prev / self.knot_time, 6))
        interval_now = int(np.round(current_time / self.knot_time, 6))
        if interval_prev < interval_now:      # a knot was passed: shift the warm start
            self.father.transform_primal_splines(lambda coeffs, basis, T: T.dot(coeffs))
        self.current_time_prev = current_time

    def init_primal_transform(self, basis):
        return shiftoverknot_T(basis)

    def init_dual_transform(self, basis):
        return None

    def initialize(self, current_time):
        Point2pointProblem.initialize(self, current_time)
        self.current_time_prev = current_time

    def _rel_time(self, current_time):
        if self.init_time is None:
            return np.round(current_time - self.start_time, 6) % self.knot_time
        return self.init_time

    def store(self, current_time, update_time, sample_time):
        horizon_time = self.options['horizon_time']
        rel = self._rel_time(current_time)
        for vehicle in self.vehicles:
            n_samp = int(round((horizon_time - rel) / sample_time, 6)) + 1
            time_axis = np.linspace(rel, rel + (n_samp - 1) * sample_time, n_samp)
            segments = [self.father.get_variables(vehicle, 'splines_seg' + str(k))
                        for k in range(vehicle.n_seg)]
            vehicle.store(current_time, sample_time, segments, horizon_time, time_axis)

    def simulate(self, current_time, simulation_time, sample_time):
        horizon_time = self.options['horizon_time']
        rel = self._rel_time(current_time)
        if horizon_time - rel < simulation_time:
            simulation_time = horizon_time - rel
        self.compute_partial_objective(current_time, simulation_time)
        Problem.simulate(self, current_time, simulation_time, sample_time)

    def compute_partial_objective(self, current_time, update_time):
        horizon_time = self.options['horizon_time']
        t0 = (np.round(current_time - self.start_time, 6) % self.knot_time) / horizon_time
        t1 = t0 + update_time / horizon_time
        part = 0.
        for v in range(len(self.vehicles)):
            for k in range(self.term_con_len[v]):
                g = self.father.get_variables(self, 'g' + str(k))[0]
                part += horizon_time * definite_integral(g, t0, t1)
        self.objective += part

    def compute_objective(self):
        if self.objective == 0:
            obj = 0.
            for v in range(len(self.vehicles)):
                for k in range(self.term_con_len[v]):
                    g = self.father.get_variables(self, 'g' + str(k))[0]
                    obj += self.options['horizon_time'] * g.integral()
            return obj
        return self.objective


class FreeEndPoint2point(FixedTPoint2point):
    """Fixed horizon, free end point (`problems/point2point.py:376-418`): the terminal conditions
    listed in `free_ind[vehicle]` become the variable `conT<l>`; the building block of the
    reference's RendezVous problem (`problems/rendezvous.py:29-35`)."""

    def __init__(self, fleet, environment, options, free_ind=None):
        FixedTPoint2point.__init__(self, fleet, environment, options)
        self.free_ind = free_ind

    def construct(self):
        if self.free_ind is None:
            self.free_ind = {}
            for vehicle in self.vehicles:
                term_con = vehicle.get_terminal_constraints(vehicle.splines[0])
                self.free_ind[vehicle] = list(range(len(term_con)))
        FixedTPoint2point.construct(self)

    def define_terminal_constraints(self):
        objective = 0.
        self.term_con_len = []
        for l, vehicle in enumerate(self.vehicles):
            term_con, term_con_der = vehicle.get_terminal_constraints(vehicle.splines[0])
            conditions = np.atleast_1d(self.define_variable('conT' + str(l), len(self.free_ind[vehicle])))
            cnt = 0
            self.term_con_len.append(len(term_con))
            for k, con in enumerate(term_con):
                if k in self.free_ind[vehicle]:
                    spline, condition = con[0], conditions[cnt]
                    cnt += 1
                else:
                    spline, condition = con[0], con[1]
                g = self.define_spline_variable('g' + str(k), 1, basis=spline.basis)[0]
                objective = objective + definite_integral(g, self.t0, 1.)
                self.define_constraint(spline - condition - g, -inf, 0.)
                self.define_constraint(-spline + condition - g, -inf, 0.)
            # as in the reference (`point2point.py:416-417`): the loop over the derivative conditions
            # re-uses `spline, condition` of the last position condition, i.e. it pins the end point of
            # the LAST position spline len(term_con_der) times -- kept, the NLP has to be the same
            for con in term_con_der:
                self.define_constraint(spline(1.) - condition, 0., 0.)
        self.define_objective(objective)


class FreeTPoint2point(Point2pointProblem):
    """Free end time (`point2point.py:269-369`): the motion time T is a variable and the
    objective; hard terminal constraints; after every update the spline is re-expressed on the
    remaining horizon (`shift_spline`) and T reduced by the update time.

    The reference resolves symbols by name, so that the parameter `T` the base class defines first
    ends up unused (every `T` becomes the variable) but stays in the parameter vector; that
    layout is kept.  Its `t` is always 0 for this problem class (the time axis resets every
    update, `point2point.py:299-306`), so t0 = t/T is the constant 0 here; `set_init_time` is not
    supported for free-T problems."""

    def __init__(self, fleet, environment, options):
        Point2pointProblem.__init__(self, fleet, environment, options)
        self.objective = 0.

    def define_time(self):
        self.define_parameter('T')                       # kept for the reference's parameter layout
        self.t = self.define_parameter('t')
        self.T = self.define_variable('T', value=10.)
        self.t0 = 0.

    def construct(self):
        Point2pointProblem.construct(self)
        self.define_objective(self.T)
        self.define_constraint(-self.T, -inf, 0.)          # positive motion time
        self.define_init_constraints()
        self.define_terminal_constraints()

    def define_terminal_constraints(self):
        for vehicle in self.vehicles:
            term_con, term_con_der = vehicle.get_terminal_constraints(vehicle.splines[0])
            if self.options.get('no_term_con_der'):
                term_con_der = []
            for spline, condition in term_con + term_con_der:
                self.define_constraint(spline(1.) - condition, 0., 0.)

    def set_parameters(self, current_time):
        if self.init_time is not None:
            raise NotImplementedError('set_init_time is not supported for free-T problems')
        return {self: {'t': 0.}}

    # -- deployment --------------------------------------------------------------------
    def horizon(self):
        return float(np.asarray(self.father.get_variables(self, 'T')).reshape(-1)[0])

    def init_step(self, current_time, update_time):
        if (current_time - self.start_time) > 0:
            T = self.horizon()
            if T < 2 * update_time:            # almost arrived: lower the update time
                update_time = T - update_time
                target_time = T
            else:
                target_time = T - update_time
            from .splines import shift_spline_T
            cache = {}

            def shift(coeffs, basis, _T=None):
                key = id(basis)
                if key not in cache:
                    cache[key] = shift_spline_T(basis, update_time / target_time)
                return cache[key].dot(coeffs)
            self.father.transform_primal_splines(shift)
            self.father.set_variables(target_time, self, 'T')

    def store(self, current_time, update_time, sample_time):
        horizon_time = self.horizon()
        if horizon_time < sample_time:
            return
        for vehicle in self.vehicles:
            n_samp = int(round(horizon_time / sample_time, 6)) + 1
            time_axis = np.linspace(0., (n_samp - 1) * sample_time, n_samp)
            segments = [self.father.get_variables(vehicle, 'splines_seg' + str(k))
                        for k in range(vehicle.n_seg)]
            vehicle.store(current_time, sample_time, segments, horizon_time, time_axis)

    def simulate(self, current_time, simulation_time, sample_time):
        horizon_time = self.horizon()
        if horizon_time < sample_time:
            return
        simulation_time = min(simulation_time, horizon_time)
        self.objective = current_time + simulation_time - self.start_time
        Problem.simulate(self, current_time, simulation_time, sample_time)

    def stop_criterium(self, current_time, update_time):
        if self.horizon() < update_time:
            return True
        return Point2pointProblem.stop_criterium(self, current_time, update_time)

    def compute_objective(self):
        return self.objective
