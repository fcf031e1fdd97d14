# This file is derived from OMG-tools (meco-group/omg-tools, `omgtools/problems/problem.py`, `omgtools/problems/point2point.py` (API surface, option keys, definition order)).
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
# Modifications: written anew for this repository on the same public classes, option names, definition order and messages
# (they fix the flat x / p / g layouts of the drop-in boundary), on explicit polynomials (symbolic.py) instead of CasADi and
# with the solver call replaced by the HIP path (backend.py).  Distributed under the same licence (COPYING.LESSER beside this file).
"""`Problem` and the point-to-point family (`Point2point` factory, `FixedTPoint2point`, `FreeEndPoint2point`,
`FreeTPoint2point`): what one receding-horizon solve IS -- written for this package against the behaviour of the reference's
`problems/problem.py` (options 54-74, init 85-91, solve 103-136, predict 138-163, reset_init_guess 165-181, simulate 187-192)
and `problems/point2point.py` (factory 28-35, construct 53-62, initial rows 64-70, fixed T 126-266, free T 269-369, free end
376-418): same class names, constructor arguments, option keys, variable / parameter names and ORDER of definitions (the flat
layouts are pinned by tests/golden/nlp_*.npz).

`Problem.solve` is the call this whole repository exists for: it keeps the reference's shape
`result = self.problem(x0=, p=, lbg=, ubg=)` / `self.problem.stats()['return_status']` (`problems/problem.py:113-119`), and
`self.problem` is the HIP solver object of backend.py where the reference holds a CasADi `nlpsol`.

The formulation, in this package's words.  Time runs in spline time tau in [0, 1] over a horizon of T seconds; the plan began
t seconds ago (t0 = t / T), so the vehicle is at tau = t0 NOW: initial conditions are rows at t0, the objective integrates from
t0, and when t passes a knot the plan is re-expressed on a horizon that starts one knot interval later (`init_step`).  Reaching
the target is soft: |spline - target| <= g with the slack spline g integrated in the objective; the terminal derivatives are
pinned to zero.  With a free end time T is itself the variable and the objective."""
import time

import numpy as np

from .opti import OptiChild, OptiFather, inf
from .plotting import PlotLayer
from .vehicles import get_fleet_vehicles
from .splines import definite_integral, evalspline, shiftoverknot_T, since_knot

_NESTED_OPTIONS = ('solver_options', 'codegen')


class Problem(OptiChild, PlotLayer):

    def __init__(self, fleet, environment, options=None, label='problem'):
        OptiChild.__init__(self, label)
        PlotLayer.__init__(self)
        self.environment = environment
        self.fleet, self.vehicles = get_fleet_vehicles(fleet)
        self.iteration, self.update_times = 0, []
        self.set_default_options()
        self.set_options(options or {})
        # the order of the children is the order of their blocks in x, p and g (`problem.py:45-48`)
        self.father = OptiFather(list(self.vehicles) + list(environment.obstacles) + [self, environment])

    def set_default_options(self):
        # (the reference's option names: the HIP solver reads 'ipopt.tol' / 'ipopt.max_iter' and ignores the rest)
        ipopt = {'ipopt.tol': 1e-3, 'ipopt.warm_start_init_point': 'yes', 'ipopt.fixed_variable_treatment': 'make_constraint'}
        ipopt.update({'ipopt.print_level': 0, 'print_time': 0})
        self.options = {'verbose': 2, 'solver': 'ipopt', 'solver_options': {'ipopt': ipopt}, 'codegen': {'build': None, 'flags': '-O0'}}

    def set_options(self, options):
        for key, value in options.items():
            if key == 'solver_options':
                for solver, settings in value.items():
                    self.options['solver_options'].setdefault(solver, {}).update(settings)
            elif key == 'codegen':
                self.options['codegen'].update(value)
            else:
                self.options[key] = value

    # ---- building the NLP ----------------------------------------------------------------------------------------------
    def construct(self):
        self.environment.init()
        for veh in self.vehicles:
            veh.init()

    def init(self):
        """Define everything (`construct`), hand the template to the solver factory; returns the build time."""
        self.father.reset()
        with self.father.table:
            self.construct()
            self.problem, seconds = self.father.construct_problem(self.options)
        self.father.init_transformations(self.init_primal_transform, self.init_dual_transform)
        return seconds

    def init_primal_transform(self, basis):
        return None

    def init_dual_transform(self, basis):
        return None

    def init_step(self, current_time, update_time):
        pass

    def initialize(self, current_time):
        pass

    def reinitialize(self, father=None):
        target = father if father is not None else self.father
        target.init_variables()
        target.init_parameters()

    # ---- the hot call ------------------------------------------------------------------------------------------------------
    def solve(self, current_time, update_time):
        now = current_time - self.start_time
        self.init_step(now, update_time)
        fa = self.father
        x0, p = fa.get_variables(), fa.set_parameters(now)
        lbg, ubg = fa.update_bounds(now)
        tic = time.time()
        result = self.problem(x0=x0, p=p, lbg=lbg, ubg=ubg)
        spent = time.time() - tic
        fa.set_variables(result['x'])
        fa.set_dual_variables(result['lam_g'])
        outcome = self.problem.stats()['return_status']
        if outcome == 'Maximum_CpuTime_Exceeded':
            if now != 0.0:
                print('Maximum solving time exceeded, resetting initial guess')
                self.reset_init_guess()
                print(outcome)
        elif outcome != 'Solve_Succeeded':
            print(outcome)
        if self.options['verbose'] >= 2:
            if self.iteration % 20 == 0:
                rule = '----|------------|------------'
                print('\n'.join((rule, '%3s | %10s | %10s ' % ('It', 't upd', 'time'), rule)))
            self.iteration += 1
            print('%3d | %.4e | %.4e ' % (self.iteration, spent, now))
        self.update_times.append(spent)

    def predict(self, current_time, predict_time, sample_time, states=None, inputs=None, dinputs=None, delay=0,
                enforce_states=False, enforce_inputs=False):
        n = len(self.vehicles)

        def spread(values):
            """None, one vehicle's value, or a list over the vehicles -> a list over the vehicles"""
            if values is None:
                return [None] * n
            single = n == 1 and (not isinstance(values, list) or isinstance(values[0], float))
            return [values] if single else values
        per = [spread(v) for v in (states, inputs, dinputs)]
        first = current_time == self.start_time                 # the first update starts from the state the user set
        for k, veh in enumerate(self.vehicles):
            veh.predict(current_time, predict_time, sample_time, per[0][k], per[1][k], per[2][k], delay,
                        enforce_states or first, enforce_inputs)

    def reset_init_guess(self, init_guess=None):
        if init_guess is None:
            guesses = [veh.get_init_spline_value() for veh in self.vehicles]
        else:
            guesses = init_guess if isinstance(init_guess, list) else [init_guess]
        for veh, guess in zip(self.vehicles, guesses):
            segments = guess if isinstance(guess, list) else [guess]
            if len(segments) != veh.n_seg:
                raise ValueError('Each spline segment of the vehicle should receive an initial guess.')
            for l, coeffs in enumerate(segments):
                if coeffs.shape[1] != veh.n_spl:
                    raise ValueError('Each vehicle spline should receive an initial guess.')
                self.father.set_variables(coeffs, child=veh, name='splines_seg%d' % l)

    def simulate(self, current_time, simulation_time, sample_time):
        for veh in self.vehicles:
            veh.simulate(simulation_time, sample_time)
        self.environment.simulate(simulation_time, sample_time)
        self.fleet.update_plots()
        self.update_plots()

    def stop_criterium(self, current_time, update_time):
        return False

    def final(self):
        pass

    def store(self, current_time, update_time, sample_time):
        pass

    # ---- helpers of the subclasses ---------------------------------------------------------------------------------------
    def _plan_of(self, veh):
        return [self.father.get_variables(veh, 'splines_seg%d' % k) for k in range(veh.n_seg)]

    def _hand_plan_over(self, veh, current_time, sample_time, first, horizon):
        """`veh.store` of the plan over `horizon` seconds, sampled on first, first + sample_time, ... up to its end."""
        n = int(round((horizon - first) / sample_time, 6)) + 1
        grid = np.linspace(first, first + (n - 1) * sample_time, n)
        veh.store(current_time, sample_time, self._plan_of(veh), horizon, grid)


class Point2point(object):
    """`Point2point(fleet, environment, options, freeT=False)`: the fixed- or the free-end-time class (`point2point.py:28-35`)."""

    def __new__(cls, fleet, environment, options=None, freeT=False):
        return (FreeTPoint2point if freeT else FixedTPoint2point)(fleet, environment, options)


class Point2pointProblem(Problem):

    def __init__(self, fleet, environment, options):
        Problem.__init__(self, fleet, environment, options, label='p2p')
        self.init_time, self.start_time = None, 0.

    def set_default_options(self):
        Problem.set_default_options(self)
        self.options['inter_vehicle_avoidance'] = False

    def define_time(self):
        """Parameters T (horizon) and t (time since the plan began), t0 = t / T (`point2point.py:53-55`)."""
        self.T = self.define_parameter('T')
        self.t = self.define_parameter('t')
        self.t0 = self.t / self.T

    def construct(self):
        self.define_time()
        Problem.construct(self)
        for veh in self.vehicles:
            segments = veh.define_splines(n_seg=1)
            veh.define_trajectory_constraints(segments[0], self.T)
            self.environment.define_collision_constraints(veh, segments, self.T)
        if self.options['inter_vehicle_avoidance'] and len(self.vehicles) > 1:
            self.environment.define_intervehicle_collision_constraints(self.vehicles, self.T)

    def define_init_constraints(self):
        """The plan passes through the predicted state NOW: rows at tau = t0."""
        for veh in self.vehicles:
            for spline, value in veh.get_initial_constraints(veh.splines[0], self.T):
                self.define_constraint(evalspline(spline, self.t0) - value, 0., 0.)

    def initialize(self, current_time):
        self.start_time = current_time

    def reinitialize(self, father=None):
        target = father if father is not None else self.father
        Problem.reinitialize(self)
        for veh in self.vehicles:
            for k, coeffs in enumerate(veh.get_init_spline_value()[:veh.n_seg]):
                target.set_variables(coeffs, veh, 'splines_seg%d' % k)

    def set_init_time(self, time):
        self.init_time = time

    def reset_init_time(self):
        self.init_time = None

    def stop_criterium(self, current_time, update_time):
        return all(veh.check_terminal_conditions() for veh in self.vehicles)

    def final(self):
        self.reset_init_time()
        total = self.compute_objective()
        if self.options['verbose'] >= 1:
            ms = 1000. * np.asarray(self.update_times)
            print('\nWe reached our target!')
            for name, value, unit in (('Objective:', total, ''), ('Max update time:', ms.max(), ' ms'), ('Av update time:', ms.sum() / len(ms), ' ms')):
                print('%-18s %6g%s' % (name, value, unit))

    def compute_objective(self):
        raise NotImplementedError('Please implement this method!')

    def _reach_softly(self, k, spline, target):
        """Slack spline g<k> with |spline - target| <= g; returns its share of the objective (integral of g from t0)."""
        g = self.define_spline_variable('g%d' % k, 1, basis=spline.basis)[0]
        self.define_constraint(spline - target - g, -inf, 0.)
        self.define_constraint(-spline + target - g, -inf, 0.)
        return definite_integral(g, self.t0, 1.)


class FixedTPoint2point(Point2pointProblem):

    def __init__(self, fleet, environment, options):
        Point2pointProblem.__init__(self, fleet, environment, options)
        self.objective = 0.
        intervals = self.vehicles[0].knot_intervals
        if intervals is None:
            raise ValueError('A constant knot interval should be used for a fixed T point2point problem.')
        # (seconds per knot interval, in whole milliseconds of the horizon like the reference)
        self.knot_time = (int(self.options['horizon_time'] * 1000.) / intervals) / 1000.

    def set_default_options(self):
        Point2pointProblem.set_default_options(self)
        self.options.update(horizon_time=10., hard_term_con=False, no_term_con_der=False)

    def construct(self):
        Point2pointProblem.construct(self)
        self.define_init_constraints()
        self.define_terminal_constraints()

    def define_terminal_constraints(self):
        cost = 0.
        self.term_con_len = []
        for veh in self.vehicles:
            reach, rest = veh.get_terminal_constraints(veh.splines[0])
            self.term_con_len.append(len(reach))
            for k, (spline, target) in enumerate(reach):
                cost = cost + self._reach_softly(k, spline, target)
                if self.options['hard_term_con']:
                    self.define_constraint(spline(1.) - target, 0., 0.)
            for spline, target in ([] if self.options.get('no_term_con_der') else rest):
                self.define_constraint(spline(1.) - target, 0., 0.)
        self.define_objective(cost)

    def _since_knot(self, elapsed):
        """t: seconds since the last knot the plan passed (or the time the user pinned with `set_init_time`)."""
        return since_knot(elapsed, self.knot_time) if self.init_time is None else self.init_time

    def set_parameters(self, current_time):
        return {self: {'t': self._since_knot(current_time), 'T': self.options['horizon_time']}}

    # ---- between two solves ----------------------------------------------------------------------------------------------
    def init_primal_transform(self, basis):
        return shiftoverknot_T(basis)

    def initialize(self, current_time):
        Point2pointProblem.initialize(self, current_time)
        self.current_time_prev = current_time

    def init_step(self, current_time, update_time):
        """A knot was passed since the previous solve: move the warm start to the horizon that begins one interval later."""
        before = getattr(self, 'current_time_prev', 0)
        passed = lambda when: int(np.round(when / self.knot_time, 6))
        if passed(before) < passed(current_time):
            self.father.transform_primal_splines(lambda coeffs, basis, T: T.dot(coeffs))
        self.current_time_prev = current_time

    def store(self, current_time, update_time, sample_time):
        begun = self._since_knot(current_time - self.start_time)
        for veh in self.vehicles:
            self._hand_plan_over(veh, current_time, sample_time, begun, self.options['horizon_time'])

    def simulate(self, current_time, simulation_time, sample_time):
        left = self.options['horizon_time'] - self._since_knot(current_time - self.start_time)
        span = min(simulation_time, left)
        self.compute_partial_objective(current_time, span)
        Problem.simulate(self, current_time, span, sample_time)

    def _slacks(self):
        for count in self.term_con_len:
            for k in range(count):
                yield self.father.get_variables(self, 'g%d' % k)[0]

    def compute_partial_objective(self, current_time, update_time):
        """What the part of the plan that is executed now costs (summed up over the run as `self.objective`)."""
        T = self.options['horizon_time']
        a = since_knot(current_time - self.start_time, self.knot_time) / T
        b = a + update_time / T
        self.objective += sum(T * definite_integral(g, a, b) for g in self._slacks())

    def compute_objective(self):
        if self.objective != 0:
            return self.objective
        return sum(self.options['horizon_time'] * g.integral() for g in self._slacks()) if self.term_con_len else 0.


class FreeEndPoint2point(FixedTPoint2point):
    """Fixed horizon, free end point (`problems/point2point.py:376-418`): the terminal conditions listed in `free_ind[vehicle]`
    become the variable `conT<l>` -- the building block of the rendez-vous problem (`problems/rendezvous.py:29-35`)."""

    def __init__(self, fleet, environment, options, free_ind=None):
        FixedTPoint2point.__init__(self, fleet, environment, options)
        self.free_ind = free_ind

    def construct(self):
        if self.free_ind is None:               # every terminal condition of every vehicle is free
            self.free_ind = dict((veh, list(range(len(veh.get_terminal_constraints(veh.splines[0]))))) for veh in self.vehicles)
        FixedTPoint2point.construct(self)

    def define_terminal_constraints(self):
        cost = 0.
        self.term_con_len = []
        for l, veh in enumerate(self.vehicles):
            reach, rest = veh.get_terminal_constraints(veh.splines[0])
            free = list(self.free_ind[veh])
            ends = np.atleast_1d(self.define_variable('conT%d' % l, len(free)))
            self.term_con_len.append(len(reach))
            taken = 0
            for k, (spline, fixed) in enumerate(reach):
                if k in free:
                    target, taken = ends[taken], taken + 1
                else:
                    target = fixed
                cost = cost + self._reach_softly(k, spline, target)
            # The reference's loop over the derivative conditions (`point2point.py:416-417`) re-uses the spline and the target
            # of the LAST position condition: it pins the end point of that spline len(rest) times.  Kept: the NLP has to be the same.
            for _ in rest:
                self.define_constraint(spline(1.) - target, 0., 0.)
        self.define_objective(cost)


class FreeTPoint2point(Point2pointProblem):
    """Free end time (`point2point.py:269-369`): the motion time T is a variable and the objective, the terminal conditions
    are hard; after every update the plan is re-expressed on what is left of it and T reduced by the update time.

    The reference resolves symbols by name: the parameter `T` its base class defines first ends up unused (every `T` is the
    variable) but keeps its place in the parameter vector -- so it does here.  Its `t` is 0 throughout (the time axis starts
    anew with every update, `point2point.py:299-306`), so t0 = 0; `set_init_time` has no meaning for this class."""

    def __init__(self, fleet, environment, options):
        Point2pointProblem.__init__(self, fleet, environment, options)
        self.objective = 0.

    def define_time(self):
        self.define_parameter('T')                       # (the unused slot of the reference's parameter layout)
        self.t = self.define_parameter('t')
        self.T = self.define_variable('T', value=10.)
        self.t0 = 0.

    def construct(self):
        Point2pointProblem.construct(self)
        self.define_objective(self.T)
        self.define_constraint(-self.T, -inf, 0.)          # a motion time is positive
        self.define_init_constraints()
        self.define_terminal_constraints()

    def define_terminal_constraints(self):
        for veh in self.vehicles:
            reach, rest = veh.get_terminal_constraints(veh.splines[0])
            for spline, target in reach + ([] if self.options.get('no_term_con_der') else rest):
                self.define_constraint(spline(1.) - target, 0., 0.)

    def set_parameters(self, current_time):
        if self.init_time is not None:
            raise NotImplementedError('set_init_time is not supported for free-T problems')
        return {self: {'t': 0.}}

    # ---- between two solves ----------------------------------------------------------------------------------------------
    def horizon(self):
        return float(np.asarray(self.father.get_variables(self, 'T')).reshape(-1)[0])

    def init_step(self, current_time, update_time):
        if not (current_time - self.start_time) > 0:
            return
        from .splines import shift_spline_T
        T = self.horizon()
        if T < 2 * update_time:                # almost there: a shorter update, the horizon stays
            update_time, remaining = T - update_time, T
        else:
            remaining = T - update_time
        moved = {}

        def onto_the_rest(coeffs, basis, _T=None):
            if id(basis) not in moved:
                moved[id(basis)] = shift_spline_T(basis, update_time / remaining)
            return moved[id(basis)].dot(coeffs)
        self.father.transform_primal_splines(onto_the_rest)
        self.father.set_variables(remaining, self, 'T')

    def store(self, current_time, update_time, sample_time):
        T = self.horizon()
        if T >= sample_time:
            for veh in self.vehicles:
                self._hand_plan_over(veh, current_time, sample_time, 0., T)

    def simulate(self, current_time, simulation_time, sample_time):
        T = self.horizon()
        if T >= sample_time:
            span = min(simulation_time, T)
            self.objective = current_time + span - self.start_time
            Problem.simulate(self, current_time, span, sample_time)

    def stop_criterium(self, current_time, update_time):
        return self.horizon() < update_time or Point2pointProblem.stop_criterium(self, current_time, update_time)

    def compute_objective(self):
        return self.objective
