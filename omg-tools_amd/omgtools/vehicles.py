# This file is derived from OMG-tools (meco-group/omg-tools, `omgtools/vehicles/vehicle.py`, `holonomic.py`, `holonomic3d.py`, `quadrotor.py`, `fleet.py` (API surface, option keys, definition order)).
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
"""Vehicle models `Vehicle`, `Holonomic`, `Holonomic3D`, `Quadrotor` and the `Fleet` container -- written for this package
against the behaviour of the reference's `vehicles/vehicle.py` (define_knots 80-87, define_splines 105-120, collision rows 2-D
122-190 / 3-D 192-232, store 250-300, predict 302-337, simulate 359-401), `vehicles/holonomic.py` (62-175),
`vehicles/holonomic3d.py` (44-152), `vehicles/quadrotor.py` (48-163) and `vehicles/fleet.py` (25-98): same class names,
constructor arguments, option keys, parameter / variable names and the same ORDER of constraint definitions (it fixes the row
layout of g, pinned by tests/golden/nlp_*.npz), on this package's polynomial layer (symbolic.py) instead of CasADi.

How the models are organised here: a vehicle is a set of B-spline outputs; a `_PointMass` is the family whose state is the
position and whose input is the velocity (the two holonomic classes differ in their dimension and in how their limits are
named, nothing else); the quadrotor's outputs are flat outputs (x, z) from which thrust and pitch rate follow.  The part of
`Vehicle` below the modelling section is the host-side bookkeeping of a receding-horizon run (sampled trajectories, prediction
of the next initial state, simulation of the plant).
"""
import numpy as np

from .opti import OptiChild, inf
from .plotting import PlotLayer
from .shapes import Circle, Rectangle, Square
from .splines import BSplineBasis, BSpline, definite_integral, concat_splines, sample_splines

_PLANT_KEYS = ('state', 'input', 'pose')      # signals a simulated plant produces itself (everything else is read off the plan)


def _as_list(thing):
    return thing if isinstance(thing, list) else [thing]


def _whole_samples(span, sample_time):
    return int(np.round(span / sample_time, 6))


class Vehicle(OptiChild, PlotLayer):

    def __init__(self, n_spl, degree, shapes, options=None):
        OptiChild.__init__(self, 'vehicle')
        PlotLayer.__init__(self)
        self.shapes = _as_list(shapes)
        dims = set(shape.n_dim for shape in self.shapes)
        if len(dims) != 1:
            raise ValueError('All vehicle shapes should have same spatial dimension.')
        self.n_dim = dims.pop()
        self.n_spl, self.degree = n_spl, degree
        self.prediction, self.init_spline_values, self.to_simulate = {}, None, True
        self._ic_version = 0                       # bumped whenever a state / input is handed in from outside (formation.py)
        self.set_default_options()
        self.set_options(options or {})
        self.define_knots(knot_intervals=10)

    def set_default_options(self):
        self.options = dict(safety_distance=0., safety_weight=10., room_constraints=True, stop_tol=1.e-3,
                            ideal_prediction=False, ideal_update=False, time_constant=0.1, input_disturbance=None)
        self.options['1storder_delay'] = False

    def set_options(self, options):
        self.options.update(options)

    def define_knots(self, **kwargs):
        """knot_intervals=K: clamped uniform knots on [0, 1]; knots=...: the caller's vector."""
        if 'knot_intervals' in kwargs:
            K = self.knot_intervals = kwargs['knot_intervals']
            self.knots = np.concatenate((np.zeros(self.degree), np.linspace(0., 1., K + 1), np.ones(self.degree)))
        if 'knots' in kwargs:
            self.knots = kwargs['knots']
        self.basis = BSplineBasis(self.knots, self.degree)

    def set_init_spline_values(self, values, n_seg=1):
        want = (len(self.basis), self.n_spl)
        for k in range(n_seg):
            if values[k].shape != want:
                raise ValueError('Initial guess has wrong dimensions for spline %d, required: %s while you gave: %s'
                                 % (k, want, values[k].shape))
        self.init_spline_values = [values[k] for k in range(n_seg)]

    # ---- modelling ---------------------------------------------------------------------------------------------------
    def init(self):
        pass

    def define_splines(self, n_seg=1):
        """One spline variable `splines_seg<k>` (len(basis) x n_spl) per segment.  Its construction-time value is zero unless
        `set_init_spline_values` was called (`vehicle.py:112-115`); the guess proper comes with `Problem.reinitialize`."""
        start, self.init_spline_values = self.init_spline_values or [None] * n_seg, None
        self.n_seg = n_seg
        self.splines = [self.define_spline_variable('splines_seg%d' % k, self.n_spl, value=start[k]) for k in range(n_seg)]
        return self.splines

    def _margin(self, s, k, t, horizon_time):
        """The soft safety margin of hyperplane k of shape s: a spline eps in [0, safety_distance] that the objective
        pulls down (weight x its integral over the remaining horizon); 0 when no safety distance is asked for."""
        reach = self.options['safety_distance']
        if not reach > 0.:
            return 0.
        eps = self.define_spline_variable('eps_%d%d' % (s, k))[0]
        self.define_objective(self.options['safety_weight'] * definite_integral(eps, t / horizon_time, 1.))
        self.define_constraint(eps - reach, -inf, 0.)
        self.define_constraint(-eps, -inf, 0.)
        return eps

    @staticmethod
    def _side_2d(normal, offset_b, point, centre, clearance, tg_ha, offset):
        """a . (R(theta) point + centre + offset e(theta)) - b + clearance <= 0 multiplied by (1 + tan^2(theta / 2)) > 0, with
        cos / sin of the heading in tangent-half-angle form (tg_ha = 0: a vehicle that does not turn)."""
        one_m, one_p, two = 1. - tg_ha**2, 1 + tg_ha**2, 2 * tg_ha
        row = (normal[0] * point[0] + normal[1] * point[1]) * one_m
        row += (-normal[0] * point[1] + normal[1] * point[0]) * two
        cx = centre[0] * one_p + offset * (1 - tg_ha**2)
        cy = centre[1] * one_p + offset * two
        row += normal[0] * cx + normal[1] * cy
        row += (-offset_b + clearance) * one_p
        return row

    def define_collision_constraints_2d(self, hyperplanes, room, positions, horizon_time, tg_ha=0, offset=0):
        t = self.define_symbol('t')
        reach = self.options['safety_distance']
        per_shape = positions if isinstance(positions[0], list) else [positions]
        straight = isinstance(tg_ha, (int, float)) and tg_ha == 0.
        for s, shape in enumerate(self.shapes):
            centre = per_shape[s]
            points, radii = shape.get_checkpoints()
            for k, plane in enumerate(hyperplanes.get(shape, [])):
                eps = self._margin(s, k, t, horizon_time)
                grow = plane.get('slack', 1)
                for l, point in enumerate(points):
                    self.define_constraint(self._side_2d(plane['a'], plane['b'], point, centre, grow * radii[l] + reach - eps,
                                                         tg_ha, offset), -inf, 0)
            if not self.options['room_constraints']:
                continue
            extent = room['shape'].get_canvas_limits()
            box = [extent[k] + room['position'][k] for k in range(self.n_dim)]
            upright = lambda sh: isinstance(sh, (Rectangle, Square)) and sh.orientation == 0
            if upright(room['shape']) and (isinstance(shape, Circle) or upright(shape)) and straight:
                # an axis-aligned room around an axis-aligned vehicle: plain bounds on the coordinates
                for point in points:
                    for k in range(self.n_dim):
                        self.define_constraint(-(point[k] + centre[k]) + box[k][0] + radii[0], -inf, 0.)
                        self.define_constraint((point[k] + centre[k]) - box[k][1] + radii[0], -inf, 0.)
            else:
                walls = room['shape'].get_hyperplanes(position=room['position'])
                for l, point in enumerate(points):
                    for wall in walls.values():
                        self.define_constraint(self._side_2d(wall['a'], wall['b'], point, centre, radii[l], tg_ha, offset), -inf, 0)

    def define_collision_constraints_3d(self, hyperplanes, room, positions, horizon_time):
        t = self.define_symbol('t')
        reach = self.options['safety_distance']
        per_shape = positions if isinstance(positions[0], list) else [positions]
        for s, shape in enumerate(self.shapes):
            centre = per_shape[s]
            points, radii = shape.get_checkpoints()
            for k, plane in enumerate(hyperplanes.get(shape, [])):
                eps = self._margin(s, k, t, horizon_time)
                a = plane['a']
                for l, point in enumerate(points):
                    along = a[0] * (point[0] + centre[0])
                    for q in (1, 2):
                        along = along + a[q] * (point[q] + centre[q])
                    self.define_constraint(along - plane['b'] + radii[l] + reach - eps, -inf, 0)
            if self.options['room_constraints']:
                extent = room['shape'].get_canvas_limits()
                box = [extent[k] + room['position'][k] for k in range(self.n_dim)]
                for point in points:
                    for k in range(3):
                        self.define_constraint(-(point[k] + centre[k]) + box[k][0], -inf, 0.)
                        self.define_constraint((point[k] + centre[k]) - box[k][1], -inf, 0.)

    def get_fleet_center(self, splines, rel_pos, substitute=True):
        centre = [spline + shift for spline, shift in zip(splines, list(rel_pos))]
        return self.define_substitute('fleet_center', centre) if substitute else centre

    def set_parameters(self, current_time):
        return {self: {}}

    # ---- the plan as sampled trajectories ------------------------------------------------------------------------------
    def _sampled(self, splines, grid, current_time):
        """What `splines2signals` gives on `grid` plus time, pose and the raw splines (and the fleet centre of a formation)."""
        out = self.splines2signals(splines, grid)
        if not set(['state', 'input']).issubset(out):
            raise ValueError('Signals should contain at least state, input and pose.')
        out['time'] = grid - grid[0] + current_time
        out['pose'] = self._state2pose(out['state'])
        out['splines'] = np.c_[sample_splines(splines, grid)]
        return out

    def store(self, current_time, sample_time, spline_segments, segment_times, time_axis=None, **kwargs):
        spans = _as_list(segment_times)
        extra_knots = kwargs['continuity'] - (self.degree - 1) if 'continuity' in kwargs else None
        self.result_spline_segments = spline_segments
        plan = self.result_splines = concat_splines(spline_segments, spans, n_insert=extra_knots)
        if time_axis is None:
            n = int(round(sum(spans) / sample_time, 6)) + 1
            time_axis = np.linspace(0., (n - 1) * sample_time, n)
        self.trajectories = self._sampled(plan, time_axis, current_time)
        if hasattr(self, 'rel_pos_c') and 'fleet_center' not in self.trajectories:
            shifted = [spline + shift for spline, shift in zip(plan, self.rel_pos_c)]
            self.trajectories['fleet_center'] = np.c_[sample_splines(shifted, time_axis)]
        # the same at the knots that are still ahead
        d, knots = self.degree, plan[0].basis.knots
        first = knots[d] + time_axis[0]
        ahead = np.r_[first, [kn for kn in knots[d + 1:-d] if kn > first]]
        self.trajectories_kn = self._sampled(plan, ahead, current_time)
        for table in (self.trajectories, self.trajectories_kn):
            for key, val in table.items():
                if val.ndim == 1:
                    table[key] = val.reshape(1, -1)

    def predict(self, current_time, predict_time, sample_time, state0=None, input0=None, dinput0=None, delay=0,
                enforce_states=False, enforce_inputs=False):
        """The initial condition of the next solve: handed in (`enforce_*`), read off the plan `predict_time` ahead (ideal), or
        the measured / simulated state integrated over the planned inputs."""
        if enforce_states:
            given = [state0, input0, dinput0] if enforce_inputs else [state0]
            while given and given[-1] is None:
                given.pop()
            if enforce_inputs and len(given) >= 2 and all(g is not None for g in given):
                self.set_initial_conditions(given[0], **dict(zip(('input', 'dinput'), given[1:])))
            elif not enforce_inputs and given:
                self.set_initial_conditions(given[0])
            elif hasattr(self, 'signals'):
                last = [self.signals[key][:, -1] for key in (('state', 'input', 'dinput') if enforce_inputs else ('state',))]
                self.set_initial_conditions(*last)
            return
        at = _whole_samples(predict_time, sample_time) + delay
        plan = self.trajectories
        if self.options['ideal_prediction']:
            self.prediction.update((key, val[:, at]) for key, val in plan.items())
            return
        self.prediction.update((key, val[:, at]) for key, val in plan.items() if key not in _PLANT_KEYS)
        if state0 is None:
            state0 = self.signals['state'][:, -(at - delay) - 1]
        reached = self.integrate_ode(state0, plan['input'][:, delay:], predict_time, sample_time)[:, -1]
        self.prediction['state'], self.prediction['input'] = reached, plan['input'][:, at]
        self.prediction['pose'] = self._state2pose(reached)

    # ---- the plant (host-side simulation) ------------------------------------------------------------------------------
    def overrule_state(self, state):
        state = np.array(state)
        pose = self._state2pose(state)
        self.signals['state'][:, -1], self.signals['pose'][:, -1] = state, pose
        self.prediction['state'], self.prediction['pose'] = state, pose
        self._ic_version += 1

    def overrule_input(self, input, dinput=None):
        input = np.array(input)
        self.signals['input'][:, -1] = self.prediction['input'] = input
        if dinput is not None:
            self.signals['dinput'][:, -1] = self.prediction['dinput'] = dinput
        self._ic_version += 1

    def simulate(self, simulation_time, sample_time):
        plan = self.trajectories
        if self.to_simulate:
            if not hasattr(self, 'signals'):
                self.signals = dict((key, np.c_[val[:, 0]]) for key, val in plan.items())
            n = _whole_samples(simulation_time, sample_time)
            follow = list(plan) if self.options['ideal_update'] else [key for key in plan if key not in _PLANT_KEYS]
            for key in follow:
                self.signals[key] = np.c_[self.signals[key], plan[key][:, 1:n + 1]]
            if not self.options['ideal_update']:
                moved = self.integrate_ode(self.signals['state'][:, -1], plan['input'], simulation_time, sample_time)[:, 1:n + 1]
                self.signals['input'] = np.c_[self.signals['input'], plan['input'][:, 1:n + 1]]
                self.signals['state'] = np.c_[self.signals['state'], moved]
                self.signals['pose'] = np.c_[self.signals['pose'], self._state2pose(moved)]
        if not hasattr(self, 'traj_storage'):
            self.traj_storage, self.traj_storage_kn, self.pred_storage = {}, {}, {}
        copies = int(simulation_time / sample_time)
        for book, page in ((self.traj_storage, plan), (self.traj_storage_kn, self.trajectories_kn), (self.pred_storage, self.prediction)):
            for key, val in page.items():
                book.setdefault(key, []).extend([val] * copies)
        self.update_plots()

    def _state2pose(self, state):
        if state.ndim <= 1:
            return self.state2pose(state)
        return np.array([self.state2pose(state[:, k]) for k in range(state.shape[1])]).T

    def integrate_ode(self, state0, input, integration_time, sample_time, ode=None):
        """Classical Runge-Kutta on the sample grid, inputs interpolated linearly between samples (the reference integrates
        with scipy's odeint, `vehicle.py:412-423`; its C++ export with this scheme, `export/vehicles/Vehicle.cpp:82-110`)."""
        rhs = ode if ode is not None else self.ode
        n = int(integration_time / sample_time) + 1
        last = input.shape[1] - 1
        out = np.zeros((len(state0), n))
        out[:, 0] = state0
        h = sample_time
        for k in range(n - 1):
            u0, u1 = input[:, min(k, last)], input[:, min(k + 1, last)]
            um = 0.5 * u0 + 0.5 * u1
            x = out[:, k]
            s1 = rhs(x, u0)
            s2 = rhs(x + 0.5 * h * s1, um)
            s3 = rhs(x + 0.5 * h * s2, um)
            s4 = rhs(x + h * s3, u1)
            out[:, k + 1] = x + h / 6. * (s1 + 2 * s2 + 2 * s3 + s4)
        return out


class _PointMass(Vehicle):
    """State = position, input = velocity: the spline outputs ARE the position coordinates.  Rows, in the reference's order:
    velocity below / above, acceleration below / above (per axis) or their 2-norms; initial position and velocity;
    terminal position (soft, the problem adds the slack) and vanishing terminal derivatives."""

    def set_default_options(self):
        Vehicle.set_default_options(self)
        self.options.update({'syslimit': 'norm_inf'})

    def _axis_limits(self):
        """[(v_min, v_max, a_min, a_max) per axis]"""
        raise NotImplementedError

    def define_trajectory_constraints(self, splines, horizon_time=None):
        T = self.define_symbol('T') if horizon_time is None else horizon_time
        vel = [s.derivative() for s in splines]
        acc = [s.derivative(2) for s in splines]
        kind = self.options['syslimit']
        if kind == 'norm_2':
            speed2, acc2 = vel[0]**2, acc[0]**2
            for v, a in zip(vel[1:], acc[1:]):
                speed2, acc2 = speed2 + v**2, acc2 + a**2
            self.define_constraint(speed2 - (T**2) * self.vmax**2, -inf, 0.)
            self.define_constraint(acc2 - (T**4) * self.amax**2, -inf, 0.)
        elif kind == 'norm_inf':
            lim = self._axis_limits()
            for k, v in enumerate(vel):
                self.define_constraint(-v + T * lim[k][0], -inf, 0.)
            for k, v in enumerate(vel):
                self.define_constraint(v - T * lim[k][1], -inf, 0.)
            for k, a in enumerate(acc):
                self.define_constraint(-a + (T**2) * lim[k][2], -inf, 0.)
            for k, a in enumerate(acc):
                self.define_constraint(a - (T**2) * lim[k][3], -inf, 0.)
        else:
            raise ValueError('Only norm_2 and norm_inf are defined as system limit.')

    def get_initial_constraints(self, splines, horizon_time=None):
        T = self.define_symbol('T') if horizon_time is None else horizon_time
        n = self.n_spl
        state0, input0 = self.define_parameter('state0', n), self.define_parameter('input0', n)
        return [(splines[k], state0[k]) for k in range(n)] + [(splines[k].derivative(), T * input0[k]) for k in range(n)]

    def get_terminal_constraints(self, splines, horizon_time=None):
        target = self.define_parameter('poseT', self.n_spl)
        at_rest = [(s.derivative(order), 0.) for order in range(1, self.degree + 1) for s in splines]
        return [[(s, target[k]) for k, s in enumerate(splines)], at_rest]

    def set_initial_conditions(self, state, input=None):
        self.prediction['state'] = np.asarray(state, dtype=float)
        self.prediction['input'] = np.zeros(self.n_spl) if input is None else np.asarray(input, float)
        self._ic_version += 1

    def set_terminal_conditions(self, position):
        self.poseT = np.asarray(position, dtype=float)

    def get_init_spline_value(self, subgoals=None):
        """Coefficients on the straight line from the current position to the target."""
        return [np.linspace(self.prediction['state'], self.poseT, len(self.basis))]

    def check_terminal_conditions(self):
        off = np.linalg.norm(self.signals['state'][:, -1] - self.poseT)
        moving = np.linalg.norm(self.signals['input'][:, -1])
        return not (off > self.options['stop_tol'] or moving > self.options['stop_tol'])

    def set_parameters(self, current_time):
        out = Vehicle.set_parameters(self, current_time)
        out[self].update(state0=self.prediction['state'], input0=self.prediction['input'], poseT=self.poseT)
        return out

    def _signals(self, splines, time):
        vel = np.c_[sample_splines([s.derivative() for s in splines], time)]
        return {'state': np.c_[sample_splines(list(splines), time)], 'input': vel, 'v_tot': np.sqrt((vel**2).sum(axis=0))}, \
            np.c_[sample_splines([s.derivative(2) for s in splines], time)]

    def ode(self, state, input):
        return input


class Holonomic(_PointMass):

    def __init__(self, shapes=None, options=None, bounds=None):
        given = bounds or {}
        Vehicle.__init__(self, n_spl=2, degree=3, shapes=Circle(0.1) if shapes is None else shapes, options=options)
        if self.options.get('syslimit', 'norm_inf') == 'norm_inf':
            # per-axis limits vxmin .. aymax; the short names vmin / vmax / amin / amax set both axes at once
            for stem, low, high in (('v', -0.5, 0.5), ('a', -1., 1.)):
                for end, default in (('min', low), ('max', high)):
                    for axis in 'xy':
                        setattr(self, stem + axis + end, given.get(stem + end, given.get(stem + axis + end, default)))
        elif self.options['syslimit'] == 'norm_2':
            self.vmax, self.amax = given.get('vmax', 0.5), given.get('amax', 1.)

    def _axis_limits(self):
        return [(self.vxmin, self.vxmax, self.axmin, self.axmax), (self.vymin, self.vymax, self.aymin, self.aymax)]

    def define_trajectory_constraints(self, splines, horizon_time):
        _PointMass.define_trajectory_constraints(self, splines, horizon_time)

    def get_initial_constraints(self, splines, horizon_time):
        return _PointMass.get_initial_constraints(self, splines, horizon_time)

    def set_initial_conditions(self, state, input=None):
        _PointMass.set_initial_conditions(self, state, input)
        self.prediction['dinput'] = np.zeros(2)

    def define_collision_constraints(self, hyperplanes, room, splines, horizon_time):
        self.define_collision_constraints_2d(hyperplanes, room, [splines[0], splines[1]], horizon_time)

    def splines2signals(self, splines, time):
        out, acc = self._signals(splines[:2], time)
        out['dinput'] = acc
        return out

    def state2pose(self, state):
        return np.r_[state, 0.]


class Holonomic3D(_PointMass):

    def __init__(self, shapes, options=None, bounds=None):
        given = bounds or {}
        Vehicle.__init__(self, n_spl=3, degree=3, shapes=shapes, options=options)
        self.vmin, self.vmax = given.get('vmin', -0.5), given.get('vmax', 0.5)
        self.amin, self.amax = given.get('amin', -1.), given.get('amax', 1.)

    def _axis_limits(self):
        return [(self.vmin, self.vmax, self.amin, self.amax)] * 3

    def get_init_spline_value(self):
        return _PointMass.get_init_spline_value(self)

    def define_collision_constraints(self, hyperplanes, room, splines, horizon_time=None):
        self.define_collision_constraints_3d(hyperplanes, room, list(splines[:3]), horizon_time)

    def splines2signals(self, splines, time):
        out, acc = self._signals(splines, time)
        out['a'] = acc
        return out

    def state2pose(self, state):
        return np.r_[state, np.zeros(3)]


class Quadrotor(Vehicle):
    """Planar quadrotor in its flat outputs (x, z): with pitch theta, thrust u1 and pitch rate u2,
    x'' = u1 sin(theta), z'' = u1 cos(theta) - g  =>  u1^2 = x''^2 + (z'' + g)^2,  u2 = (x''' (z'' + g) - x'' z''') / u1^2.
    In spline time tau = t / T every derivative carries a power of T, which the rows below multiply out."""

    def __init__(self, radius=0.2, options=None, bounds=None):
        given = bounds or {}
        Vehicle.__init__(self, n_spl=2, degree=4, shapes=Circle(radius), options=options)
        self.radius, self.g = radius, 9.81
        self.u1min, self.u1max = given.get('u1min', 2.), given.get('u1max', 15.)
        self.u2min, self.u2max = given.get('u2min', -8.), given.get('u2max', 8.)

    def set_default_options(self):
        Vehicle.set_default_options(self)
        self.options['stop_tol'] = 1.e-2

    def define_trajectory_constraints(self, splines, horizon_time=None):
        T = self.define_symbol('T') if horizon_time is None else horizon_time
        x, z = splines
        x2, z2, x3, z3 = x.derivative(2), z.derivative(2), x.derivative(3), z.derivative(3)
        lift = z2 + self.g * (T**2)                              # (z'' + g) T^2
        thrust2 = x2**2 + lift**2                                # u1^2 T^4
        self.define_constraint(-thrust2 + (T**4) * self.u1min**2, -inf, 0.)
        self.define_constraint(thrust2 - (T**4) * self.u1max**2, -inf, 0.)
        turning = x3 * lift - x2 * z3                            # u2 u1^2 T^5
        self.define_constraint(-turning + thrust2 * (T * self.u2min), -inf, 0.)
        self.define_constraint(turning - thrust2 * (T * self.u2max), -inf, 0.)

    def get_initial_constraints(self, splines, horizon_time=None):
        T = self.define_symbol('T') if horizon_time is None else horizon_time
        given = [self.define_parameter(name, 2) for name in ('spl0', 'dspl0', 'ddspl0')]
        scale = [1., T, T**2]
        rows = []
        for order in range(3):
            for k, s in enumerate(splines):
                rows.append((s if order == 0 else s.derivative(order), given[order][k] if order == 0 else scale[order] * given[order][k]))
        return rows

    def get_terminal_constraints(self, splines, horizon_time=None):
        target = self.define_parameter('poseT', 2)
        at_rest = [(s.derivative(order), 0.) for order in range(1, self.degree + 1) for s in splines]
        return [[(s, target[k]) for k, s in enumerate(splines)], at_rest]

    def set_initial_conditions(self, state, input=None):
        self.prediction['state'] = np.r_[np.asarray(state, float)[:2], np.zeros(3)]
        self.prediction['dspl'], self.prediction['ddspl'] = np.zeros(2), np.zeros(2)
        self._ic_version += 1

    def set_terminal_conditions(self, position):
        self.poseT = np.asarray(position, dtype=float)

    def get_init_spline_value(self):
        """Straight line with the first and last `degree` coefficients held at the end points (a plan that starts and ends at rest)."""
        start, goal = self.prediction['state'][:2], self.poseT
        n, d = len(self.basis), self.degree
        ramp = np.linspace(start, goal, n - 2 * d)
        return [np.vstack((np.tile(start, (d, 1)), ramp, np.tile(goal, (d, 1))))]

    def check_terminal_conditions(self):
        tol = self.options['stop_tol']
        off = np.linalg.norm(self.signals['pose'][:2, -1] - self.poseT)
        return not (off > tol or np.linalg.norm(self.signals['dspl'][:, -1]) > tol)

    def set_parameters(self, current_time):
        out = Vehicle.set_parameters(self, current_time)
        out[self].update(spl0=self.prediction['state'][:2], dspl0=self.prediction['dspl'], ddspl0=self.prediction['ddspl'],
                         poseT=self.poseT)
        return out

    def define_collision_constraints(self, hyperplanes, room, splines, horizon_time=None):
        self.define_collision_constraints_2d(hyperplanes, room, [splines[0], splines[1]], horizon_time)

    def splines2signals(self, splines, time):
        x, z = splines[0], splines[1]
        (xs, zs), (x1, z1), (x2, z2), (x3, z3) = (sample_splines([x.derivative(o) if o else x, z.derivative(o) if o else z], time)
                                                   for o in range(4))
        lift = z2 + self.g
        thrust2 = lift**2 + x2**2
        return {'state': np.c_[xs, zs, x1, z1, np.arctan2(x2, lift)].T,
                'input': np.c_[np.sqrt(x2**2 + lift**2), (x3 * lift - x2 * z3) / thrust2].T,
                'dspl': np.c_[x1, z1].T, 'ddspl': np.c_[x2, z2].T}

    def state2pose(self, state):
        return np.r_[state[0], state[1], -state[4]]

    def ode(self, state, input):
        pitch, thrust, rate = state[4], input[0], input[1]
        return np.r_[state[2:4], thrust * np.sin(pitch), thrust * np.cos(pitch) - self.g, rate]


# ---------------------------------------------------------------------------------------------------------------------------
class Fleet(PlotLayer):
    """Vehicles + who talks to whom + the formation they keep (`vehicles/fleet.py:37-98`)."""

    def __init__(self, vehicles=None, interconnection='circular'):
        PlotLayer.__init__(self)
        self.vehicles = _as_list(vehicles or [])
        self.interconnection = interconnection
        self.set_neighbors()

    def add_vehicle(self, vehicles):
        self.vehicles.extend(_as_list(vehicles))
        self.set_neighbors()

    def get_neighbors(self, vehicle):
        return self.nghb_list[vehicle]

    def set_neighbors(self):
        n = self.N = len(self.vehicles)
        if self.interconnection == 'circular':
            pick = lambda l: [(l + 1) % n, (l - 1) % n]                 # successor first, like the reference
        elif self.interconnection == 'full':
            pick = lambda l: [k for k in range(n) if k != l]
        else:
            raise ValueError('Interconnection type ' + self.interconnection + ' not understood.')
        self.nghb_list = dict((veh, [self.vehicles[k] for k in pick(l)]) for l, veh in enumerate(self.vehicles))

    def set_configuration(self, configuration, orientation=0.):
        """Per vehicle: its place in the formation, a list (planar ones are turned by -orientation) or a dict {output index: value}."""
        if len(configuration) != self.N:
            raise ValueError('You should provide configuration info for each vehicle.')
        c, s = np.cos(-orientation), np.sin(-orientation)
        self.configuration = {}
        for veh, place in zip(self.vehicles, configuration):
            if isinstance(place, dict):
                self.configuration[veh] = dict(place)
                continue
            place = list(place)
            if len(place) == 2:
                place = [place[0] * c - place[1] * s, place[0] * s + place[1] * c]
            self.configuration[veh] = dict(enumerate(place))
        self.set_rel_pos_c()
        self.rel_config = {}
        for veh in self.vehicles:
            here = self.configuration[veh]
            self.rel_config[veh] = {}
            for other in self.get_neighbors(veh):
                there = self.configuration[other]
                if len(here) != len(there):
                    raise ValueError('All vehicles should have same number of variables for which the configuration is imposed.')
                self.rel_config[veh][other] = [here[i] - there[j] for i, j in zip(sorted(here), sorted(there))]

    def set_rel_pos_c(self):
        if not hasattr(self, 'configuration'):
            raise ValueError('No configuration set!')
        for veh in self.vehicles:
            place = self.configuration[veh]
            veh.rel_pos_c = [-place[k] for k in sorted(place)]

    def get_rel_config(self, vehicle):
        return self.rel_config[vehicle]

    def _each(self, method, *columns):
        for veh, args in zip(self.vehicles, zip(*columns)):
            getattr(veh, method)(*args)

    def set_initial_conditions(self, states, inputs=None):
        self._each('set_initial_conditions', states, inputs if inputs is not None else [None] * len(states))

    def set_terminal_conditions(self, conditions):
        self._each('set_terminal_conditions', conditions)

    def overrule_state(self, states):
        self._each('overrule_state', states)

    def overrule_input(self, inputs):
        self._each('overrule_input', inputs)


def get_fleet_vehicles(var):
    """(fleet, vehicles) of a vehicle, a list of vehicles, a fleet or a list that holds one."""
    if isinstance(var, Fleet):
        return var, var.vehicles
    if isinstance(var, Vehicle):
        return Fleet(var), [var]
    if isinstance(var, list) and var:
        if isinstance(var[0], Fleet):
            return var[0], var[0].vehicles
        if isinstance(var[0], Vehicle):
            return Fleet(var), var
    raise TypeError('expected a Vehicle, a list of Vehicles or a Fleet')
