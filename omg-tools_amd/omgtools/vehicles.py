# This file is derived from OMG-tools (meco-group/omg-tools, `omgtools/vehicles/vehicle.py`, `holonomic.py`, `holonomic3d.py`, `quadrotor.py`, `fleet.py`).
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

"""Vehicle models: `Vehicle`, `Holonomic`, `Holonomic3D`, `Quadrotor`, `Fleet`.

Same constructor/option names and the same constraint *definitions* (order
included, it fixes the row layout of g) as the reference's
`vehicles/vehicle.py` (define_knots 80-87, define_splines 105-120,
define_collision_constraints_2d 122-190, _3d 192-232, store 250-300,
predict 302-337, simulate 359-401), `vehicles/holonomic.py` (62-175),
`vehicles/holonomic3d.py` (44-152), `vehicles/quadrotor.py` (48-163) and
`vehicles/fleet.py` (25-98) -- written against the polynomial coefficient layer
(symbolic.py) instead of CasADi.
"""
import numpy as np

from .opti import OptiChild, inf
from .plotting import PlotLayer
from .shapes import Circle, Rectangle, Square
from .splines import (BSplineBasis, BSpline, definite_integral, concat_splines,
                      sample_splines)


class Vehicle(OptiChild, PlotLayer):

    def __init__(self, n_spl, degree, shapes, options=None):
        OptiChild.__init__(self, 'vehicle')
        PlotLayer.__init__(self)
        self.shapes = shapes if isinstance(shapes, list) else [shapes]
        self.n_dim = self.shapes[0].n_dim
        for shape in self.shapes:
            if shape.n_dim != self.n_dim:
                raise ValueError('All vehicle shapes should have same spatial dimension.')
        self.prediction = {}
        self.init_spline_values = None
        self.degree = degree
        self.to_simulate = True
        self.set_default_options()
        self.set_options(options or {})
        self.define_knots(knot_intervals=10)
        self.n_spl = n_spl

    def set_default_options(self):
        self.options = {'safety_distance': 0., 'safety_weight': 10.,
                        'room_constraints': True, 'stop_tol': 1.e-3,
                        'ideal_prediction': False, 'ideal_update': False,
                        '1storder_delay': False, 'time_constant': 0.1,
                        'input_disturbance': None}

    def set_options(self, options):
        self.options.update(options)

    def define_knots(self, **kwargs):
        if 'knot_intervals' in kwargs:
            self.knot_intervals = kwargs['knot_intervals']
            self.knots = np.r_[np.zeros(self.degree),
                               np.linspace(0., 1., self.knot_intervals + 1),
                               np.ones(self.degree)]
        if 'knots' in kwargs:
            self.knots = kwargs['knots']
        self.basis = BSplineBasis(self.knots, self.degree)

    def set_init_spline_values(self, values, n_seg=1):
        self.init_spline_values = [0] * n_seg
        for k in range(n_seg):
            if values[k].shape != (len(self.basis), self.n_spl):
                raise ValueError('Initial guess has wrong dimensions for spline %d, '
                                 'required: %s while you gave: %s' %
                                 (k, (len(self.basis), self.n_spl), values[k].shape))
            self.init_spline_values[k] = values[k]

    # -- optimisation modelling -----------------------------------------------
    def init(self):
        pass

    def define_splines(self, n_seg=1):
        self.n_seg = n_seg
        self.splines = []
        if self.init_spline_values is not None:
            init, self.init_spline_values = self.init_spline_values, None
        else:
            # as in the reference (`vehicle.py:112-115`) the construction-time value
            # is zero unless set_init_spline_values() was called; the initial guess
            # proper is installed by Problem.reinitialize()
            init = [None] * n_seg
        for k in range(n_seg):
            self.splines.append(self.define_spline_variable(
                'splines_seg' + str(k), self.n_spl, value=init[k]))
        return self.splines

    def _safety_slack(self, s, k, t, horizon_time):
        sd = self.options['safety_distance']
        if sd > 0.:
            eps = self.define_spline_variable('eps_' + str(s) + str(k))[0]
            self.define_objective(self.options['safety_weight'] *
                                  definite_integral(eps, t / horizon_time, 1.))
            self.define_constraint(eps - sd, -inf, 0.)
            self.define_constraint(-eps, -inf, 0.)
            return eps
        return 0.

    def define_collision_constraints_2d(self, hyperplanes, room, positions,
                                        horizon_time, tg_ha=0, offset=0):
        t = self.define_symbol('t')
        sd = self.options['safety_distance']
        positions = [positions] if not isinstance(positions[0], list) else positions
        for s, shape in enumerate(self.shapes):
            position = positions[s]
            checkpoints, rad = shape.get_checkpoints()
            if shape in hyperplanes:
                for k, hyperplane in enumerate(hyperplanes[shape]):
                    a, b = hyperplane['a'], hyperplane['b']
                    sl = hyperplane.get('slack', 1)
                    eps = self._safety_slack(s, k, t, horizon_time)
                    for l, chck in enumerate(checkpoints):
                        con = (a[0] * chck[0] + a[1] * chck[1]) * (1. - tg_ha**2)
                        con += (-a[0] * chck[1] + a[1] * chck[0]) * (2 * tg_ha)
                        pos0 = position[0] * (1 + tg_ha**2) + offset * (1 - tg_ha**2)
                        pos1 = position[1] * (1 + tg_ha**2) + offset * (2 * tg_ha)
                        con += a[0] * pos0 + a[1] * pos1
                        con += (-b + sl * rad[l] + sd - eps) * (1 + tg_ha**2)
                        self.define_constraint(con, -inf, 0)
            if self.options['room_constraints']:
                lims = room['shape'].get_canvas_limits()
                room_limits = [lims[k] + room['position'][k] for k in range(self.n_dim)]
                aligned_room = isinstance(room['shape'], (Rectangle, Square)) and \
                    room['shape'].orientation == 0.0
                aligned_veh = isinstance(shape, Circle) or \
                    (isinstance(shape, (Rectangle, Square)) and shape.orientation == 0)
                if aligned_room and aligned_veh and isinstance(tg_ha, (int, float)) and tg_ha == 0.:
                    for chck in checkpoints:
                        for k in range(self.n_dim):
                            self.define_constraint(
                                -(chck[k] + position[k]) + room_limits[k][0] + rad[0], -inf, 0.)
                            self.define_constraint(
                                (chck[k] + position[k]) - room_limits[k][1] + rad[0], -inf, 0.)
                else:
                    hyp_room = room['shape'].get_hyperplanes(position=room['position'])
                    for l, chck in enumerate(checkpoints):
                        for hpp in hyp_room.values():
                            con = (hpp['a'][0] * chck[0] + hpp['a'][1] * chck[1]) * (1. - tg_ha**2)
                            con += (-hpp['a'][0] * chck[1] + hpp['a'][1] * chck[0]) * (2 * tg_ha)
                            pos0 = position[0] * (1 + tg_ha**2) + offset * (1 - tg_ha**2)
                            pos1 = position[1] * (1 + tg_ha**2) + offset * (2 * tg_ha)
                            con += hpp['a'][0] * pos0 + hpp['a'][1] * pos1
                            con += (-hpp['b'] + rad[l]) * (1 + tg_ha**2)
                            self.define_constraint(con, -inf, 0)

    def define_collision_constraints_3d(self, hyperplanes, room, positions, horizon_time):
        t = self.define_symbol('t')
        sd = self.options['safety_distance']
        positions = [positions] if not isinstance(positions[0], list) else positions
        for s, shape in enumerate(self.shapes):
            position = positions[s]
            checkpoints, rad = shape.get_checkpoints()
            if shape in hyperplanes:
                for k, hyperplane in enumerate(hyperplanes[shape]):
                    a, b = hyperplane['a'], hyperplane['b']
                    eps = self._safety_slack(s, k, t, horizon_time)
                    for l, chck in enumerate(checkpoints):
                        con = a[0] * (chck[0] + position[0])
                        for q in range(1, 3):
                            con = con + a[q] * (chck[q] + position[q])
                        self.define_constraint(con - b + rad[l] + sd - eps, -inf, 0)
            if self.options['room_constraints']:
                lims = room['shape'].get_canvas_limits()
                room_limits = [lims[k] + room['position'][k] for k in range(self.n_dim)]
                for chck in checkpoints:
                    for k in range(3):
                        self.define_constraint(-(chck[k] + position[k]) + room_limits[k][0], -inf, 0.)
                        self.define_constraint((chck[k] + position[k]) - room_limits[k][1], -inf, 0.)

    def get_fleet_center(self, splines, rel_pos, substitute=True):
        rel_pos = list(rel_pos)
        center = [s + rp for s, rp in zip(splines, rel_pos)]
        if substitute:
            return self.define_substitute('fleet_center', center)
        return center

    def set_parameters(self, current_time):
        return {self: {}}

    # -- deployment ----------------------------------------------------------------
    def store(self, current_time, sample_time, spline_segments, segment_times,
              time_axis=None, **kwargs):
        if not isinstance(segment_times, list):
            segment_times = [segment_times]
        n_insert = kwargs['continuity'] - (self.degree - 1) if 'continuity' in kwargs else None
        self.result_spline_segments = spline_segments
        splines = concat_splines(spline_segments, segment_times, n_insert=n_insert)
        self.result_splines = splines
        horizon_time = sum(segment_times)
        if time_axis is None:
            n_samp = int(round(horizon_time / sample_time, 6)) + 1
            time_axis = np.linspace(0., (n_samp - 1) * sample_time, n_samp)
        self.trajectories = self.splines2signals(splines, time_axis)
        if not set(['state', 'input']).issubset(self.trajectories):
            raise ValueError('Signals should contain at least state, input and pose.')
        self.trajectories['time'] = time_axis - time_axis[0] + current_time
        self.trajectories['pose'] = self._state2pose(self.trajectories['state'])
        self.trajectories['splines'] = np.c_[sample_splines(splines, time_axis)]
        if hasattr(self, 'rel_pos_c') and 'fleet_center' not in self.trajectories:
            self.trajectories['fleet_center'] = np.c_[sample_splines(
                [s + rp for s, rp in zip(splines, self.rel_pos_c)], time_axis)]
        knots = splines[0].basis.knots
        t0 = knots[self.degree] + time_axis[0]
        time_axis_kn = np.r_[t0, [k for k in knots[self.degree + 1:-self.degree] if k > t0]]
        self.trajectories_kn = self.splines2signals(splines, time_axis_kn)
        self.trajectories_kn['time'] = time_axis_kn - time_axis_kn[0] + current_time
        self.trajectories_kn['pose'] = self._state2pose(self.trajectories_kn['state'])
        self.trajectories_kn['splines'] = np.c_[sample_splines(splines, time_axis_kn)]
        for traj in (self.trajectories, self.trajectories_kn):
            for key in traj:
                if traj[key].ndim == 1:
                    traj[key] = traj[key].reshape(1, -1)

    def predict(self, current_time, predict_time, sample_time, state0=None, input0=None,
                dinput0=None, delay=0, enforce_states=False, enforce_inputs=False):
        if enforce_states and enforce_inputs:
            if all(l is not None for l in [state0, input0, dinput0]):
                self.set_initial_conditions(state0, input=input0, dinput=dinput0)
            elif all(l is not None for l in [state0, input0]):
                self.set_initial_conditions(state0, input=input0)
            elif hasattr(self, 'signals'):
                self.set_initial_conditions(self.signals['state'][:, -1],
                                            self.signals['input'][:, -1],
                                            self.signals['dinput'][:, -1])
            return
        if enforce_states:
            if state0 is not None:
                self.set_initial_conditions(state0)
            elif hasattr(self, 'signals'):
                self.set_initial_conditions(self.signals['state'][:, -1])
            return
        n_samp = int(np.round(predict_time / sample_time, 6))
        if self.options['ideal_prediction']:
            for key in self.trajectories:
                self.prediction[key] = self.trajectories[key][:, n_samp + delay]
        else:
            for key in self.trajectories:
                if key not in ['state', 'input', 'pose']:
                    self.prediction[key] = self.trajectories[key][:, n_samp + delay]
            inp = self.trajectories['input'][:, delay:]
            if state0 is None:
                state0 = self.signals['state'][:, -n_samp - 1]
            state = self.integrate_ode(state0, inp, predict_time, sample_time)
            self.prediction['state'] = state[:, -1]
            self.prediction['input'] = self.trajectories['input'][:, n_samp + delay]
            self.prediction['pose'] = self._state2pose(state[:, -1])

    # -- simulation (host harness, numpy) ---------------------------------------------
    def overrule_state(self, state):
        state = np.array(state)
        self.signals['state'][:, -1] = state
        self.signals['pose'][:, -1] = self._state2pose(state)
        self.prediction['state'] = state
        self.prediction['pose'] = self._state2pose(state)

    def overrule_input(self, input, dinput=None):
        input = np.array(input)
        self.signals['input'][:, -1] = input
        self.prediction['input'] = input
        if dinput is not None:
            self.signals['dinput'][:, -1] = dinput
            self.prediction['dinput'] = dinput

    def simulate(self, simulation_time, sample_time):
        if self.to_simulate:
            if not hasattr(self, 'signals'):
                self.signals = {key: np.c_[self.trajectories[key][:, 0]]
                                for key in self.trajectories}
            n_samp = int(np.round(simulation_time / sample_time, 6))
            if self.options['ideal_update']:
                for key in self.trajectories:
                    self.signals[key] = np.c_[self.signals[key],
                                              self.trajectories[key][:, 1:n_samp + 1]]
            else:
                for key in self.trajectories:
                    if key not in ['state', 'input', 'pose']:
                        self.signals[key] = np.c_[self.signals[key],
                                                  self.trajectories[key][:, 1:n_samp + 1]]
                inp = self.trajectories['input']
                state = self.integrate_ode(self.signals['state'][:, -1], inp,
                                           simulation_time, sample_time)
                self.signals['input'] = np.c_[self.signals['input'], inp[:, 1:n_samp + 1]]
                self.signals['state'] = np.c_[self.signals['state'], state[:, 1:n_samp + 1]]
                self.signals['pose'] = np.c_[self.signals['pose'],
                                             self._state2pose(state[:, 1:n_samp + 1])]
        if not hasattr(self, 'traj_storage'):
            self.traj_storage, self.traj_storage_kn, self.pred_storage = {}, {}, {}
        repeat = int(simulation_time / sample_time)
        for memory, dic in ((self.traj_storage, self.trajectories),
                            (self.traj_storage_kn, self.trajectories_kn),
                            (self.pred_storage, self.prediction)):
            for key in dic:
                memory.setdefault(key, []).extend([dic[key]] * repeat)
        self.update_plots()

    def _state2pose(self, state):
        if state.ndim <= 1:
            return self.state2pose(state)
        return np.c_[[self.state2pose(state[:, k]) for k in range(state.shape[1])]].T

    def integrate_ode(self, state0, input, integration_time, sample_time, ode=None):
        """Classical RK4 on the sample grid with linearly interpolated inputs
        (reference: scipy odeint, `vehicle.py:412-423`; C++ export: RK4,
        `export/vehicles/Vehicle.cpp:82-110`)."""
        ode = self.ode if ode is None else ode
        n_samp = int(integration_time / sample_time) + 1
        state = np.zeros((len(state0), n_samp))
        state[:, 0] = state0

        def u(k, frac):
            k0 = min(k, input.shape[1] - 1)
            k1 = min(k + 1, input.shape[1] - 1)
            return (1 - frac) * input[:, k0] + frac * input[:, k1]

        h = sample_time
        for k in range(n_samp - 1):
            x = state[:, k]
            k1 = ode(x, u(k, 0.))
            k2 = ode(x + 0.5 * h * k1, u(k, 0.5))
            k3 = ode(x + 0.5 * h * k2, u(k, 0.5))
            k4 = ode(x + h * k3, u(k, 1.))
            state[:, k + 1] = x + h / 6. * (k1 + 2 * k2 + 2 * k3 + k4)
        return state


class Holonomic(Vehicle):

    def __init__(self, shapes=None, options=None, bounds=None):
        bounds = bounds or {}
        Vehicle.__init__(self, n_spl=2, degree=3,
                         shapes=Circle(0.1) if shapes is None else shapes, options=options)
        if self.options.get('syslimit', 'norm_inf') == 'norm_inf':
            for name, default in (('vxmin', -0.5), ('vymin', -0.5), ('vxmax', 0.5),
                                  ('vymax', 0.5), ('axmin', -1.), ('aymin', -1.),
                                  ('axmax', 1.), ('aymax', 1.)):
                setattr(self, name, bounds.get(name, default))
            for short, pair in (('vmin', ('vxmin', 'vymin')), ('vmax', ('vxmax', 'vymax')),
                                ('amin', ('axmin', 'aymin')), ('amax', ('axmax', 'aymax'))):
                if short in bounds:
                    for name in pair:
                        setattr(self, name, bounds[short])
        elif self.options['syslimit'] == 'norm_2':
            self.vmax = bounds.get('vmax', 0.5)
            self.amax = bounds.get('amax', 1.)

    def set_default_options(self):
        Vehicle.set_default_options(self)
        self.options.update({'syslimit': 'norm_inf'})

    def define_trajectory_constraints(self, splines, horizon_time):
        x, y = splines
        dx, dy = x.derivative(), y.derivative()
        ddx, ddy = x.derivative(2), y.derivative(2)
        T = horizon_time
        if self.options['syslimit'] == 'norm_2':
            self.define_constraint((dx**2 + dy**2) - (T**2) * self.vmax**2, -inf, 0.)
            self.define_constraint((ddx**2 + ddy**2) - (T**4) * self.amax**2, -inf, 0.)
        elif self.options['syslimit'] == 'norm_inf':
            self.define_constraint(-dx + T * self.vxmin, -inf, 0.)
            self.define_constraint(-dy + T * self.vymin, -inf, 0.)
            self.define_constraint(dx - T * self.vxmax, -inf, 0.)
            self.define_constraint(dy - T * self.vymax, -inf, 0.)
            self.define_constraint(-ddx + (T**2) * self.axmin, -inf, 0.)
            self.define_constraint(-ddy + (T**2) * self.aymin, -inf, 0.)
            self.define_constraint(ddx - (T**2) * self.axmax, -inf, 0.)
            self.define_constraint(ddy - (T**2) * self.aymax, -inf, 0.)
        else:
            raise ValueError('Only norm_2 and norm_inf are defined as system limit.')

    def get_initial_constraints(self, splines, horizon_time):
        state0 = self.define_parameter('state0', 2)
        input0 = self.define_parameter('input0', 2)
        x, y = splines
        dx, dy = x.derivative(), y.derivative()
        return [(x, state0[0]), (y, state0[1]),
                (dx, horizon_time * input0[0]), (dy, horizon_time * input0[1])]

    def get_terminal_constraints(self, splines, horizon_time=None):
        position = self.define_parameter('poseT', 2)
        x, y = splines
        term_con = [(x, position[0]), (y, position[1])]
        term_con_der = []
        for d in range(1, self.degree + 1):
            term_con_der.extend([(x.derivative(d), 0.), (y.derivative(d), 0.)])
        return [term_con, term_con_der]

    def set_initial_conditions(self, state, input=None):
        self.prediction['state'] = np.asarray(state, dtype=float)
        self.prediction['input'] = np.zeros(2) if input is None else np.asarray(input, float)
        self.prediction['dinput'] = np.zeros(2)

    def set_terminal_conditions(self, position):
        self.poseT = np.asarray(position, dtype=float)

    def get_init_spline_value(self, subgoals=None):
        pos0, posT = self.prediction['state'], self.poseT
        init = np.zeros((len(self.basis), 2))
        for k in range(2):
            init[:, k] = np.linspace(pos0[k], posT[k], len(self.basis))
        return [init]

    def check_terminal_conditions(self):
        tol = self.options['stop_tol']
        return not (np.linalg.norm(self.signals['state'][:, -1] - self.poseT) > tol or
                    np.linalg.norm(self.signals['input'][:, -1]) > tol)

    def set_parameters(self, current_time):
        parameters = Vehicle.set_parameters(self, current_time)
        parameters[self]['state0'] = self.prediction['state']
        parameters[self]['input0'] = self.prediction['input']
        parameters[self]['poseT'] = self.poseT
        return parameters

    def define_collision_constraints(self, hyperplanes, room, splines, horizon_time):
        self.define_collision_constraints_2d(hyperplanes, room, [splines[0], splines[1]],
                                             horizon_time)

    def splines2signals(self, splines, time):
        x, y = splines[0], splines[1]
        inp = np.c_[sample_splines([x.derivative(), y.derivative()], time)]
        return {'state': np.c_[sample_splines([x, y], time)], 'input': inp,
                'v_tot': np.sqrt(inp[0, :]**2 + inp[1, :]**2),
                'dinput': np.c_[sample_splines([x.derivative(2), y.derivative(2)], time)]}

    def state2pose(self, state):
        return np.r_[state, 0.]

    def ode(self, state, input):
        return input


class Holonomic3D(Vehicle):

    def __init__(self, shapes, options=None, bounds=None):
        bounds = bounds or {}
        Vehicle.__init__(self, n_spl=3, degree=3, shapes=shapes, options=options)
        self.vmin, self.vmax = bounds.get('vmin', -0.5), bounds.get('vmax', 0.5)
        self.amin, self.amax = bounds.get('amin', -1.), bounds.get('amax', 1.)

    def set_default_options(self):
        Vehicle.set_default_options(self)
        self.options.update({'syslimit': 'norm_inf'})

    def define_trajectory_constraints(self, splines, horizon_time=None):
        T = self.define_symbol('T') if horizon_time is None else horizon_time
        d1 = [s.derivative() for s in splines]
        d2 = [s.derivative(2) for s in splines]
        if self.options['syslimit'] == 'norm_2':
            self.define_constraint((d1[0]**2 + d1[1]**2 + d1[2]**2) - (T**2) * self.vmax**2, -inf, 0.)
            self.define_constraint((d2[0]**2 + d2[1]**2 + d2[2]**2) - (T**4) * self.amax**2, -inf, 0.)
        elif self.options['syslimit'] == 'norm_inf':
            for s in d1:
                self.define_constraint(-s + T * self.vmin, -inf, 0.)
            for s in d1:
                self.define_constraint(s - T * self.vmax, -inf, 0.)
            for s in d2:
                self.define_constraint(-s + (T**2) * self.amin, -inf, 0.)
            for s in d2:
                self.define_constraint(s - (T**2) * self.amax, -inf, 0.)
        else:
            raise ValueError('Only norm_2 and norm_inf are defined as system limit.')

    def get_initial_constraints(self, splines, horizon_time=None):
        T = self.define_symbol('T') if horizon_time is None else horizon_time
        state0 = self.define_parameter('state0', 3)
        input0 = self.define_parameter('input0', 3)
        return [(splines[k], state0[k]) for k in range(3)] + \
               [(splines[k].derivative(), T * input0[k]) for k in range(3)]

    def get_terminal_constraints(self, splines, horizon_time=None):
        position = self.define_parameter('poseT', 3)
        term_con = [(splines[k], position[k]) for k in range(3)]
        term_con_der = []
        for d in range(1, self.degree + 1):
            term_con_der.extend([(s.derivative(d), 0.) for s in splines])
        return [term_con, term_con_der]

    def set_initial_conditions(self, state, input=None):
        self.prediction['state'] = np.asarray(state, dtype=float)
        self.prediction['input'] = np.zeros(3) if input is None else np.asarray(input, float)

    def set_terminal_conditions(self, position):
        self.poseT = np.asarray(position, dtype=float)

    def get_init_spline_value(self):
        pos0, posT = self.prediction['state'], self.poseT
        init = np.zeros((len(self.basis), 3))
        for k in range(3):
            init[:, k] = np.linspace(pos0[k], posT[k], len(self.basis))
        return [init]

    def check_terminal_conditions(self):
        tol = self.options['stop_tol']
        return not (np.linalg.norm(self.signals['state'][:, -1] - self.poseT) > tol or
                    np.linalg.norm(self.signals['input'][:, -1]) > tol)

    def set_parameters(self, current_time):
        parameters = Vehicle.set_parameters(self, current_time)
        parameters[self]['state0'] = self.prediction['state']
        parameters[self]['input0'] = self.prediction['input']
        parameters[self]['poseT'] = self.poseT
        return parameters

    def define_collision_constraints(self, hyperplanes, room, splines, horizon_time=None):
        self.define_collision_constraints_3d(hyperplanes, room, list(splines[:3]), horizon_time)

    def splines2signals(self, splines, time):
        inp = np.c_[sample_splines([s.derivative() for s in splines], time)]
        return {'state': np.c_[sample_splines(splines, time)], 'input': inp,
                'v_tot': np.sqrt((inp**2).sum(axis=0)),
                'a': np.c_[sample_splines([s.derivative(2) for s in splines], time)]}

    def state2pose(self, state):
        return np.r_[state, np.zeros(3)]

    def ode(self, state, input):
        return input


class Quadrotor(Vehicle):

    def __init__(self, radius=0.2, options=None, bounds=None):
        bounds = bounds or {}
        Vehicle.__init__(self, n_spl=2, degree=4, shapes=Circle(radius), options=options)
        self.radius = radius
        self.u1min, self.u1max = bounds.get('u1min', 2.), bounds.get('u1max', 15.)
        self.u2min, self.u2max = bounds.get('u2min', -8.), bounds.get('u2max', 8.)
        self.g = 9.81

    def set_default_options(self):
        Vehicle.set_default_options(self)
        self.options['stop_tol'] = 1.e-2

    def define_trajectory_constraints(self, splines, horizon_time=None):
        T = self.define_symbol('T') if horizon_time is None else horizon_time
        x, y = splines
        ddx, ddy = x.derivative(2), y.derivative(2)
        dddx, dddy = x.derivative(3), y.derivative(3)
        g_tf = self.g * (T**2)
        thrust2 = ddx**2 + (ddy + g_tf)**2
        self.define_constraint(-thrust2 + (T**4) * self.u1min**2, -inf, 0.)
        self.define_constraint(thrust2 - (T**4) * self.u1max**2, -inf, 0.)
        rate = dddx * (ddy + g_tf) - ddx * dddy
        self.define_constraint(-rate + thrust2 * (T * self.u2min), -inf, 0.)
        self.define_constraint(rate - thrust2 * (T * self.u2max), -inf, 0.)

    def get_initial_constraints(self, splines, horizon_time=None):
        T = self.define_symbol('T') if horizon_time is None else horizon_time
        spl0 = self.define_parameter('spl0', 2)
        dspl0 = self.define_parameter('dspl0', 2)
        ddspl0 = self.define_parameter('ddspl0', 2)
        x, y = splines
        return [(x, spl0[0]), (y, spl0[1]),
                (x.derivative(), T * dspl0[0]), (y.derivative(), T * dspl0[1]),
                (x.derivative(2), (T**2) * ddspl0[0]), (y.derivative(2), (T**2) * ddspl0[1])]

    def get_terminal_constraints(self, splines, horizon_time=None):
        position = self.define_parameter('poseT', 2)
        x, y = splines
        term_con = [(x, position[0]), (y, position[1])]
        term_con_der = []
        for d in range(1, self.degree + 1):
            term_con_der.extend([(x.derivative(d), 0.), (y.derivative(d), 0.)])
        return [term_con, term_con_der]

    def set_initial_conditions(self, state, input=None):
        self.prediction['state'] = np.r_[np.asarray(state, float)[:2], np.zeros(3)]
        self.prediction['dspl'] = np.zeros(2)
        self.prediction['ddspl'] = np.zeros(2)

    def set_terminal_conditions(self, position):
        self.poseT = np.asarray(position, dtype=float)

    def get_init_spline_value(self):
        pos0, posT = self.prediction['state'][:2], self.poseT
        n, d = len(self.basis), self.degree
        init = np.zeros((n, 2))
        for k in range(2):
            init[:, k] = np.r_[pos0[k] * np.ones(d), np.linspace(pos0[k], posT[k], n - 2 * d),
                               posT[k] * np.ones(d)]
        return [init]

    def check_terminal_conditions(self):
        tol = self.options['stop_tol']
        return not (np.linalg.norm(self.signals['pose'][:2, -1] - self.poseT) > tol or
                    np.linalg.norm(self.signals['dspl'][:, -1]) > tol)

    def set_parameters(self, current_time):
        parameters = Vehicle.set_parameters(self, current_time)
        parameters[self]['spl0'] = self.prediction['state'][:2]
        parameters[self]['dspl0'] = self.prediction['dspl']
        parameters[self]['ddspl0'] = self.prediction['ddspl']
        parameters[self]['poseT'] = self.poseT
        return parameters

    def define_collision_constraints(self, hyperplanes, room, splines, horizon_time=None):
        self.define_collision_constraints_2d(hyperplanes, room, [splines[0], splines[1]],
                                             horizon_time)

    def splines2signals(self, splines, time):
        x, y = splines[0], splines[1]
        x_s, y_s = sample_splines([x, y], time)
        dx_s, dy_s = sample_splines([x.derivative(), y.derivative()], time)
        ddx_s, ddy_s = sample_splines([x.derivative(2), y.derivative(2)], time)
        dddx_s, dddy_s = sample_splines([x.derivative(3), y.derivative(3)], time)
        theta = np.arctan2(ddx_s, ddy_s + self.g)
        u1 = np.sqrt(ddx_s**2 + (ddy_s + self.g)**2)
        u2 = (dddx_s * (ddy_s + self.g) - ddx_s * dddy_s) / ((ddy_s + self.g)**2 + ddx_s**2)
        return {'state': np.c_[x_s, y_s, dx_s, dy_s, theta].T, 'input': np.c_[u1, u2].T,
                'dspl': np.c_[dx_s, dy_s].T, 'ddspl': np.c_[ddx_s, ddy_s].T}

    def state2pose(self, state):
        return np.r_[state[0], state[1], -state[4]]

    def ode(self, state, input):
        theta, u1, u2 = state[4], input[0], input[1]
        return np.r_[state[2:4], u1 * np.sin(theta), u1 * np.cos(theta) - self.g, u2]


# ---------------------------------------------------------------------------
class Fleet(PlotLayer):
    """Vehicle container with neighbour topology and formation configuration
    (`vehicles/fleet.py:37-98`)."""

    def __init__(self, vehicles=None, interconnection='circular'):
        PlotLayer.__init__(self)
        vehicles = vehicles or []
        self.vehicles = vehicles if isinstance(vehicles, list) else [vehicles]
        self.interconnection = interconnection
        self.set_neighbors()

    def add_vehicle(self, vehicles):
        self.vehicles.extend(vehicles if isinstance(vehicles, list) else [vehicles])
        self.set_neighbors()

    def get_neighbors(self, vehicle):
        return self.nghb_list[vehicle]

    def set_neighbors(self):
        self.N = len(self.vehicles)
        self.nghb_list = {}
        for l, vehicle in enumerate(self.vehicles):
            if self.interconnection == 'circular':
                nghb_ind = [(self.N + l + 1) % self.N, (self.N + l - 1) % self.N]
            elif self.interconnection == 'full':
                nghb_ind = [k for k in range(self.N) if k != l]
            else:
                raise ValueError('Interconnection type ' + self.interconnection +
                                 ' not understood.')
            self.nghb_list[vehicle] = [self.vehicles[ind] for ind in nghb_ind]

    def set_configuration(self, configuration, orientation=0.):
        if len(configuration) != self.N:
            raise ValueError('You should provide configuration info for each vehicle.')
        cth, sth = np.cos(-orientation), np.sin(-orientation)
        self.configuration = {}
        for l, config in enumerate(configuration):
            if isinstance(config, dict):
                self.configuration[self.vehicles[l]] = dict(config)
                continue
            config = list(config)
            if len(config) == 2:
                config = [config[0] * cth - config[1] * sth, config[0] * sth + config[1] * cth]
            self.configuration[self.vehicles[l]] = {k: c for k, c in enumerate(config)}
        self.set_rel_pos_c()
        self.rel_config = {}
        for vehicle in self.vehicles:
            mine = self.configuration[vehicle]
            self.rel_config[vehicle] = {}
            for nghb in self.get_neighbors(vehicle):
                theirs = self.configuration[nghb]
                if len(mine) != len(theirs):
                    raise ValueError('All vehicles should have same number of variables '
                                     'for which the configuration is imposed.')
                self.rel_config[vehicle][nghb] = [
                    mine[a] - theirs[b] for a, b in zip(sorted(mine), sorted(theirs))]

    def set_rel_pos_c(self):
        if not hasattr(self, 'configuration'):
            raise ValueError('No configuration set!')
        for veh in self.vehicles:
            veh.rel_pos_c = [-self.configuration[veh][k] for k in sorted(self.configuration[veh])]

    def get_rel_config(self, vehicle):
        return self.rel_config[vehicle]

    def set_initial_conditions(self, states, inputs=None):
        inputs = [None] * len(states) if inputs is None else inputs
        for state, inp, vehicle in zip(states, inputs, self.vehicles):
            vehicle.set_initial_conditions(state, inp)

    def set_terminal_conditions(self, conditions):
        for condition, vehicle in zip(conditions, self.vehicles):
            vehicle.set_terminal_conditions(condition)

    def overrule_state(self, states):
        for state, vehicle in zip(states, self.vehicles):
            vehicle.overrule_state(state)

    def overrule_input(self, inputs):
        for inp, vehicle in zip(inputs, self.vehicles):
            vehicle.overrule_input(inp)


def get_fleet_vehicles(var):
    if isinstance(var, Fleet):
        return var, var.vehicles
    if isinstance(var, list):
        if isinstance(var[0], Vehicle):
            return Fleet(var), var
        if isinstance(var[0], Fleet):
            return var[0], var[0].vehicles
    if isinstance(var, Vehicle):
        return Fleet(var), [var]
    raise TypeError('expected a Vehicle, a list of Vehicles or a Fleet')
