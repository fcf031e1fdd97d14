# This file is derived from OMG-tools (meco-group/omg-tools, `omgtools/environment/environment.py`, `obstacle.py`).
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

"""`Environment` and `Obstacle`: rooms, obstacle motion model and the
separating-hyperplane variables/constraints.

Behavioural spec: reference `environment/environment.py` (constructor 29-61,
copy 67-70, define_collision_constraints 102-146, init 182-184) and
`environment/obstacle.py` (Obstacle factory 34-42, ObstaclexD.init 80-121,
set_parameters 142-155, prepare_simulation 172-233, simulate 246-264,
Obstacle2D.define_collision_constraints 334-343, Obstacle3D 528-533).
Rotating obstacles (NURBS cos/sin, obstacle.py:299-332), bouncing and
inter-vehicle avoidance are outside the hot-path scope.
"""
import warnings

import numpy as np

from .opti import OptiChild, inf
from .plotting import PlotLayer
from .splines import BSplineBasis, BSpline
from .symbolic import vertcat


class Obstacle(object):
    def __new__(cls, initial, shape, simulation=None, options=None):
        simulation = simulation or {}
        options = options or {}
        if shape.n_dim == 2:
            return Obstacle2D(initial, shape, simulation, options)
        if shape.n_dim == 3:
            return Obstacle3D(initial, shape, simulation, options)
        raise ValueError('obstacle shape must be 2-D or 3-D')


class ObstaclexD(OptiChild):

    def __init__(self, initial, shape, simulation, options):
        OptiChild.__init__(self, 'obstacle')
        self.simulation = simulation
        initial = dict(initial)
        if 'trajectories' in simulation:
            for key, traj in simulation['trajectories'].items():
                if 0 in traj['time']:
                    initial[key] = traj['values'][list(traj['time']).index(0)]
        self.set_default_options()
        self.set_options(options)
        self.shape = shape
        self.n_dim = shape.n_dim
        self.basis = BSplineBasis([0, 0, 0, 1, 1, 1], 2)
        self.initial = initial
        self.prepare_simulation(initial, simulation)

    def set_default_options(self):
        self.options = {'draw': True, 'avoid': True, 'spline_traj': False,
                        'spline_params': {'knots': [0, 0, 0, 1, 1, 1], 'degree': 2,
                                          'coeffs': [0, 0, 0]}, 'bounce': False}

    def set_options(self, options):
        self.options.update(options)

    # -- optimisation modelling -------------------------------------------------
    def init(self, horizon_times=None):
        if not self.options['spline_traj']:
            x = self.define_parameter('x', self.n_dim)
            v = self.define_parameter('v', self.n_dim)
            a = self.define_parameter('a', self.n_dim)
            self.t = self.define_symbol('t')
            if horizon_times is None:
                horizon_times = [self.define_symbol('T')]
            elif not isinstance(horizon_times, list):
                horizon_times = [horizon_times]
            # state at the start of the horizon (t seconds ago)
            v0 = [v[k] - self.t * a[k] for k in range(self.n_dim)]
            pos0 = [x[k] - self.t * v0[k] - 0.5 * (self.t**2) * a[k] for k in range(self.n_dim)]
            self.pos_spline = [0] * self.n_dim
            for T in horizon_times:
                for k in range(self.n_dim):
                    self.pos_spline[k] = BSpline(self.basis, vertcat(
                        pos0[k], 0.5 * v0[k] * T + pos0[k],
                        pos0[k] + v0[k] * T + 0.5 * a[k] * (T**2)))
                pos0 = [self.pos_spline[k](1.) for k in range(self.n_dim)]
        else:
            self.basis = BSplineBasis(self.options['spline_params']['knots'],
                                      self.options['spline_params']['degree'])
            coeffs = self.define_parameter('traj_coeffs', len(self.basis), self.n_dim)
            self.pos_spline = [BSpline(self.basis, coeffs[:, k]) for k in range(self.n_dim)]
        checkpoints, _ = self.shape.get_checkpoints()
        self.checkpoints = np.atleast_1d(
            self.define_parameter('checkpoints', len(checkpoints) * self.n_dim))
        self.rad = np.atleast_1d(self.define_parameter('rad', len(checkpoints)))

    def define_collision_constraints(self, hyperplanes):
        raise ValueError('Please implement this method.')

    def set_parameters(self, current_time):
        parameters = {self: {}}
        if not self.options['spline_traj']:
            parameters[self]['x'] = self.signals['position'][:, -1]
            parameters[self]['v'] = self.signals['velocity'][:, -1]
            parameters[self]['a'] = self.signals['acceleration'][:, -1]
        else:
            parameters[self]['traj_coeffs'] = self.options['spline_params']['coeffs']
        checkpoints, rad = self.shape.get_checkpoints()
        parameters[self]['checkpoints'] = np.reshape(checkpoints, (len(checkpoints) * self.n_dim,))
        parameters[self]['rad'] = rad
        return parameters

    # -- deployment / simulation (host harness) -------------------------------------
    def set_state(self, dictionary):
        for key in ['position', 'velocity', 'acceleration']:
            if key in dictionary:
                self.signals[key] = np.c_[dictionary[key]]
            else:
                self.signals[key] = np.zeros((self.n_dim, 1))

    def prepare_simulation(self, initial, simulation):
        # events: at time tau add `delta` to position/velocity/acceleration
        self._events = []
        for l, key in enumerate(['position', 'velocity', 'acceleration']):
            traj = simulation.get('trajectories', {}).get(key)
            if traj is None:
                continue
            if len(traj['time']) != len(traj['values']):
                raise ValueError('Dimension mismatch between time array and values for ' +
                                 key + ' trajectory.')
            for tau, val in zip(traj['time'], traj['values']):
                if tau != 0:
                    self._events.append((float(tau), l, np.asarray(val, dtype=float)))
        self._events.sort(key=lambda e: e[0])
        self.signals = {'time': np.array([0.])}
        for key in ['position', 'velocity', 'acceleration']:
            self.signals[key] = np.c_[initial[key]].astype(float) if key in initial \
                else np.zeros((self.n_dim, 1))
        self.signals['orientation'] = np.array([[initial.get('orientation', 0.)]], dtype=float)
        self.signals['angular_velocity'] = np.array(
            [[initial.get('angular_velocity', 0.)]], dtype=float)

    def simulate(self, simulation_time, sample_time):
        """Double integrator with the user's step increments, advanced exactly
        per sample (reference: odeint, `obstacle.py:246-264`)."""
        n_samp = int(np.round(simulation_time / sample_time, 6))
        p = self.signals['position'][:, -1].copy()
        v = self.signals['velocity'][:, -1].copy()
        a = self.signals['acceleration'][:, -1].copy()
        t = self.signals['time'][-1]
        P, V, A, TT = [], [], [], []
        for _ in range(n_samp):
            p = p + v * sample_time + 0.5 * a * sample_time**2
            v = v + a * sample_time
            t_new = t + sample_time
            for tau, l, delta in self._events:
                if t < tau <= t_new + 1e-12:
                    if l == 0:
                        p = p + delta
                    elif l == 1:
                        v = v + delta
                    else:
                        a = a + delta
            t = t_new
            P.append(p.copy()); V.append(v.copy()); A.append(a.copy()); TT.append(t)
        if n_samp:
            self.signals['position'] = np.c_[self.signals['position'], np.array(P).T]
            self.signals['velocity'] = np.c_[self.signals['velocity'], np.array(V).T]
            self.signals['acceleration'] = np.c_[self.signals['acceleration'], np.array(A).T]
            self.signals['time'] = np.r_[self.signals['time'], TT]

    def draw(self, t=-1):
        return [], []


class Obstacle2D(ObstaclexD):

    def init(self, horizon_times=None):
        ObstaclexD.init(self, horizon_times=horizon_times)
        if self.signals['angular_velocity'][:, -1] != 0.:
            raise NotImplementedError('rotating obstacles are outside the hot-path scope')
        theta = self.signals['orientation'][:, -1][0]
        self.cos, self.sin = np.cos(theta), np.sin(theta)
        self.gon_weight = 1.

    def define_collision_constraints(self, hyperplanes):
        n = self.n_dim
        for hyperplane in hyperplanes:
            a, b = hyperplane['a'], hyperplane['b']
            for l in range(len(self.checkpoints) // n):
                cx, cy = self.checkpoints[l * n], self.checkpoints[l * n + 1]
                xpos = self.pos_spline[0] * self.gon_weight + cx * self.cos - cy * self.sin
                ypos = self.pos_spline[1] * self.gon_weight + cx * self.sin + cy * self.cos
                self.define_constraint(-(a[0] * xpos + a[1] * ypos) +
                                       self.gon_weight * (b + self.rad[l]), -inf, 0.)


class Obstacle3D(ObstaclexD):

    def define_collision_constraints(self, hyperplanes):
        n = self.n_dim
        for hyperplane in hyperplanes:
            a, b = hyperplane['a'], hyperplane['b']
            for l in range(len(self.checkpoints) // n):
                acc = a[0] * (self.checkpoints[l * n] + self.pos_spline[0])
                for k in range(1, n):
                    acc = acc + a[k] * (self.checkpoints[l * n + k] + self.pos_spline[k])
                self.define_constraint(-acc + b + self.rad[l], -inf, 0.)


class Environment(OptiChild, PlotLayer):

    def __init__(self, room, obstacles=None):
        OptiChild.__init__(self, 'environment')
        PlotLayer.__init__(self)
        self.room = room if isinstance(room, list) else [room]
        self.n_dim = self.room[0]['shape'].n_dim
        for room in self.room:
            if room['shape'].n_dim != self.n_dim:
                raise ValueError('You try to combine rooms of different dimensions, '
                                 'which is invalid')
            room.setdefault('position', [0.] * self.n_dim)
            room.setdefault('orientation', 0. if self.n_dim == 2 else [0., 0., 0.])
            room.setdefault('draw', False)
        self.obstacles, self.n_obs = [], 0
        for obstacle in (obstacles or []):
            self.add_obstacle(obstacle)

    def copy(self):
        obstacles = [Obstacle(o.initial, o.shape, o.simulation, o.options)
                     for o in self.obstacles]
        return Environment(self.room, obstacles)

    def add_obstacle(self, obstacle):
        if isinstance(obstacle, list):
            for obst in obstacle:
                self.add_obstacle(obst)
            return
        if obstacle.n_dim == 2 and self.n_dim == 3:
            warnings.warn('You are combining a 2D obstacle with a 3D environment. The 2D '
                          'obstacle is transformed to a 3D one by extending it infinitely '
                          'in z dimension.')
        if obstacle.n_dim == 3 and self.n_dim == 2:
            raise ValueError('Not possible to combine %dD obstacle with %dD environment.' %
                             (obstacle.n_dim, self.n_dim))
        self.obstacles.append(obstacle)
        self.n_obs += 1

    def define_collision_constraints(self, vehicle, splines, horizon_times):
        if vehicle.n_dim != self.n_dim:
            raise ValueError('Not possible to combine %dD vehicle with %dD environment.' %
                             (vehicle.n_dim, self.n_dim))
        horizon_times = horizon_times if isinstance(horizon_times, list) else [horizon_times]
        # hyperplanes are piecewise linear on the vehicle's breakpoints
        knots = np.r_[0., vehicle.knots[vehicle.degree:-vehicle.degree], 1.]
        basis = BSplineBasis(knots, 1)
        for idx in range(vehicle.n_seg):
            room = self.room[idx]
            hyp_veh, hyp_obs = {}, {}
            obs_to_add = room['obstacles'] if 'obstacles' in room else self.obstacles
            for k, shape in enumerate(vehicle.shapes):
                hyp_veh[shape] = []
                for l, obstacle in enumerate(obs_to_add):
                    obstacle.init(horizon_times=horizon_times[:idx + 1])
                    if not obstacle.options['avoid']:
                        continue
                    hyp_obs.setdefault(obstacle, [])
                    tag = '_' + vehicle.label + '_seg' + str(idx) + '_' + str(k) + str(l)
                    a = self.define_spline_variable('a' + tag, obstacle.n_dim, basis=basis)
                    b = self.define_spline_variable('b' + tag, 1, basis=basis)[0]
                    norm2 = a[0] * a[0]
                    for p in range(1, obstacle.n_dim):
                        norm2 = norm2 + a[p] * a[p]
                    self.define_constraint(norm2 - 1, -inf, 0.)
                    if self.n_dim == 3 and obstacle.n_dim == 2:
                        hyp_veh[shape].append(
                            {'a': [a[0], a[1], BSpline(basis, np.zeros(len(basis)))], 'b': b})
                    else:
                        hyp_veh[shape].append({'a': a, 'b': b})
                    hyp_obs[obstacle].append({'a': a, 'b': b})
                    obstacle.define_collision_constraints(hyp_obs[obstacle])
            vehicle.define_collision_constraints(hyp_veh, room, splines[idx], horizon_times[idx])

    def define_intervehicle_collision_constraints(self, vehicles, horizon_times):
        """One separating hyperplane per pair of vehicle shapes, seen with opposite signs by the two
        vehicles (`environment/environment.py:148-176`)."""
        horizon_times = horizon_times if isinstance(horizon_times, list) else [horizon_times]
        for idx in range(vehicles[0].n_seg):
            hyp_veh = {veh: {sh: [] for sh in veh.shapes} for veh in vehicles}
            for k in range(len(vehicles)):
                for l in range(k + 1, len(vehicles)):
                    veh1, veh2 = vehicles[k], vehicles[l]
                    if veh1 is veh2:
                        continue
                    if veh1.n_dim != veh2.n_dim:
                        raise ValueError('Not possible to combine %dD and %dD vehicle.' % (veh1.n_dim, veh2.n_dim))
                    knots = np.r_[0., np.union1d(veh1.knots[veh1.degree:-veh1.degree],
                                                 veh2.knots[veh2.degree:-veh2.degree]), 1.]
                    basis = BSplineBasis(knots, 1)
                    for kk, shape1 in enumerate(veh1.shapes):
                        for ll, shape2 in enumerate(veh2.shapes):
                            tag = '_%s_seg%d_%d_%s_%d' % (veh1.label, idx, kk, veh2.label, ll)
                            a = self.define_spline_variable('a' + tag, self.n_dim, basis=basis)
                            b = self.define_spline_variable('b' + tag, 1, basis=basis)[0]
                            norm2 = a[0] * a[0]
                            for p in range(1, self.n_dim):
                                norm2 = norm2 + a[p] * a[p]
                            self.define_constraint(norm2 - 1, -inf, 0.)
                            hyp_veh[veh1][shape1].append({'a': a, 'b': b})
                            hyp_veh[veh2][shape2].append({'a': [-a_i for a_i in a], 'b': -b})
            for vehicle in vehicles:
                vehicle.define_collision_constraints(hyp_veh[vehicle], self.room[idx], vehicle.splines[idx],
                                                     horizon_times[idx])

    def init(self, horizon_times=None):
        for obstacle in self.obstacles:
            obstacle.init(horizon_times=horizon_times)

    def simulate(self, simulation_time, sample_time):
        for obstacle in self.obstacles:
            obstacle.simulate(simulation_time, sample_time)
        self.update_plots()

    def draw(self, t=-1):
        return [], []
