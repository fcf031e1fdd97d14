# This file is derived from OMG-tools (meco-group/omg-tools, `omgtools/environment/environment.py`, `omgtools/environment/obstacle.py` (API surface, option keys, definition order)).
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
"""`Environment` and `Obstacle`: rooms, the obstacles' motion model, and the separating hyperplanes between every vehicle shape
and every obstacle -- written for this package against the behaviour of the reference's `environment/environment.py`
(constructor 29-61, copy 67-70, define_collision_constraints 102-146, inter-vehicle hyperplanes 148-176, init 182-184) and
`environment/obstacle.py` (factory 34-42, ObstaclexD.init 80-121, set_parameters 142-155, prepare_simulation 172-233,
simulate 246-264, Obstacle2D rows 334-343, Obstacle3D rows 528-533): same class names, constructor arguments, option keys,
variable / parameter / constraint names and definition ORDER (the flat x / p / g layouts are part of the drop-in boundary,
`tests/golden/nlp_*.npz`).  Rotating obstacles (`obstacle.py:299-332`) and bouncing are outside this module's scope.

The model, in the words of this package: an obstacle is a point that moves with piecewise constant acceleration (`_Motion`),
seen by the optimisation problem as a quadratic B-spline over the horizon whose three coefficients follow from position,
velocity and acceleration at the time of the solve; a hyperplane is a pair of degree-1 splines (a, b) on the vehicle's
breakpoints with |a| <= 1, a . (vehicle point) - b <= -r_vehicle on one side and a . (obstacle point) - b >= r_obstacle on the other.
"""
import warnings

import numpy as np

from .opti import OptiChild, inf
from .plotting import PlotLayer
from .splines import BSplineBasis, BSpline
from .symbolic import vertcat

_KINDS = ('position', 'velocity', 'acceleration')


class _Motion(object):
    """Point mass with piecewise constant acceleration and the user's jumps (`simulation['trajectories']`: at time tau ADD the
    value to position / velocity / acceleration), sampled exactly: x += v h + a h^2 / 2, v += a h per sample."""

    def __init__(self, n_dim, initial, trajectories):
        self.jumps = []
        for kind, traj in (trajectories or {}).items():
            if kind not in _KINDS:
                continue
            if len(traj['time']) != len(traj['values']):
                raise ValueError('Dimension mismatch between time array and values for ' + kind + ' trajectory.')
            self.jumps += [(float(tau), _KINDS.index(kind), np.asarray(val, dtype=float))
                           for tau, val in zip(traj['time'], traj['values']) if tau != 0]
        self.jumps.sort(key=lambda j: j[0])
        self.start = [np.c_[initial[k]].astype(float) if k in initial else np.zeros((n_dim, 1)) for k in _KINDS]

    def advance(self, state, t, h, n):
        """n samples of length h from (x, v, a) at time t: [n + 1 states incl. the first], times."""
        cur = [s.copy() for s in state]
        rows, times = [], []
        for _ in range(n):
            cur[0] = cur[0] + h * cur[1] + (0.5 * h * h) * cur[2]
            cur[1] = cur[1] + h * cur[2]
            for tau, which, jump in self.jumps:
                if t < tau <= t + h + 1e-12:
                    cur[which] = cur[which] + jump
            t += h
            rows.append([c.copy() for c in cur])
            times.append(t)
        return rows, times


class Obstacle(object):
    """Factory: `Obstacle(initial, shape, simulation, options)` gives the class of the shape's dimension."""

    def __new__(cls, initial, shape, simulation=None, options=None):
        try:
            kind = {2: Obstacle2D, 3: Obstacle3D}[shape.n_dim]
        except KeyError:
            raise ValueError('obstacle shape must be 2-D or 3-D')
        return kind(initial, shape, simulation or {}, options or {})


class ObstaclexD(OptiChild):

    def __init__(self, initial, shape, simulation, options):
        OptiChild.__init__(self, 'obstacle')
        self.shape, self.n_dim = shape, shape.n_dim
        self.simulation = simulation
        start = dict(initial)
        # a trajectory entry at time 0 replaces the initial value of that quantity
        for kind, traj in simulation.get('trajectories', {}).items():
            times = list(traj['time'])
            if 0 in times:
                start[kind] = traj['values'][times.index(0)]
        self.initial = start
        self.set_default_options()
        self.set_options(options)
        self.basis = BSplineBasis([0, 0, 0, 1, 1, 1], 2)
        self.prepare_simulation(start, simulation)

    def set_default_options(self):
        quadratic = {'knots': [0, 0, 0, 1, 1, 1], 'degree': 2, 'coeffs': [0, 0, 0]}
        self.options = {'draw': True, 'avoid': True, 'bounce': False, 'spline_traj': False, 'spline_params': quadratic}

    def set_options(self, options):
        self.options.update(options)

    # -- what the optimisation problem sees ----------------------------------------------------------------------------------
    def init(self, horizon_times=None):
        nd = self.n_dim
        if self.options['spline_traj']:
            spec = self.options['spline_params']
            self.basis = BSplineBasis(spec['knots'], spec['degree'])
            table = self.define_parameter('traj_coeffs', len(self.basis), nd)
            self.pos_spline = [BSpline(self.basis, table[:, k]) for k in range(nd)]
        else:
            x, v, a = (self.define_parameter(name, nd) for name in 'xva')
            self.t = self.define_symbol('t')
            if horizon_times is None:
                horizon_times = [self.define_symbol('T')]
            spans = horizon_times if isinstance(horizon_times, list) else [horizon_times]
            # x, v, a are the values NOW; the horizon began t seconds ago: roll the state back to its start ...
            v_start = [v[k] - self.t * a[k] for k in range(nd)]
            at = [x[k] - self.t * v_start[k] - 0.5 * (self.t**2) * a[k] for k in range(nd)]
            # ... and express x(tau T) = at + v_start tau T + a (tau T)^2 / 2 in the quadratic Bernstein basis, span after span
            self.pos_spline = [0] * nd
            for T in spans:
                for k in range(nd):
                    mid = 0.5 * v_start[k] * T + at[k]
                    end = at[k] + v_start[k] * T + 0.5 * a[k] * (T**2)
                    self.pos_spline[k] = BSpline(self.basis, vertcat(at[k], mid, end))
                at = [self.pos_spline[k](1.) for k in range(nd)]
        points, _ = self.shape.get_checkpoints()
        self.checkpoints = np.atleast_1d(self.define_parameter('checkpoints', nd * len(points)))
        self.rad = np.atleast_1d(self.define_parameter('rad', len(points)))

    def define_collision_constraints(self, hyperplanes):
        raise ValueError('Please implement this method.')

    def _point_count(self):
        return len(self.checkpoints) // self.n_dim

    def set_parameters(self, current_time):
        mine = {}
        if self.options['spline_traj']:
            mine['traj_coeffs'] = self.options['spline_params']['coeffs']
        else:
            for name, kind in zip('xva', _KINDS):
                mine[name] = self.signals[kind][:, -1]
        points, radii = self.shape.get_checkpoints()
        mine['checkpoints'] = np.reshape(points, (self.n_dim * len(points),))
        mine['rad'] = radii
        return {self: mine}

    # -- host-side simulation --------------------------------------------------------------------------------------------
    def set_state(self, dictionary):
        for kind in _KINDS:
            self.signals[kind] = np.c_[dictionary[kind]] if kind in dictionary else np.zeros((self.n_dim, 1))

    def prepare_simulation(self, initial, simulation):
        self._motion = _Motion(self.n_dim, initial, simulation.get('trajectories'))
        self._events = self._motion.jumps
        self.signals = dict(zip(_KINDS, self._motion.start))
        self.signals['time'] = np.zeros(1)
        for name in ('orientation', 'angular_velocity'):
            self.signals[name] = np.array([[initial.get(name, 0.)]], dtype=float)

    def simulate(self, simulation_time, sample_time):
        """`obstacle.py:246-264` integrates the same model with odeint; here every sample is exact."""
        n = int(np.round(simulation_time / sample_time, 6))
        if n <= 0:
            return
        now = [self.signals[kind][:, -1] for kind in _KINDS]
        rows, times = self._motion.advance(now, self.signals['time'][-1], sample_time, n)
        for i, kind in enumerate(_KINDS):
            self.signals[kind] = np.c_[self.signals[kind], np.array([r[i] for r in rows]).T]
        self.signals['time'] = np.r_[self.signals['time'], times]

    def draw(self, t=-1):
        return [], []


class Obstacle2D(ObstaclexD):

    def init(self, horizon_times=None):
        ObstaclexD.init(self, horizon_times=horizon_times)
        if self.signals['angular_velocity'][:, -1] != 0.:
            raise NotImplementedError('rotating obstacles are outside the hot-path scope')
        heading = self.signals['orientation'][:, -1][0]
        self.cos, self.sin = np.cos(heading), np.sin(heading)
        self.gon_weight = 1.

    def define_collision_constraints(self, hyperplanes):
        """Every checkpoint of the shape, turned by the obstacle's heading and carried by the position spline, on the far side of
        every hyperplane by at least its radius: -(a . q) + w (b + r) <= 0."""
        w, c, s = self.gon_weight, self.cos, self.sin
        for plane in hyperplanes:
            a, b = plane['a'], plane['b']
            for l in range(self._point_count()):
                px, py = self.checkpoints[2 * l], self.checkpoints[2 * l + 1]
                qx = self.pos_spline[0] * w + px * c - py * s
                qy = self.pos_spline[1] * w + px * s + py * c
                self.define_constraint(-(a[0] * qx + a[1] * qy) + w * (b + self.rad[l]), -inf, 0.)


class Obstacle3D(ObstaclexD):

    def define_collision_constraints(self, hyperplanes):
        nd = self.n_dim
        for plane in hyperplanes:
            a, b = plane['a'], plane['b']
            for l in range(self._point_count()):
                reach = a[0] * (self.checkpoints[nd * l] + self.pos_spline[0])
                for k in range(1, nd):
                    reach = reach + a[k] * (self.checkpoints[nd * l + k] + self.pos_spline[k])
                self.define_constraint(-reach + b + self.rad[l], -inf, 0.)


def _breakpoints(vehicle):
    """Interior breakpoints of a vehicle's spline basis, with the ends of the unit interval."""
    d = vehicle.degree
    return vehicle.knots[d:-d]


class Environment(OptiChild, PlotLayer):

    def __init__(self, room, obstacles=None):
        OptiChild.__init__(self, 'environment')
        PlotLayer.__init__(self)
        self.room = list(room) if isinstance(room, list) else [room]
        self.n_dim = self.room[0]['shape'].n_dim
        upright = 0. if self.n_dim == 2 else [0., 0., 0.]
        for part in self.room:
            if part['shape'].n_dim != self.n_dim:
                raise ValueError('You try to combine rooms of different dimensions, which is invalid')
            for key, default in (('position', [0.] * self.n_dim), ('orientation', upright), ('draw', False)):
                part.setdefault(key, default)
        self.obstacles, self.n_obs = [], 0
        self.add_obstacle(list(obstacles or []))

    def copy(self):
        return Environment(self.room, [Obstacle(o.initial, o.shape, o.simulation, o.options) for o in self.obstacles])

    def add_obstacle(self, obstacle):
        for one in (obstacle if isinstance(obstacle, list) else [obstacle]):
            if isinstance(one, list):
                self.add_obstacle(one)
                continue
            if one.n_dim > self.n_dim:
                raise ValueError('Not possible to combine %dD obstacle with %dD environment.' % (one.n_dim, self.n_dim))
            if one.n_dim < self.n_dim:
                warnings.warn('You are combining a 2D obstacle with a 3D environment. The 2D obstacle is transformed to a 3D '
                              'one by extending it infinitely in z dimension.')
            self.obstacles.append(one)
            self.n_obs += 1

    # -- separating hyperplanes ----------------------------------------------------------------------------------------------
    def _hyperplane(self, tag, n_normal, basis):
        """Variables a (n_normal splines) and b of one hyperplane and its row |a|^2 <= 1."""
        a = self.define_spline_variable('a' + tag, n_normal, basis=basis)
        b = self.define_spline_variable('b' + tag, 1, basis=basis)[0]
        length2 = a[0] * a[0]
        for k in range(1, n_normal):
            length2 = length2 + a[k] * a[k]
        self.define_constraint(length2 - 1, -inf, 0.)
        return a, b

    def define_collision_constraints(self, vehicle, splines, horizon_times):
        if vehicle.n_dim != self.n_dim:
            raise ValueError('Not possible to combine %dD vehicle with %dD environment.' % (vehicle.n_dim, self.n_dim))
        spans = horizon_times if isinstance(horizon_times, list) else [horizon_times]
        basis = BSplineBasis(np.r_[0., _breakpoints(vehicle), 1.], 1)      # hyperplanes: piecewise linear on the vehicle's breakpoints
        for seg in range(vehicle.n_seg):
            room = self.room[seg]
            facing = room.get('obstacles', self.obstacles)
            for_vehicle, for_obstacle = {}, {}
            for k, shape in enumerate(vehicle.shapes):
                for_vehicle[shape] = []
                for l, obstacle in enumerate(facing):
                    obstacle.init(horizon_times=spans[:seg + 1])
                    if not obstacle.options['avoid']:
                        continue
                    a, b = self._hyperplane('_%s_seg%d_%d%d' % (vehicle.label, seg, k, l), obstacle.n_dim, basis)
                    if self.n_dim == 3 and obstacle.n_dim == 2:         # a cylinder along z: the normal has no z component
                        normal = [a[0], a[1], BSpline(basis, np.zeros(len(basis)))]
                    else:
                        normal = a
                    for_vehicle[shape].append({'a': normal, 'b': b})
                    planes = for_obstacle.setdefault(obstacle, [])
                    planes.append({'a': a, 'b': b})
                    obstacle.define_collision_constraints(planes)
            vehicle.define_collision_constraints(for_vehicle, room, splines[seg], spans[seg])

    def define_intervehicle_collision_constraints(self, vehicles, horizon_times):
        """One hyperplane per pair of shapes of two different vehicles, seen with opposite signs by the two
        (`environment/environment.py:148-176`)."""
        spans = horizon_times if isinstance(horizon_times, list) else [horizon_times]
        for seg in range(vehicles[0].n_seg):
            seen = dict((veh, dict((shape, []) for shape in veh.shapes)) for veh in vehicles)
            for i, first in enumerate(vehicles):
                for second in vehicles[i + 1:]:
                    if first is second:
                        continue
                    if first.n_dim != second.n_dim:
                        raise ValueError('Not possible to combine %dD and %dD vehicle.' % (first.n_dim, second.n_dim))
                    basis = BSplineBasis(np.r_[0., np.union1d(_breakpoints(first), _breakpoints(second)), 1.], 1)
                    for k1, shape1 in enumerate(first.shapes):
                        for k2, shape2 in enumerate(second.shapes):
                            a, b = self._hyperplane('_%s_seg%d_%d_%s_%d' % (first.label, seg, k1, second.label, k2), self.n_dim, basis)
                            seen[first][shape1].append({'a': a, 'b': b})
                            seen[second][shape2].append({'a': [-component for component in a], 'b': -b})
            for veh in vehicles:
                veh.define_collision_constraints(seen[veh], self.room[seg], veh.splines[seg], spans[seg])

    # -- bookkeeping -----------------------------------------------------------------------------------------------------------
    def init(self, horizon_times=None):
        for obstacle in self.obstacles:
            obstacle.init(horizon_times=horizon_times)

    def simulate(self, simulation_time, sample_time):
        for obstacle in self.obstacles:
            obstacle.simulate(simulation_time, sample_time)
        self.update_plots()

    def draw(self, t=-1):
        return [], []
