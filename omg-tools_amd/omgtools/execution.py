# This file is derived from OMG-tools (meco-group/omg-tools, `omgtools/execution/simulator.py`, `omgtools/execution/deployer.py` (API surface)).
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
"""`Simulator` / `Deployer`: the host-side drivers of the receding-horizon loop -- the CALLERS of the hot path (SURVEY.md §2
row 21), written for this package against the behaviour of the reference's `execution/simulator.py` (run 39-62, update 92-99,
update_timing 105-111, run_once 113-137, sleep) and `execution/deployer.py` (update 43-79): same class names, constructor
arguments, method names and return values, so that scripts written for OMG-tools drive the HIP path unchanged.

One update of a vehicle's plan is  predict -> solve -> store  on the problem object; `Deployer` decides how long the update
is and which delay the prediction has to bridge, `Simulator` advances a sampled clock in between and asks the problem to
simulate the vehicles.  Where the original package is installed, `omgx_shim` runs its own classes on the HIP path instead.
"""
import numpy as np

from .plotting import PlotLayer


def _samples(span, sample_time):
    """Whole samples in a time span (the reference rounds to 6 decimals before truncating)."""
    return int(np.round(span / sample_time, 6))


def _per_vehicle(problem, pick):
    """`pick(vehicle)` of the only vehicle, or a dict over the vehicles' names."""
    fleet = problem.vehicles
    return pick(fleet[0]) if len(fleet) == 1 else dict((str(v), pick(v)) for v in fleet)


class Deployer(object):
    """Hands measured (or predicted) states to the problem and returns fresh trajectories, one `update` per call."""

    def __init__(self, problem, sample_time=0.01, update_time=0.1):
        self.problem = problem
        self.sample_time, self.update_time = sample_time, update_time
        self.current_time = 0.
        self.iteration0 = True

    def set_problem(self, problem):
        self.problem = problem

    def reset(self):
        self.problem.reinitialize()
        self.iteration0 = True

    def _remaining(self, store):
        """Seconds of `store` ('signals' / 'trajectories' of the first vehicle) beyond the time of the last update."""
        t_end = getattr(self.problem.vehicles[0], store)['time'][0, -1]
        return float(t_end - self.current_time)

    def update(self, current_time, states=None, inputs=None, dinputs=None, update_time=None,
               enforce_states=False, enforce_inputs=False):
        now = float(current_time)
        span = update_time if update_time else self.update_time
        lead = self.problem.vehicles[0]
        # a plan that ends within this update shortens it (signals take precedence over the bare trajectories)
        for store in ('signals', 'trajectories'):
            if hasattr(lead, store):
                left = self._remaining(store)
                if round(span - left, 4) >= self.sample_time:
                    span = left
                break
        if self.iteration0:
            self.problem.initialize(now)
            self.iteration0, delay = False, 0
        else:
            # samples between the end of the previous update and now that the prediction has to skip
            delay = int((now - self.current_time - span) / self.sample_time)
        if hasattr(lead, 'trajectories') and delay + _samples(span, self.sample_time) > _samples(self._remaining('trajectories'), self.sample_time):
            delay = 0
        problem = self.problem
        problem.predict(now, span, self.sample_time, states, inputs, dinputs, delay, enforce_states, enforce_inputs)
        problem.solve(now, span)
        problem.store(now, span, self.sample_time)
        self.current_time = now
        return _per_vehicle(problem, lambda v: v.trajectories)


class Simulator(object):
    """`Deployer` + a sampled clock + the problem's own vehicle / obstacle simulation: `run()` until the stop criterion."""

    def __init__(self, problem, sample_time=0.01, update_time=0.1):
        self.problem = problem
        self.sample_time, self.update_time = sample_time, update_time
        self.deployer = Deployer(problem, sample_time, update_time)
        PlotLayer.simulator = self
        self.reset_timing()

    def set_problem(self, problem):
        self.problem = problem
        self.deployer.set_problem(problem)

    # -- clock ---------------------------------------------------------------------------------------------------------
    def reset_timing(self):
        self.current_time, self.time = 0., np.zeros(1)

    def update_timing(self, update_time=None):
        span = update_time if update_time else self.update_time
        self.current_time += span
        n = _samples(span, self.sample_time)
        last = self.time[-1]
        self.time = np.concatenate((self.time, np.linspace(last + self.sample_time, last + n * self.sample_time, n)))

    def _simulated_ahead(self):
        """How far the first vehicle's stored signals reach beyond the clock."""
        return float(self.problem.vehicles[0].signals['time'][0, -1] - self.current_time)

    # -- driving -------------------------------------------------------------------------------------------------------
    def update(self):
        self.deployer.update(self.current_time)
        self.problem.simulate(self.current_time, self.update_time, self.sample_time)
        return self.problem.stop_criterium(self.current_time, self.update_time)

    def _outcome(self):
        return (_per_vehicle(self.problem, lambda v: v.traj_storage), _per_vehicle(self.problem, lambda v: v.signals))

    def run(self):
        self.deployer.reset()
        while True:
            done = self.update()
            ahead = self._simulated_ahead()
            # (the reference's test, kept with its operator precedence: `stop or update_time - ahead` compared as a number)
            short = (done or self.update_time - ahead) > self.sample_time
            self.update_timing(max(0, ahead - self.sample_time) if short else None)
            if done:
                break
        self.problem.final()
        return self._outcome()

    def step(self, update_time=0.1):
        done = self.update()
        if done:
            self.update_timing(self._simulated_ahead())
            self.problem.final()
        else:
            self.update_timing(update_time)
        traj, signals = self._outcome()
        state = _per_vehicle(self.problem, lambda v: v.signals['state'][:, -1])
        return state, self.current_time, self.problem.options['horizon_time'], done, traj, signals

    def run_once(self, simulate=True, **kwargs):
        """One open-loop solve, executed to its end."""
        self.deployer.reset()
        self.deployer.update(self.current_time, None, update_time=np.inf)
        if not simulate:
            return None
        self.problem.simulate(self.current_time, np.inf, self.sample_time)
        self.problem.final()
        self.update_timing(self._simulated_ahead())
        return _per_vehicle(self.problem, lambda v: v.trajectories)

    def sleep(self, sleep_time):
        self.problem.sleep(self.current_time, sleep_time, self.sample_time)
        self.update_timing(sleep_time)
