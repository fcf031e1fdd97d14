# This file is derived from OMG-tools (meco-group/omg-tools, `omgtools/execution/simulator.py`, `deployer.py`).
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

"""`Simulator` / `Deployer`: the receding-horizon driver loop (host side).

Behavioural spec: reference `execution/simulator.py` (run 39-62, update 92-99,
update_timing 105-111, run_once 113-137) and `execution/deployer.py`
(update 43-79).  These are the *callers* of the hot path.
"""
import numpy as np

from .plotting import PlotLayer


class Deployer(object):

    def __init__(self, problem, sample_time=0.01, update_time=0.1):
        self.set_problem(problem)
        self.update_time = update_time
        self.sample_time = sample_time
        self.current_time = 0.
        self.iteration0 = True

    def set_problem(self, problem):
        self.problem = problem

    def reset(self):
        self.iteration0 = True
        self.problem.reinitialize()

    def _time_left(self, key):
        veh = self.problem.vehicles[0]
        return float(getattr(veh, key)['time'][0, -1] - self.current_time)

    def update(self, current_time, states=None, inputs=None, dinputs=None, update_time=None,
               enforce_states=False, enforce_inputs=False):
        current_time = float(current_time)
        if not update_time:
            update_time = self.update_time
        veh = self.problem.vehicles[0]
        # shorten the update when less than update_time of trajectory is left
        if hasattr(veh, 'signals'):
            if round(update_time - self._time_left('signals'), 4) >= self.sample_time:
                update_time = self._time_left('signals')
        elif hasattr(veh, 'trajectories'):
            if round(update_time - self._time_left('trajectories'), 4) >= self.sample_time:
                update_time = self._time_left('trajectories')
        if self.iteration0:
            self.iteration0 = False
            self.problem.initialize(current_time)
            delay = 0
        else:
            delay = int((current_time - self.current_time - update_time) / self.sample_time)
        if hasattr(veh, 'trajectories'):
            if (delay + int(np.round(update_time / self.sample_time, 6))) > \
                    int(np.round(self._time_left('trajectories') / self.sample_time, 6)):
                delay = 0
        self.problem.predict(current_time, update_time, self.sample_time, states, inputs,
                             dinputs, delay, enforce_states, enforce_inputs)
        self.problem.solve(current_time, update_time)
        self.problem.store(current_time, update_time, self.sample_time)
        self.current_time = current_time
        if len(self.problem.vehicles) == 1:
            return self.problem.vehicles[0].trajectories
        return {str(v): v.trajectories for v in self.problem.vehicles}


class Simulator(object):

    def __init__(self, problem, sample_time=0.01, update_time=0.1):
        self.deployer = Deployer(problem, sample_time, update_time)
        self.update_time = update_time
        self.sample_time = sample_time
        self.problem = problem
        PlotLayer.simulator = self
        self.reset_timing()

    def set_problem(self, problem):
        self.deployer.set_problem(problem)
        self.problem = problem

    def _results(self):
        if len(self.problem.vehicles) == 1:
            veh = self.problem.vehicles[0]
            return veh.traj_storage, veh.signals
        return ({str(v): v.traj_storage for v in self.problem.vehicles},
                {str(v): v.signals for v in self.problem.vehicles})

    def run(self):
        self.deployer.reset()
        stop = False
        while not stop:
            stop = self.update()
            simulated = float(self.problem.vehicles[0].signals['time'][0, -1] - self.current_time)
            if (stop or self.update_time - simulated) > self.sample_time:
                self.update_timing(max(0, simulated - self.sample_time))
            else:
                self.update_timing()
        self.problem.final()
        return self._results()

    def step(self, update_time=0.1):
        stop = self.update()
        if stop:
            self.update_timing(float(self.problem.vehicles[0].signals['time'][0, -1] -
                                     self.current_time))
            self.problem.final()
        else:
            self.update_timing(update_time)
        motion_time = self.problem.options['horizon_time']
        traj, signals = self._results()
        if len(self.problem.vehicles) == 1:
            state = self.problem.vehicles[0].signals['state'][:, -1]
        else:
            state = {str(v): v.signals['state'][:, -1] for v in self.problem.vehicles}
        return state, self.current_time, motion_time, stop, traj, signals

    def update(self):
        self.deployer.update(self.current_time)
        self.problem.simulate(self.current_time, self.update_time, self.sample_time)
        return self.problem.stop_criterium(self.current_time, self.update_time)

    def reset_timing(self):
        self.current_time = 0.
        self.time = np.r_[0.]

    def update_timing(self, update_time=None):
        update_time = self.update_time if not update_time else update_time
        self.current_time += update_time
        n_samp = int(np.round(update_time / self.sample_time, 6))
        self.time = np.r_[self.time, np.linspace(self.time[-1] + self.sample_time,
                                                 self.time[-1] + n_samp * self.sample_time,
                                                 n_samp)]

    def run_once(self, simulate=True, **kwargs):
        self.deployer.reset()
        self.deployer.update(self.current_time, None, update_time=np.inf)
        if not simulate:
            return None
        self.problem.simulate(self.current_time, np.inf, self.sample_time)
        self.problem.final()
        self.update_timing(float(self.problem.vehicles[0].signals['time'][0, -1] -
                                 self.current_time))
        if len(self.problem.vehicles) == 1:
            return self.problem.vehicles[0].trajectories
        return {str(v): v.trajectories for v in self.problem.vehicles}

    def sleep(self, sleep_time):
        self.problem.sleep(self.current_time, sleep_time, self.sample_time)
        self.update_timing(sleep_time)
