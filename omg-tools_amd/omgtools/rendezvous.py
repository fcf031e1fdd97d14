# This file is derived from OMG-tools (meco-group/omg-tools, `omgtools/problems/rendezvous.py`, `admm.py` (API surface)).
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
# Modifications: explicit polynomials instead of CasADi graphs, one shared x-update template for the whole
# fleet, the ADMM iteration as batched device kernels (admm.py).

"""RendezVous ADMM: every vehicle plans to a FREE end point and the fleet agrees on where to meet.

Behavioural spec: reference `problems/rendezvous.py:26-67` (FreeEndPoint2point sub-problems,
`problems/point2point.py:376-418`; coupling constraints `centre_i - centre_j = 0` between neighbours on the
fleet centre `conT + rel_pos_c` of the free end points) on top of `problems/admm.py` (x-update 63-115, z-update
117-168, multiplier update 248-268, residuals 270-307).  The shared quantity is a plain vector of n_dim numbers
(no spline, hence no knot transform anywhere); the z-update projects onto z_i = z_ij for every neighbour.

Same device machinery as the formation problem (formation.py / admm.py): one x-update template for the fleet,
consensus state inside p [B, n_par], `BatchADMM.iterate`.  The kernels see the shared vector as n_dim "splines"
with one coefficient each (layout L = 1).  Pinned against the reference's own graphs by
tests/golden/admm_rendezvous.npz (generator tests/golden/generate_golden_admm.py).
"""
import numpy as np

from .formation import FormationPoint2point
from .consensus import consensus_matrix, RendezVousLayout      # noqa: F401
from .opti import OptiChild, OptiFather
from .problems import FreeEndPoint2point
from .symbolic import Poly


class RendezVousUpdater(OptiChild):
    """Owner of the ADMM parameters of one agent's x-update (`admm.py:63-72`)."""

    def __init__(self):
        OptiChild.__init__(self, 'admm')

    def construct(self, center, n_nghb):
        ns = len(center)
        z_i = np.atleast_1d(self.define_parameter('z_i', ns))
        z_ji = np.atleast_1d(self.define_parameter('z_ji', n_nghb * ns))
        l_i = np.atleast_1d(self.define_parameter('l_i', ns))
        l_ji = np.atleast_1d(self.define_parameter('l_ji', n_nghb * ns))
        rho = self.define_parameter('rho')
        obj = Poly()
        for k in range(ns):
            pairs = [(z_i[k], l_i[k])] + [(z_ji[j * ns + k], l_ji[j * ns + k]) for j in range(n_nghb)]
            for z, l in pairs:
                diff = center[k] - z
                obj = obj + l * diff + 0.5 * rho * diff * diff
        self.define_objective(obj)


def build_rendezvous_template(vehicle, environment, n_nghb, options=None):
    """The x-update NLP of one rendez-vous agent as an `NLPTemplate` (children in the reference's order
    `[vehicle, problem, environment, admm] + obstacles`, `problems/dualmethod.py:52-53`)."""
    import omgtools.backend as be
    opts = {'verbose': 0}
    opts.update(options or {})
    problem = FreeEndPoint2point(vehicle, environment, opts, {vehicle: list(range(vehicle.n_dim))})
    updater = RendezVousUpdater()
    father = OptiFather([vehicle, problem, environment, updater] + environment.obstacles)
    problem.father = father
    with father.table:
        rel_pos_c = np.atleast_1d(vehicle.define_parameter('rel_pos_c', vehicle.n_dim))
        problem.construct()
        # the free end point, requested like the reference does (`rendezvous.py:44`: define_symbol('conT0', ...))
        conT = np.asarray(vehicle.define_symbol('conT0', vehicle.n_dim), dtype=object).reshape(-1)
        center = vehicle.get_fleet_center(list(conT), list(np.asarray(rel_pos_c, dtype=object).reshape(-1)), substitute=True)
        updater.construct(center, n_nghb)
        saved = be.create_nlp
        be.create_nlp = lambda tpl, opt, name='': (None, 0.)   # the batch solver is created by the caller
        try:
            father.construct_problem(opts)
        finally:
            be.create_nlp = saved
    father.init_transformations(problem.init_primal_transform, problem.init_dual_transform)
    return problem, updater, father


class RendezVous(FormationPoint2point):
    """Drop-in for the reference's `RendezVous` (`problems/rendezvous.py:26-89` on top of `ADMMProblem`): same
    constructor and options, the methods `Simulator` / `Deployer` call.  Host bookkeeping per vehicle as in
    `FormationPoint2point` (sub-problems are `FreeEndPoint2point`), one shared x-update template, a dual update
    = one `BatchADMM.iterate` over the fleet."""

    _consensus_is_spline = False

    def __init__(self, fleet, environment, options=None, ops='hip'):
        opts = {'rho': 2.}
        opts.update(options or {})
        FormationPoint2point.__init__(self, fleet, environment, opts, ops=ops)

    def _build_template(self, vehicle, environment, n_nghb, options):
        # the reference frees the end point in the dimensions the fleet configuration names
        # (`rendezvous.py:30-47`: free_ind = configuration[veh].keys()); this path shares one x-update template and one
        # consensus layout of n_dim numbers: a configuration on a subset of the dimensions is refused, not widened
        keys = sorted(self.fleet.configuration[vehicle].keys())
        if keys != list(range(vehicle.n_dim)):
            raise NotImplementedError('RendezVous: the fleet configuration must cover all %d dimensions of the vehicle '
                                      '(got %s)' % (vehicle.n_dim, keys))
        return build_rendezvous_template(vehicle, environment, n_nghb, options)

    def _make_layout(self, tpl, vehicle, problem, updater, n_nghb):
        return RendezVousLayout(tpl, vehicle, problem, updater, n_nghb)

    def stop_criterium(self, current_time, update_time):
        """`rendezvous.py:69-85`: the vehicles have met when their positions differ by the configured offsets
        (summed squared deviation over all neighbour pairs below (5e-2)^2)."""
        res = 0.
        config = self.fleet.configuration
        for veh in self.vehicles:
            ind_veh = sorted(config[veh].keys())
            rel_conT = self.fleet.get_rel_config(veh)
            for nghb in self.fleet.get_neighbors(veh):
                ind_nghb = sorted(config[nghb].keys())
                for k, (ind_v, ind_n) in enumerate(zip(ind_veh, ind_nghb)):
                    rcT = rel_conT[nghb]
                    rcT = rcT if isinstance(rcT, float) else rcT[k]
                    res += np.linalg.norm(veh.trajectories['splines'][ind_v, 0] -
                                          nghb.trajectories['splines'][ind_n, 0] - rcT) ** 2
        return bool(np.sqrt(res) <= 5.e-2)
