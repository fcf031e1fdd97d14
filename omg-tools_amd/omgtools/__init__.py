"""MI355X-native batched spline-MPC solve path behind the omg-tools Python API.

`from omgtools import *` exposes the same names the reference's package does for
the in-scope path (reference `omgtools/__init__.py`): vehicles, shapes,
environment, point-to-point / formation problems, simulator and deployer.

The names are resolved on first use (PEP 562): `import omgtools.backend` / `omgtools.batch` / `omgtools.workloads` -- the
benchmark path -- does not import the front-end modules (vehicles, environment, problems, shapes, execution).
"""
import importlib

import numpy as np          # the reference's package namespace carries numpy as `np` (its examples rely on it)

_WHERE = {
    'shapes': ('Circle', 'Polyhedron', 'RegularPolyhedron', 'Rectangle', 'Square', 'Sphere', 'Polyhedron3D', 'RegularPrisma',
               'Cuboid', 'Cube', 'Plate'),
    'splines': ('BSplineBasis', 'BSpline'),
    'vehicles': ('Vehicle', 'Holonomic', 'Holonomic3D', 'Quadrotor', 'Fleet'),
    'environment': ('Environment', 'Obstacle'),
    'problems': ('Problem', 'Point2point', 'FixedTPoint2point', 'FreeTPoint2point', 'FreeEndPoint2point'),
    'execution': ('Simulator', 'Deployer'),
    'formation': ('FormationPoint2point',),
    'rendezvous': ('RendezVous',),
}
_MODULE_OF = dict((name, mod) for mod, names in _WHERE.items() for name in names)

__all__ = ['np'] + sorted(_MODULE_OF)


def __getattr__(name):
    mod = _MODULE_OF.get(name)
    if mod is None:
        raise AttributeError('module %r has no attribute %r' % (__name__, name))
    value = getattr(importlib.import_module('.' + mod, __name__), name)
    globals()[name] = value
    return value


def __dir__():
    return sorted(list(globals()) + list(_MODULE_OF))
