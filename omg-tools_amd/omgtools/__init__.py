"""MI355X-native batched spline-MPC solve path behind the omg-tools Python API.

`from omgtools import *` exposes the same names the reference's package does for
the in-scope path (reference `omgtools/__init__.py`): vehicles, shapes,
environment, point-to-point / formation problems, simulator and deployer.
"""
import numpy as np          # the reference's package namespace carries numpy as `np` (its examples rely on it)

from .shapes import (Circle, Polyhedron, RegularPolyhedron, Rectangle, Square, Sphere, Polyhedron3D,
                     RegularPrisma, Cuboid, Cube, Plate)
from .splines import BSplineBasis, BSpline
from .vehicles import Vehicle, Holonomic, Holonomic3D, Quadrotor, Fleet
from .environment import Environment, Obstacle
from .problems import Problem, Point2point, FixedTPoint2point, FreeTPoint2point, FreeEndPoint2point
from .execution import Simulator, Deployer
from .formation import FormationPoint2point
from .rendezvous import RendezVous

__all__ = ['np', 'RegularPrisma', 'Circle', 'Polyhedron', 'RegularPolyhedron', 'Rectangle', 'Square', 'Sphere', 'Polyhedron3D',
           'Cuboid', 'Cube', 'Plate', 'BSplineBasis', 'BSpline', 'Vehicle', 'Holonomic',
           'Holonomic3D', 'Quadrotor', 'Fleet', 'Environment', 'Obstacle', 'Problem',
           'Point2point', 'FixedTPoint2point', 'FreeTPoint2point', 'FreeEndPoint2point', 'FormationPoint2point', 'RendezVous', 'Simulator', 'Deployer']
