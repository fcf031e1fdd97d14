"""MI355X-native batched spline-MPC solve path behind the omg-tools Python API.

`from omgtools import *` exposes the same names the reference's package does for
the in-scope path (reference `omgtools/__init__.py`): vehicles, shapes,
environment, point-to-point / formation problems, simulator and deployer.
"""
from .shapes import (Circle, Polyhedron, RegularPolyhedron, Rectangle, Square, Sphere, Polyhedron3D,
                     Cuboid, Cube, Plate)
from .splines import BSplineBasis, BSpline
from .vehicles import Vehicle, Holonomic, Holonomic3D, Quadrotor, Fleet
from .environment import Environment, Obstacle
from .problems import Problem, Point2point, FixedTPoint2point, FreeTPoint2point
from .execution import Simulator, Deployer
from .formation import FormationPoint2point

__all__ = ['Circle', 'Polyhedron', 'RegularPolyhedron', 'Rectangle', 'Square', 'Sphere', 'Polyhedron3D',
           'Cuboid', 'Cube', 'Plate', 'BSplineBasis', 'BSpline', 'Vehicle', 'Holonomic',
           'Holonomic3D', 'Quadrotor', 'Fleet', 'Environment', 'Obstacle', 'Problem',
           'Point2point', 'FixedTPoint2point', 'FreeTPoint2point', 'FormationPoint2point', 'Simulator', 'Deployer']
