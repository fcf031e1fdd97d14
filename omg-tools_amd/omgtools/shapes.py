# This file is derived from OMG-tools (meco-group/omg-tools, `omgtools/basics/shape.py` (API surface, vertex order)).
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
"""Vehicle / obstacle / room shapes as PROBLEM DATA -- written for this package against the behaviour of the reference's
`basics/shape.py` (Circle 49-68, Polyhedron 130-171, RegularPolyhedron 191-212, Rectangle 215-236, Square 239-243, Sphere
282-336, Polyhedron3D 339-362, RegularPrisma 364-399, Cuboid 402-438, Cube 441-444, Plate 447-454): same class names and
constructor arguments, the same vertex ORDER (checkpoint rows and room rows of the NLP follow it), the same accessor names.

What the optimisation needs from a shape is small: a list of checkpoints with a radius each (a disc / ball is one point with
its radius, a polytope its vertices with a hair of radius), axis-aligned extents, and -- for rooms -- its faces as half
spaces a . q <= b.  Everything here is one base class over a vertex array; drawing is out of scope.
"""
import numpy as np


def _planar_turn(angle):
    c, s = np.cos(angle), np.sin(angle)
    return np.array([[c, -s], [s, c]])


def _spatial_turn(roll, pitch, yaw):
    """R = Rz(yaw) Ry(pitch) Rx(roll)."""
    cr, sr, cp, sp, cy, sy = np.cos(roll), np.sin(roll), np.cos(pitch), np.sin(pitch), np.cos(yaw), np.sin(yaw)
    rz = np.array([[cy, -sy, 0.], [sy, cy, 0.], [0., 0., 1.]])
    ry = np.array([[cp, 0., sp], [0., 1., 0.], [-sp, 0., cp]])
    rx = np.array([[1., 0., 0.], [0., cr, -sr], [0., sr, cr]])
    return rz.dot(ry).dot(rx)


def _ring(radius, n, phase=0.5):
    """n points on a circle, counted clockwise from the y axis: point l at angle (l + phase) 2 pi / n."""
    ang = (np.arange(n) + phase) * (2. * np.pi / n)
    return radius * np.vstack((np.sin(ang), np.cos(ang)))


class Shape(object):
    """A set of points (columns of `vertices`, possibly just the origin) that all carry the same `radius`."""

    n_dim = None

    def __init__(self, n_dim=None):
        if n_dim is not None:
            self.n_dim = n_dim

    def _points(self):
        return getattr(self, 'vertices', np.zeros((self.n_dim, 1)))

    def get_checkpoints(self):
        pts = self._points()
        return [[pts[k, l] for k in range(self.n_dim)] for l in range(pts.shape[1])], [self.radius] * pts.shape[1]

    def get_canvas_limits(self):
        pts = self._points()
        pad = self.radius if not hasattr(self, 'vertices') else 0.
        return [np.array([pts[k].min() - pad, pts[k].max() + pad]) for k in range(self.n_dim)]

    def draw(self, pose=None):
        return [], []


class Shape2D(Shape):
    n_dim = 2

    def __init__(self):
        Shape.__init__(self)

    def rotate(self, orientation, coordinate):
        angle = orientation[0] if isinstance(orientation, np.ndarray) else orientation
        return _planar_turn(angle).dot(coordinate)


class Shape3D(Shape):
    n_dim = 3

    def __init__(self):
        Shape.__init__(self)

    def rotate(self, orientation, coordinate):
        if len(orientation) != 3:
            raise ValueError('Orientation is a list with 3 elements: roll, pitch, yaw!')
        return _spatial_turn(*orientation).dot(coordinate)


class Circle(Shape2D):
    def __init__(self, radius):
        self.radius, self.n_chck = radius, 1


class Sphere(Shape3D):
    def __init__(self, radius):
        self.radius, self.n_chck = radius, 1


class Polyhedron(Shape2D):
    """Convex polygon; `vertices` 2 x n in the order the faces are numbered (face k runs from vertex k to vertex k + 1)."""

    def __init__(self, vertices, orientation=0., radius=1e-3):
        self.orientation, self.radius = orientation, radius
        self.vertices = self.rotate(orientation, np.asarray(vertices, dtype=float))
        self.n_vert = self.vertices.shape[1]

    def get_hyperplanes(self, **kwargs):
        """{k: {'a': outward unit normal of face k, 'b': offset}} of the polygon moved to `position`."""
        shift = np.asarray(kwargs.get('position', [0, 0]), dtype=float)[:2]
        faces = {}
        for k in range(self.n_vert):
            tail, head = self.vertices[:, k], self.vertices[:, (k + 1) % self.n_vert]
            along = head - tail
            normal = np.array([-along[1], along[0]]) / np.hypot(along[0], along[1])
            faces[k] = {'a': normal, 'b': normal[0] * (head[0] + shift[0]) + normal[1] * (head[1] + shift[1])}
        return faces


class RegularPolyhedron(Polyhedron):
    """Regular n-gon whose vertices lie on a circle of `radius` (vertex l midway between the face normals l and l + 1)."""

    def __init__(self, radius, n_vert, orientation=0.):
        Polyhedron.__init__(self, _ring(radius, n_vert), orientation)
        self.radius_polygon = radius


class Rectangle(Polyhedron):
    def __init__(self, width, height, orientation=0.):
        self.width, self.height = width, height
        half = 0.5 * np.array([[width], [height]])
        corners = np.array([[1., 1., -1., -1.], [1., -1., -1., 1.]])      # (+,+), (+,-), (-,-), (-,+): the reference's order
        Polyhedron.__init__(self, half * corners, orientation)


class Square(Rectangle):
    def __init__(self, side, orientation=0.):
        Rectangle.__init__(self, side, side, orientation)


class Polyhedron3D(Shape3D):
    def __init__(self, vertices, orientation=[0, 0, 0], radius=1e-3):
        self.orientation, self.radius = orientation, radius
        self.vertices = self.rotate(orientation, np.asarray(vertices, dtype=float))
        self.n_vert = self.vertices.shape[1]


def _prism(base, z_low, z_high):
    """3 x 2n vertices: the planar polygon `base` (2 x n) at z_low, then at z_high."""
    n = base.shape[1]
    return np.vstack((np.hstack((base, base)), np.r_[z_low * np.ones(n), z_high * np.ones(n)]))


class RegularPrisma(Polyhedron3D):
    """Right prism over a regular n-gon; `radius` is that of the circle through the vertices of the base.  Vertex l of the base
    is where the side faces l and l + 1 meet: with face normals at l * 2 pi / n from the y axis that is the ring point of phase 1/2."""

    def __init__(self, radius, height, n_faces, orientation=[0, 0, 0]):
        self.height, self.n_faces = height, n_faces
        Polyhedron3D.__init__(self, _prism(_ring(radius, n_faces), -0.5 * height, 0.5 * height), orientation)
        self.radius_outer = radius


class Cuboid(Polyhedron3D):
    def __init__(self, width, depth, height, orientation=[0, 0, 0]):
        self.width, self.depth, self.height = width, depth, height
        footprint = Rectangle(width, depth).vertices
        Polyhedron3D.__init__(self, _prism(footprint, -0.5 * height, 0.5 * height), orientation)


class Cube(Cuboid):
    def __init__(self, side, orientation=[0, 0, 0]):
        Cuboid.__init__(self, side, side, side, orientation)


class Plate(Polyhedron3D):
    """A planar shape lying in z = 0, half its thickness as the radius of every vertex."""

    def __init__(self, shape2d, height, orientation=[0, 0, 0]):
        self.shape2d = shape2d
        flat = np.vstack((shape2d.vertices, np.zeros((1, shape2d.vertices.shape[1]))))
        Polyhedron3D.__init__(self, flat, orientation, 0.5 * height)
