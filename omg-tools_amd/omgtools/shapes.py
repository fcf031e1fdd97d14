# This file is derived from OMG-tools (meco-group/omg-tools, `omgtools/basics/shape.py`).
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

"""Vehicle/obstacle shapes as *problem data*: checkpoints + radii, canvas
limits, room hyperplanes.  Behavioural spec: reference `basics/shape.py`
(Circle 49-68, Polyhedron 130-171, Rectangle 215-236, Square 239-243,
Sphere 282-336, Polyhedron3D 339-362, Cuboid 402-438, Cube 441-444,
Plate 447-454).  Drawing is out of the hot-path scope.
"""
import numpy as np


def _rot2(theta):
    c, s = np.cos(theta), np.sin(theta)
    return np.array([[c, -s], [s, c]])


def _rot3(orientation):
    roll, pitch, yaw = orientation
    cps, sps = np.cos(yaw), np.sin(yaw)
    cth, sth = np.cos(pitch), np.sin(pitch)
    cph, sph = np.cos(roll), np.sin(roll)
    return np.array([[cth * cps, sph * sth * cps - cph * sps, cph * sth * cps + sph * sps],
                     [cth * sps, sph * sth * sps + cph * cps, cph * sth * sps - sph * cps],
                     [-sth, sph * cth, cph * cth]])


class Shape(object):
    def __init__(self, n_dim):
        self.n_dim = n_dim

    def draw(self, pose=None):
        return [], []


class Shape2D(Shape):
    def __init__(self):
        Shape.__init__(self, 2)

    def rotate(self, orientation, coordinate):
        if isinstance(orientation, np.ndarray):
            orientation = orientation[0]
        return _rot2(orientation).dot(coordinate)


class Circle(Shape2D):
    def __init__(self, radius):
        Shape2D.__init__(self)
        self.radius = radius
        self.n_chck = 1

    def get_checkpoints(self):
        return [[0., 0.]], [self.radius]

    def get_canvas_limits(self):
        return [np.array([-self.radius, self.radius]),
                np.array([-self.radius, self.radius])]


class Polyhedron(Shape2D):
    def __init__(self, vertices, orientation=0., radius=1e-3):
        Shape2D.__init__(self)
        self.n_vert = vertices.shape[1]
        self.orientation = orientation
        self.vertices = self.rotate(orientation, np.asarray(vertices, dtype=float))
        self.radius = radius

    def get_checkpoints(self):
        chck = [[self.vertices[0, l], self.vertices[1, l]] for l in range(self.n_vert)]
        return chck, [self.radius] * self.n_vert

    def get_canvas_limits(self):
        lo, hi = self.vertices.min(axis=1), self.vertices.max(axis=1)
        return [np.array([lo[0], hi[0]]), np.array([lo[1], hi[1]])]

    def get_hyperplanes(self, **kwargs):
        pos = kwargs.get('position', [0, 0])
        v = np.hstack((self.vertices, self.vertices[:, :1]))
        planes = {}
        for k in range(self.n_vert):
            edge = v[:, k + 1] - v[:, k]
            normal = np.array([-edge[1], edge[0]]) / np.hypot(edge[0], edge[1])
            planes[k] = {'a': normal,
                         'b': normal[0] * (v[0, k + 1] + pos[0]) + normal[1] * (v[1, k + 1] + pos[1])}
        return planes


class RegularPolyhedron(Polyhedron):
    """Regular polygon with `n_vert` vertices on a circle of `radius` (`shape.py:191-212`: vertex l is
    the intersection of the edge lines with normals at l*dth and (l+1)*dth from the y axis, i.e. the
    point at angle (l + 1/2) dth)."""

    def __init__(self, radius, n_vert, orientation=0.):
        self.n_vert = n_vert
        dth = 2. * np.pi / n_vert
        ang = (np.arange(n_vert) + 0.5) * dth
        vertices = radius * np.vstack((np.sin(ang), np.cos(ang)))
        Polyhedron.__init__(self, vertices, orientation)
        self.radius_polygon = radius


class Rectangle(Polyhedron):
    def __init__(self, width, height, orientation=0.):
        self.width, self.height = width, height
        w, h = 0.5 * width, 0.5 * height
        # vertex order of the reference construction (`shape.py:221-236`)
        Polyhedron.__init__(self, np.array([[w, w, -w, -w], [h, -h, -h, h]]), orientation)


class Square(Rectangle):
    def __init__(self, side, orientation=0.):
        Rectangle.__init__(self, side, side, orientation)


class Shape3D(Shape):
    def __init__(self):
        Shape.__init__(self, 3)

    def rotate(self, orientation, coordinate):
        if len(orientation) != 3:
            raise ValueError('Orientation is a list with 3 elements: roll, pitch, yaw!')
        return _rot3(orientation).dot(coordinate)


class Sphere(Shape3D):
    def __init__(self, radius):
        Shape3D.__init__(self)
        self.radius = radius
        self.n_chck = 1

    def get_checkpoints(self):
        return [[0., 0., 0.]], [self.radius]

    def get_canvas_limits(self):
        return [np.array([-self.radius, self.radius]) for _ in range(3)]


class Polyhedron3D(Shape3D):
    def __init__(self, vertices, orientation=[0, 0, 0], radius=1e-3):
        Shape3D.__init__(self)
        self.n_vert = vertices.shape[1]
        self.radius = radius
        self.orientation = orientation
        self.vertices = self.rotate(orientation, np.asarray(vertices, dtype=float))

    def get_checkpoints(self):
        chck = [[self.vertices[k, l] for k in range(3)] for l in range(self.n_vert)]
        return chck, [self.radius] * self.n_vert

    def get_canvas_limits(self):
        lo, hi = self.vertices.min(axis=1), self.vertices.max(axis=1)
        return [np.array([lo[k], hi[k]]) for k in range(3)]


class RegularPrisma(Polyhedron3D):
    """Right prism over a regular n-gon (`basics/shape.py:364-399`): `radius` is that of the circle
    through the vertices of the base."""

    def __init__(self, radius, height, n_faces, orientation=[0, 0, 0]):
        self.height, self.n_faces = height, n_faces
        dth = 2 * np.pi / n_faces
        normals = np.array([[np.sin(l * dth), np.cos(l * dth)] for l in range(n_faces)])
        apothem = radius * np.cos(np.pi / n_faces)
        vertices = np.zeros((3, 2 * n_faces))
        for l in range(n_faces):
            # vertex l = intersection of the side faces l and l+1
            a = np.vstack((normals[l], normals[(l + 1) % n_faces]))
            vertices[:2, l] = np.linalg.solve(a, np.array([apothem, apothem]))
            vertices[2, l] = -0.5 * height
            vertices[:2, l + n_faces] = vertices[:2, l]
            vertices[2, l + n_faces] = 0.5 * height
        Polyhedron3D.__init__(self, vertices, orientation)
        self.radius_outer = radius


class Cuboid(Polyhedron3D):
    def __init__(self, width, depth, height, orientation=[0, 0, 0]):
        self.width, self.depth, self.height = width, depth, height
        w, d, h = 0.5 * width, 0.5 * depth, 0.5 * height
        xy = np.array([[w, w, -w, -w], [d, -d, -d, d]])
        vertices = np.vstack((np.hstack((xy, xy)), np.r_[-h * np.ones(4), h * np.ones(4)]))
        Polyhedron3D.__init__(self, vertices, orientation)


class Cube(Cuboid):
    def __init__(self, side, orientation=[0, 0, 0]):
        Cuboid.__init__(self, side, side, side, orientation)


class Plate(Polyhedron3D):
    def __init__(self, shape2d, height, orientation=[0, 0, 0]):
        self.shape2d = shape2d
        vertices = np.vstack((shape2d.vertices, np.zeros((1, shape2d.vertices.shape[1]))))
        Polyhedron3D.__init__(self, vertices, orientation, 0.5 * height)
