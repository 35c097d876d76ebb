"""URDF -> GraphChain (reference semantics: numbotics/physics/helpers.py:176-356).

Own minimal reader on ``xml.etree`` (upstream uses urdf_parser_py 0.0.4, absent here).  Same
mapping: <box> -> CUBOID(half_extents=size/2), <sphere> -> SPHERE, <cylinder> -> CYLINDER(radius,
height=length), <mesh> -> MESH; 'continuous' -> REVOLUTE; joint offset = trans_mat(xyz, rpy_xyz);
limits/velocity/effort copied when present.  Deliberate differences: every
<collision> element of a link is read (upstream: the first only, which stays available as
``Link._collision_shape``), and <dynamics damping> is parsed without upstream's NameError.
"""
import pathlib
import xml.etree.ElementTree as ET

import numpy as np
import networkx as nx

from numbotics_amd.math import rpy_matrix, trans_mat
from numbotics_amd.utils import Shape
from .chain import GraphChain, Link
from .collision import CollisionShape
from .constraint import Constraint, Joint


def _floats(text, n=None):
    vals = [float(v) for v in text.split()]
    if n is not None and len(vals) != n:
        raise ValueError(f"expected {n} numbers, got '{text}'")
    return np.array(vals, dtype=np.float64)


def _origin(elem) -> np.ndarray:
    if elem is None:
        return np.eye(4)
    origin = elem.find('origin')
    if origin is None:
        return np.eye(4)
    xyz = _floats(origin.get('xyz', '0 0 0'), 3)
    rpy = _floats(origin.get('rpy', '0 0 0'), 3)
    return trans_mat(pos=xyz, orn=rpy_matrix(rpy))


def _collision_shape(coll, base_dir) -> CollisionShape:
    offset = _origin(coll)
    geom = coll.find('geometry')
    if geom is None or len(geom) == 0:
        raise ValueError("<collision> without <geometry>")
    g = geom[0]
    if g.tag == 'box':
        return CollisionShape(Shape.CUBOID, offset=offset, half_extents=_floats(g.get('size'), 3) / 2.0)
    if g.tag == 'sphere':
        return CollisionShape(Shape.SPHERE, offset=offset, radius=float(g.get('radius')))
    if g.tag == 'cylinder':
        return CollisionShape(Shape.CYLINDER, offset=offset, radius=float(g.get('radius')),
                              height=float(g.get('length')))
    if g.tag == 'capsule':     # PyBullet URDF extension; not in urdf_parser_py (additive)
        return CollisionShape(Shape.CAPSULE, offset=offset, radius=float(g.get('radius')),
                              height=float(g.get('length')))
    if g.tag == 'mesh':
        scale = _floats(g.get('scale'), 3) if g.get('scale') else np.ones((3,))
        return CollisionShape(Shape.MESH, offset=offset, filename=str(base_dir / g.get('filename')),
                              mesh_scale=scale)
    raise ValueError(f"Unknown URDF geometry <{g.tag}>")


def _graph_from_urdf(urdf_path: str) -> nx.DiGraph:
    base_dir = pathlib.Path(urdf_path).parent
    root = ET.parse(urdf_path).getroot()
    if root.tag != 'robot':
        raise ValueError("URDF root element must be <robot>")
    G = nx.DiGraph()
    for le in root.findall('link'):
        inertial = le.find('inertial')
        mass = 0.0
        if inertial is not None and inertial.find('mass') is not None:
            mass = float(inertial.find('mass').get('value', 0.0))
        shapes = [_collision_shape(c, base_dir) for c in le.findall('collision')]
        link = Link(offset=_origin(inertial), mass=mass, collision_shapes=shapes, name=le.get('name'))
        G.add_node(le.get('name'), link=link)
    for je in root.findall('joint'):
        jtype = je.get('type')
        if jtype == 'continuous':
            ctype = Constraint.REVOLUTE
        elif jtype in ('revolute', 'prismatic', 'fixed', 'spherical'):
            ctype = Constraint[jtype.upper()]
        else:
            raise ValueError(f"Unsupported URDF joint type '{jtype}'")
        axis_e = je.find('axis')
        if axis_e is not None:
            axis = _floats(axis_e.get('xyz', '1 0 0'), 3)
        else:
            axis = np.zeros(3) if jtype == 'fixed' else np.array([1.0, 0.0, 0.0])   # URDF default axis
        args = {}
        dyn = je.find('dynamics')
        if dyn is not None and dyn.get('damping') is not None:
            args['damping'] = float(dyn.get('damping'))
        lim = je.find('limit')
        if lim is not None:
            for attr, key in (('lower', 'lower_limit'), ('upper', 'upper_limit'),
                              ('velocity', 'max_velocity'), ('effort', 'max_effort')):
                if lim.get(attr) is not None:
                    args[key] = float(lim.get(attr))
        joint = Joint(name=je.get('name'), offset=_origin(je), axis=axis, type=ctype, **args)
        parent, child = je.find('parent').get('link'), je.find('child').get('link')
        for n in (parent, child):
            if n not in G.nodes:
                raise ValueError(f"joint '{je.get('name')}' references unknown link '{n}'")
        G.add_edge(parent, child, joint=joint)
    return G


def _chain_from_urdf(urdf_path: str) -> GraphChain:
    return GraphChain(_graph_from_urdf(urdf_path))
