"""Link / Chain / GraphChain data model (reference: numbotics/physics/chain.py).

Hot-path pieces only: the link/joint graph ``_G`` (chain.py:1084-1182), the joint -> q-index map
(chain.py:516-534), joint limits (:539), ``base_pose`` (:972-980) and a plain ``configuration``
slot.  Everything that is a PyBullet accessor upstream (dynamics, contacts, Bullet IK/Jacobian) is out
of scope.
"""
import copy

import numpy as np
import networkx as nx

from numbotics_amd.utils import Shape
from . import world as _world
from .collision import CollisionShape
from .constraint import Constraint, Joint


class Link:

    def __init__(self, offset: np.ndarray, mass: float, collision_shape=None, visual_shape=None, **kwargs):
        if mass < 0:
            raise ValueError("Link mass must be positive")
        self._offset = np.asarray(offset, dtype=np.float64)
        self._mass = mass
        # Upstream keeps ONE collision shape per link (the first <collision>, physics/helpers.py:233).
        # ``_collision_shape`` is that one; ``_collision_shapes`` is every element (compound links).
        shapes = kwargs.pop('collision_shapes', None)
        if collision_shape is None:
            collision_shape = shapes[0] if shapes else CollisionShape(Shape.EMPTY)
        self._collision_shape = collision_shape
        self._collision_shapes = list(shapes) if shapes else (
            [collision_shape] if collision_shape.shape != Shape.EMPTY else [])
        self._visual_shape = visual_shape
        self._body_id = None
        self._index = None
        self._world_name = None
        self._body_name = None
        self._inertia_diagonal = kwargs.pop('inertia_diagonal', None)
        self._name = kwargs.pop('name', None)
        if kwargs:
            raise ValueError(f"Unexpected keyword arguments: {', '.join(kwargs.keys())}")

    def _registered(self):
        return not (self._body_id is None or self._index is None or self._world_name is None)

    def __str__(self):
        if not self._registered():
            raise ValueError("Link is not registered in the world")
        return f"{self._world_name}:entity_id_{self._body_id}:sub_id_{self._index}"

    def __eq__(self, other):
        return str(self) == str(other)

    def __hash__(self):
        return hash(self.name)

    @property
    def name(self):
        """Fully qualified ``world:chain:link`` (chain.py:128-133); ``_name`` is the bare URDF name."""
        if not self._registered():
            raise ValueError("Link is not registered in the world")
        return f'{self._world_name}:{self._body_name}:{self._name}'

    def distance_to(self, target, max_distance: float = np.inf):
        """Closest-point records against ``target`` (an object, a link or a chain) at the bodies' current state, one per shape
        pair, ``distance <= max_distance`` (reference: the ``getClosestPoints`` wrapper of this class)."""
        from .proximity import body_distances
        return body_distances(self, target, max_distance)

    @property
    def index(self):
        return self._index

    @property
    def world(self):
        return _world.get_world(name=self._world_name)


class Chain:

    def __init__(self, pyb_id: int, links, joints, static_base: bool = True, **kwargs):
        world = _world.get_world(name=kwargs.pop('world_name', None))
        self._world_name = world.name
        self._pyb_id = pyb_id
        self._links = list(links)
        self._joints = list(joints)
        self._static_base = static_base if self._links[0]._mass > 0 else True
        self._static = False            # chains live in the dynamic registry upstream
        self._name = kwargs.pop('name', f'chain_{self._pyb_id}')
        if kwargs:
            raise ValueError(f"Unexpected keyword arguments: {', '.join(kwargs.keys())}")
        for i, link in enumerate(self._links):
            link._body_id = self._pyb_id
            link._index = i - 1          # PyBullet convention: base is -1
            link._world_name = self._world_name
            link._body_name = self._name
            if link._name is None:
                link._name = f'link_{link._index}'
        self._links_from_indices = {link._index: link for link in self._links}

        self.__joint_to_index = {}
        j_idx = 0
        for joint in self._joints:
            if joint.type in (Constraint.PRISMATIC, Constraint.REVOLUTE):
                self.__joint_to_index[joint] = j_idx
                j_idx += 1
            elif joint.type == Constraint.SPHERICAL:
                self.__joint_to_index[joint] = [j_idx, j_idx + 1, j_idx + 2]
                j_idx += 3
        self._dof = j_idx
        moving = [j for j in self._joints if j.type != Constraint.FIXED]
        self.joint_damping = np.array([j.damping for j in moving])
        self.joint_limits = np.array([[j.lower_limit, j.upper_limit] for j in moving]).reshape(-1, 2)
        self.joint_effort_limits = np.array([j.max_effort for j in moving])
        self._base_pose = np.eye(4)
        self._configuration = np.zeros((self._dof,))
        world.register(self)

    def __str__(self):
        return f"entity_id_{self._pyb_id}"

    def __eq__(self, other):
        return str(self) == str(other)

    def __hash__(self):
        return hash(self.name)

    @property
    def name(self):
        return f'{self._world_name}:{self._name}'

    @property
    def world(self):
        return _world.get_world(name=self._world_name)

    @property
    def dof(self):
        return self._dof

    @property
    def joint_index(self):
        """joint -> q index (int, or [i,i+1,i+2] for spherical); upstream's private ``__joint_to_index``."""
        return dict(self.__joint_to_index)

    @property
    def base_pose(self):
        return self._base_pose.copy()

    @base_pose.setter
    def base_pose(self, T):
        T = np.asarray(T, dtype=np.float64)
        if T.shape != (4, 4):
            raise ValueError("base_pose must be a 4x4 matrix")
        self._base_pose = T.copy()
        w = _world.WORLD_INSTANCES.get(self._world_name)
        if w is not None:
            w._touch()

    def distance_to(self, target, max_distance: float = np.inf):
        """Every link of this chain (at ``configuration``) against ``target`` (reference: physics/chain.py:944-969)."""
        from .proximity import body_distances
        return body_distances(self, target, max_distance)

    @property
    def configuration(self):
        return self._configuration.copy()

    @configuration.setter
    def configuration(self, q):
        q = np.asarray(q, dtype=np.float64)
        if q.shape != (self._dof,):
            raise ValueError(f"configuration must have {self._dof} elements")
        self._configuration = q.copy()
        w = _world.WORLD_INSTANCES.get(self._world_name)
        if w is not None:
            w._touch()          # another arm's scene holds this chain's links as obstacles at this configuration


class GraphChain(Chain):
    """A tree of links joined by joints, given as an ``nx.DiGraph`` (node attr ``link``, edge attr ``joint``)."""

    def __init__(self, G: nx.DiGraph, static_base: bool = False, **kwargs):
        if not nx.is_tree(G):
            raise ValueError("Chain graph must be a tree")
        if not nx.is_directed_acyclic_graph(G):
            raise ValueError("Chain graph must be a directed acyclic graph")
        G = copy.deepcopy(G)
        order = list(nx.topological_sort(G))
        root = order[0]
        base_link = G.nodes[root]["link"]
        if root != base_link._name:
            raise ValueError(f"Base link name {base_link._name} must match root node name {root}")
        links, joints = [base_link], []
        for node in order[1:]:
            parents = list(G.predecessors(node))
            if len(parents) != 1:
                raise ValueError(f"Node {node} has {len(parents)} parents, expected 1")
            link = G.nodes[node]["link"]
            if link._name != node:
                raise ValueError(f"Link name {link._name} must match node name {node}")
            links.append(link)
            joints.append(G.edges[(parents[0], node)]["joint"])
        self._G = G
        world = _world.get_world(name=kwargs.get('world_name', None))
        super().__init__(world._next_id(), links, joints, static_base, **kwargs)

    @classmethod
    def from_urdf(cls, urdf_path: str):
        from .urdf import _chain_from_urdf
        return _chain_from_urdf(urdf_path)
