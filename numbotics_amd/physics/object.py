"""Obstacle bodies (reference: numbotics/physics/object.py:17-63,200-212,353-527).

Data model only: a collision shape plus a pose.  Constructor signatures and keyword handling
(``position=...``, ``name=...``, shape kwargs such as ``offset=...``) follow upstream; dynamics
accessors and wrench application are out of scope.
"""
import numpy as np

from numbotics_amd.utils import Shape, parse_shape_kwargs, logger
from . import world as _world
from .collision import CollisionShape


class PhysicsObject:

    def __init__(self, mass: float, static: bool, collision_shape=None, visual_shape=None, **kwargs):
        if mass < 0:
            raise ValueError("Link mass must be positive")
        self._mass = mass if not static else 0
        self._static = static
        world = _world.get_world(name=kwargs.pop('world_name', None))
        self._collision_shape = collision_shape if collision_shape is not None else CollisionShape(Shape.EMPTY)
        self._visual_shape = visual_shape
        self._world_name = world.name
        self._pyb_id = world._next_id()
        self._pose = np.eye(4)
        self._name = kwargs.pop('name', f'physics_object_{self._pyb_id}')
        for key in list(kwargs.keys()):
            if hasattr(self, key):
                setattr(self, key, kwargs.pop(key))
        if kwargs:
            raise ValueError(f"Unexpected keyword arguments: {', '.join(kwargs.keys())}")
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
    def mass(self):
        return self._mass

    def _moved(self):
        w = _world.WORLD_INSTANCES.get(self._world_name)
        if w is not None:
            w._touch()

    @property
    def pose(self):
        return self._pose.copy()

    @pose.setter
    def pose(self, T):
        T = np.asarray(T, dtype=np.float64)
        if T.shape != (4, 4):
            raise ValueError("pose must be a 4x4 matrix")
        self._pose = T.copy()
        self._moved()

    @property
    def position(self):
        return self._pose[:3, 3].copy()

    @position.setter
    def position(self, p):
        self._pose[:3, 3] = np.asarray(p, dtype=np.float64)
        self._moved()

    def distance_to(self, target, max_distance: float = np.inf):
        """Closest-point records against ``target`` (an object, a link or a chain) at the bodies' current state, one per shape
        pair, ``distance <= max_distance`` (reference: the ``getClosestPoints`` wrapper of this class)."""
        from .proximity import body_distances
        return body_distances(self, target, max_distance)

    @property
    def orientation(self):
        return self._pose[:3, :3].copy()

    @orientation.setter
    def orientation(self, R):
        self._pose[:3, :3] = np.asarray(R, dtype=np.float64)
        self._moved()


def _static_mass_warning(kind, mass, static):
    if mass > 0.0 and static:
        logger.warning(f'{kind} is static, mass will be ignored...')


class Cube(PhysicsObject):
    def __init__(self, mass: float, half_extent: float, static: bool = False, **kwargs):
        _static_mass_warning('Cube', mass, static)
        kwargs, shape_info = parse_shape_kwargs(kwargs)
        he = np.array([half_extent, half_extent, half_extent], dtype=np.float64)
        super().__init__(mass, static, CollisionShape(Shape.CUBE, half_extents=he, **shape_info), None, **kwargs)


class Cuboid(PhysicsObject):
    def __init__(self, mass: float, half_extents: np.ndarray, static: bool = False, **kwargs):
        _static_mass_warning('Cuboid', mass, static)
        kwargs, shape_info = parse_shape_kwargs(kwargs)
        he = np.asarray(half_extents, dtype=np.float64)
        super().__init__(mass, static, CollisionShape(Shape.CUBOID, half_extents=he, **shape_info), None, **kwargs)


class Sphere(PhysicsObject):
    def __init__(self, mass: float, radius: float, static: bool = False, **kwargs):
        _static_mass_warning('Sphere', mass, static)
        kwargs, shape_info = parse_shape_kwargs(kwargs)
        super().__init__(mass, static, CollisionShape(Shape.SPHERE, radius=radius, **shape_info), None, **kwargs)


class Capsule(PhysicsObject):
    """Capsule along local z; ``height`` is the cylindrical part (PyBullet GEOM_CAPSULE convention)."""
    def __init__(self, mass: float, radius: float, height: float, static: bool = False, **kwargs):
        _static_mass_warning('Capsule', mass, static)
        kwargs, shape_info = parse_shape_kwargs(kwargs)
        super().__init__(mass, static, CollisionShape(Shape.CAPSULE, radius=radius, height=height, **shape_info), None, **kwargs)


class Cylinder(PhysicsObject):
    """Cylinder along local z (PyBullet GEOM_CYLINDER convention)."""
    def __init__(self, mass: float, radius: float, height: float, static: bool = False, **kwargs):
        _static_mass_warning('Cylinder', mass, static)
        kwargs, shape_info = parse_shape_kwargs(kwargs)
        super().__init__(mass, static, CollisionShape(Shape.CYLINDER, radius=radius, height=height, **shape_info), None, **kwargs)


class Plane(PhysicsObject):
    """Half-space {x : n.(x - p) <= 0} with p the body position."""
    def __init__(self, mass: float, normal: np.ndarray, static: bool = False, **kwargs):
        _static_mass_warning('Plane', mass, static)
        kwargs, shape_info = parse_shape_kwargs(kwargs)
        n = np.asarray(normal, dtype=np.float64)
        super().__init__(mass, static, CollisionShape(Shape.PLANE, normal=n, **shape_info), None, **kwargs)


class Mesh(PhysicsObject):
    """Mesh obstacle (reference: physics/object.py:425-447).  Shape kwargs as upstream: ``mesh_scale``, ``offset``,
    ``auto_center``, ``convex_decomposition``.  Collision geometry = ONE convex hull of the whole file (trimesh merges the
    file's objects before the reference exports it for Bullet's GEOM_MESH), or with ``convex_decomposition=True`` one hull per
    object of the file (numbotics_amd/utils/mesh.py); the file is read when the scene is compiled."""
    def __init__(self, mass: float, filename: str, static: bool = False, **kwargs):
        self._filename = filename
        _static_mass_warning('Mesh', mass, static)
        kwargs, shape_info = parse_shape_kwargs(kwargs)
        super().__init__(mass, static, CollisionShape(Shape.MESH, filename=filename, **shape_info), None, **kwargs)
