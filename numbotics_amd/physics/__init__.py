"""Scene / robot description for the hot path (mirror of ``numbotics.physics`` names)."""
__all__ = [
    "CollisionShape", "PhysicsObject", "Cube", "Cuboid", "Sphere", "Mesh", "Plane", "Capsule", "Cylinder",
    "World", "get_world", "Link", "Chain", "GraphChain", "Constraint", "Joint", "Proximity",
]

from .constraint import Constraint, Joint
from .collision import Proximity, CollisionShape
from .world import World, get_world
from .object import PhysicsObject, Cube, Cuboid, Sphere, Mesh, Plane, Capsule, Cylinder
from .chain import Link, Chain, GraphChain
