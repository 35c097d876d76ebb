"""Shape vocabulary of the hot path.

Mirrors the names and keyword vocabulary of the reference's ``Shape`` enum
(numbotics/utils/shape.py:17-25) and ``parse_shape_kwargs`` (:118-136).  PyBullet / Meshcat
registration (:27-114) is out of scope: shapes here are plain parameter records that the
device descriptor is packed from.
"""
from enum import Enum, auto


class Shape(Enum):
    CUBE = auto()
    CUBOID = auto()
    SPHERE = auto()
    CYLINDER = auto()
    CAPSULE = auto()
    MESH = auto()
    PLANE = auto()
    EMPTY = auto()


_SHAPE_KWARGS = frozenset({
    'offset', 'half_extents', 'radius', 'height', 'width', 'normal', 'filename', 'color',
    'mesh_scale', 'auto_center', 'convex_decomposition', 'collision_margin',
})


def parse_shape_kwargs(kwargs: dict):
    """Split shape parameters out of ``kwargs`` (reference: utils/shape.py:131-136).

    Returns ``(remaining_kwargs, shape_info)``; ``kwargs`` is consumed in place like upstream.
    """
    shape_info = {}
    for key in list(kwargs.keys()):
        if key in _SHAPE_KWARGS:
            shape_info[key] = kwargs.pop(key)
    return kwargs, shape_info
