"""Collision records (reference: numbotics/physics/collision.py:25-46)."""
from dataclasses import dataclass

import numpy as np

from numbotics_amd.utils import Shape, parse_shape_kwargs


@dataclass(frozen=True)
class Proximity:
    """One closest-point record; ``distance`` is signed (negative = penetration)."""
    subject: object
    target: object
    position_on_subject: np.ndarray
    position_on_target: np.ndarray
    normal_target_to_subject: np.ndarray
    distance: float


class CollisionShape:
    """A primitive plus its parameters; ``offset`` (4x4) places it in the owner's frame.

    Additive to the reference: ``collision_margin`` (default 0.0) rounds boxes/cylinders the way
    Bullet's margins do; see DESIGN.md "distance semantics".
    """

    def __init__(self, shape: Shape, **kwargs):
        if not isinstance(shape, Shape):
            raise ValueError(f"Invalid shape type: {shape}")
        self.shape = shape
        self._shape_info = parse_shape_kwargs(kwargs)[1]

    @property
    def offset(self) -> np.ndarray:
        return np.asarray(self._shape_info.get('offset', np.eye(4)), dtype=np.float64)
