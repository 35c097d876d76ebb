"""Joint vocabulary (reference: numbotics/physics/constraint.py:11-59).

The enum values are PyBullet's ``JOINT_*`` constants, which the reference's FK kernels
compare against as integers (robots/helpers.py:37,43,57,67).
"""
import enum
from dataclasses import dataclass
from typing import Optional

import numpy as np


class Constraint(enum.Enum):
    REVOLUTE = 0
    PRISMATIC = 1
    SPHERICAL = 2
    FIXED = 4


@dataclass(frozen=True, eq=False)
class Joint:
    offset: np.ndarray
    axis: np.ndarray
    type: Constraint
    name: Optional[str] = None
    parent_pose: Optional[np.ndarray] = None
    child_pose: Optional[np.ndarray] = None
    damping: float = 0.01
    lower_limit: float = -np.inf
    upper_limit: float = np.inf
    max_velocity: float = np.inf
    max_effort: float = np.inf

    def __post_init__(self):
        object.__setattr__(self, 'offset', np.asarray(self.offset, dtype=np.float64))
        if self.type == Constraint.FIXED:
            object.__setattr__(self, 'axis', np.zeros((3,), dtype=np.float64))
        else:
            object.__setattr__(self, 'axis', np.asarray(self.axis, dtype=np.float64))

    def __hash__(self):
        return id(self) if self.name is None else hash(self.name)

    def __eq__(self, other):
        if not isinstance(other, Joint):
            return NotImplemented
        if self.name is None or other.name is None:
            return self is other
        return self.name == other.name

    @property
    def dof(self) -> int:
        if self.type == Constraint.FIXED:
            return 0
        if self.type in (Constraint.REVOLUTE, Constraint.PRISMATIC):
            return 1
        if self.type == Constraint.SPHERICAL:
            return 3
        raise ValueError(f'Invalid constraint type: {self.type}')
