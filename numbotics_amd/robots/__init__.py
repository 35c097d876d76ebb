__all__ = ["Arm", "Robot"]

from .robot import Robot
from .arm import Arm
