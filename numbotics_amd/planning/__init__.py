__all__ = ['unit_bspline']

from .trajectories import unit_bspline
