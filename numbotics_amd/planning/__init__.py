__all__ = ['unit_bspline', 'collision_mask', 'counter_example_bisection']

from .trajectories import unit_bspline
from .safe_sets import collision_mask, counter_example_bisection
