__all__ = ["Shape", "parse_shape_kwargs", "logger", "load_mesh"]

from .shape import Shape, parse_shape_kwargs
from . import logger
from .mesh import load_mesh
