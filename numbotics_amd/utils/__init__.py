__all__ = ["Shape", "parse_shape_kwargs", "logger"]

from .shape import Shape, parse_shape_kwargs
from . import logger
