"""Drop-in for utils/splat2d_cuda (__init__.py:1 -> splat.py)."""
from .functional import splat2d
from .splat import Splat2D

__all__ = ['Splat2D', 'splat2d']
