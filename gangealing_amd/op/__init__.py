"""Drop-in for the reference package models/stylegan2/op (op/__init__.py:1-2)."""
from .fused_act import FusedLeakyReLU, fused_leaky_relu
from .upfirdn2d import upfirdn2d
from . import conv2d_gradfix

__all__ = ['FusedLeakyReLU', 'fused_leaky_relu', 'upfirdn2d', 'conv2d_gradfix']
