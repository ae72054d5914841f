import torch.nn as nn

from .functional import splat2d

__all__ = ['Splat2D', 'splat2d']


class Splat2D(nn.Module):
    """nn.Module face of splat2d.  (The reference's Splat2D.forward passes a mismatched argument list,
    utils/splat2d_cuda/splat.py:12-13, and is never used; this one takes splat2d's arguments.)"""

    def forward(self, input, coordinates, values, sigma, soft_normalize=False):
        return splat2d(input, coordinates, values, sigma, soft_normalize)

    def extra_repr(self):
        return ''
