"""splat2d(input, coordinates, values, sigma, soft_normalize=False) with the argument checks and
error behaviour of utils/splat2d_cuda/functional.py:31-64 (forward only; backward raises
NotImplementedError; non-GPU input raises NotImplementedError), running csrc/splat2d.hip."""
import torch
import torch.autograd as ag

from .. import _lib

__all__ = ['splat2d']


class Splat2DFunction(ag.Function):
    @staticmethod
    def forward(ctx, input, coordinates, values, sigma, soft_normalize=False):
        assert coordinates.dtype == torch.float32 and values.dtype == torch.float32, \
            'Splat2D only takes float coordinates and values, got {} and {} instead.'.format(coordinates.type(), values.type())
        assert coordinates.size(0) == values.size(0) and coordinates.size(1) == values.size(1), \
            'coordinates should be size (N, num_points, 2) and values should be size (N, num_points, *), got {} and {} instead.'.format(coordinates.shape, values.shape)
        assert input.size(0) == coordinates.size(0) and input.dim() == 4, \
            'input should be of size (N, *, H, W), got {} instead'.format(input.shape)
        assert sigma.size(0) == input.size(0), 'sigma should be a tensor of size (N,)'
        if not coordinates.is_cuda:
            raise NotImplementedError('Splat2D currently only has support for GPU (HIP).')
        input = input.contiguous()
        coordinates = coordinates.contiguous()
        values = values.contiguous()
        sigma = sigma.contiguous().float()
        n, c, h, w = input.shape
        output = torch.empty_like(input)
        if output.numel() == 0:
            return output
        alpha = torch.empty((n, h, w), dtype=torch.float32, device=input.device)
        _lib.call('gg_splat2d_f32', output, alpha, input, coordinates, values, sigma, n, coordinates.size(1), c, h, w,
                  int(bool(soft_normalize)))
        return output

    @staticmethod
    def backward(ctx, grad_output):
        raise NotImplementedError


splat2d = Splat2DFunction.apply
