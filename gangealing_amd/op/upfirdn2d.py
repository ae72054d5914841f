"""upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)) - same signature and semantics as
models/stylegan2/op/upfirdn2d.py:147-158, running gg_upfirdn2d_f32/f64 (csrc/upfirdn2d.hip).

The gradient of upfirdn2d is upfirdn2d with up<->down, flipped taps and the g_pad of
upfirdn2d.py:113-118, so a single autograd Function whose backward re-applies itself gives
first- and second-order gradients (the reference needs two Function classes, :21-144).
No gradient flows to `kernel` (as in the reference).
"""
import torch
from torch.autograd import Function

from .. import _lib

_SUFFIX = {torch.float32: 'f32', torch.float64: 'f64'}


def _out_size(in_h, in_w, kh, kw, up_x, up_y, down_x, down_y, px0, px1, py0, py1):
    out_h = (in_h * up_y + py0 + py1 - kh) // down_y + 1
    out_w = (in_w * up_x + px0 + px1 - kw) // down_x + 1
    return out_h, out_w


def _launch(x, kernel, up_x, up_y, down_x, down_y, px0, px1, py0, py1):
    if x.dtype not in _SUFFIX:
        raise TypeError(f'upfirdn2d: unsupported dtype {x.dtype} (float32 / float64)')
    n, c, in_h, in_w = x.shape
    kh, kw = kernel.shape
    out_h, out_w = _out_size(in_h, in_w, kh, kw, up_x, up_y, down_x, down_y, px0, px1, py0, py1)
    x = x.contiguous()
    kernel = kernel.to(dtype=x.dtype).contiguous()
    out = torch.empty((n, c, max(out_h, 0), max(out_w, 0)), dtype=x.dtype, device=x.device)
    if out.numel():
        _lib.call('gg_upfirdn2d_' + _SUFFIX[x.dtype], out, x, kernel, n * c, in_h, in_w, kh, kw,
                  up_x, up_y, down_x, down_y, px0, px1, py0, py1)
    return out


class UpFirDn2d(Function):
    @staticmethod
    def forward(ctx, input, kernel, up, down, pad):
        up_x, up_y = up
        down_x, down_y = down
        px0, px1, py0, py1 = pad
        kh, kw = kernel.shape
        _, _, in_h, in_w = input.shape
        out_h, out_w = _out_size(in_h, in_w, kh, kw, up_x, up_y, down_x, down_y, px0, px1, py0, py1)
        ctx.save_for_backward(kernel)
        ctx.conf = (up, down)
        # padding of the adjoint (reference upfirdn2d.py:113-118)
        ctx.g_pad = (kw - px0 - 1, in_w * up_x - out_w * down_x + px0 - up_x + 1,
                     kh - py0 - 1, in_h * up_y - out_h * down_y + py0 - up_y + 1)
        return _launch(input, kernel, up_x, up_y, down_x, down_y, px0, px1, py0, py1)

    @staticmethod
    def backward(ctx, grad_output):
        (kernel,) = ctx.saved_tensors
        up, down = ctx.conf
        grad_input = UpFirDn2d.apply(grad_output, torch.flip(kernel, [0, 1]), down, up, ctx.g_pad)
        return grad_input, None, None, None, None


def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):
    if input.device.type != 'cuda':
        raise _lib.HipLibraryError('upfirdn2d: HIP tensors only (the CPU restatement is oracle/np_ops.upfirdn2d)')
    return UpFirDn2d.apply(input, kernel, (up, up), (down, down), (pad[0], pad[1], pad[0], pad[1]))
