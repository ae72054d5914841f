"""MipmapWarp / Warp / BilinearDownsample with the class names, constructor arguments, buffers and
forward signatures of models/spatial_transformers/antialiased_sampling.py, running the fused HIP
kernels of csrc/mipmap_warp.hip and csrc/stn_ops.hip.

What changes under the hood (results are the same, see tests/test_gpu_parity.py):
  * no (N, C*D, H, W) Gaussian stack and no `.item()` host sync (antialiased_sampling.py:52): the
    un-upsampled pyramid (3 small launches) is sampled directly, the per-pixel level is computed in
    the sampling kernel;
  * one backward kernel produces the grid gradient, including the term that flows through the
    fractional mip level; the image gradient is produced only when the input requires it.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function

from .. import _lib

_PAD_MODES = {'zeros': 0, 'border': 1, 'reflection': 2}


def _check(t, name):
    if t.device.type != 'cuda':
        raise _lib.HipLibraryError(f'{name}: HIP tensors only (CPU restatement: oracle/np_ops.py)')
    if t.dtype != torch.float32:
        raise TypeError(f'{name}: float32 only, got {t.dtype}')


MAX_LEVELS = 8        # kMaxLevels of csrc/mipmap_warp.hip: max_num_levels <= 8, the reference's default (:22)


def _num_levels(max_level, size):
    """Pyramid levels the sampling can touch: ceil(max_level) + 1, at most down to the 1 x 1 level of a `size` base."""
    return int(min(int(np.ceil(max_level)) + 1, int(np.log2(size)) + 1, MAX_LEVELS))


def _level_views(rest, n, c, hp, wp, levels):
    """Views of the levels 1 .. levels-1 held consecutively in `rest` (level l: (N,C,hp>>l,wp>>l))."""
    views, off = [], 0
    for lvl in range(1, levels):
        h, w = hp >> lvl, wp >> lvl
        views.append(rest[off:off + n * c * h * w].view(n, c, h, w))
        off += n * c * h * w
    return views


def _rest_numel(n, c, hp, wp, levels):
    return sum(n * c * (hp >> lvl) * (wp >> lvl) for lvl in range(1, levels))


def _build_pyramid(base, levels=4):
    """base (N,C,S,S), S a power of two -> (rest, [level 1, ...]): the un-upsampled Gaussian pyramid below the base, all
    levels in ONE buffer (the layout gg_mipmap_warp_* takes) plus per-level views of it."""
    n, c, h, w = base.shape
    if levels <= 1:
        return None, []
    rest = torch.empty(_rest_numel(n, c, h, w, levels), dtype=base.dtype, device=base.device)
    views = _level_views(rest, n, c, h, w, levels)
    src = base
    for lvl, dst in enumerate(views, start=1):
        _lib.call('gg_mip_downsample2x_f32', dst, src, n * c, h >> (lvl - 1), w >> (lvl - 1))
        src = dst
    return rest, views


# Diagnostics (tests/test_gpu_stn_decisions.py): a list of per-call pins for the mip level's arg-max - entry k (an int8
# (N,ho,wo) tensor, or None) is consumed by the k-th anti-aliased warp issued while the list is installed, and its
# backward routes the level's sub-gradient through the pinned neighbours (gg_mipmap_warp_bwd_f32: level_arg_pin).
# None (always, outside that test) costs nothing.
LEVEL_ARG_PINS = None


class _MipmapWarpFn(Function):
    @staticmethod
    def forward(ctx, inputs, grid, max_level, min_level, padding_mode, antialias):
        _check(inputs, 'MipmapWarp')
        _check(grid, 'MipmapWarp')
        inputs = inputs.contiguous()
        grid = grid.contiguous()
        n, c, h, w = inputs.shape
        ho, wo = grid.shape[1], grid.shape[2]
        pad_l = 0
        base = inputs
        rest, nlev = None, 1
        if antialias:
            if h != w:
                raise NotImplementedError('MipmapWarp expects square inputs (as the reference: antialiased_sampling.py:128)')
            if max_level > MAX_LEVELS - 1:
                raise NotImplementedError(f'MipmapWarp: max_num_levels <= {MAX_LEVELS} (the reference default)')
            log_size = np.log2(w)
            if not float(log_size).is_integer():      # reflect-pad to the next power of two (:130-137)
                target = int(2 ** np.ceil(log_size))
                total = target - w
                pad_l = int(total // 2)
                pad_r = int(total - pad_l)
                base = F.pad(inputs, (pad_l, pad_r, pad_l, pad_r), mode='reflect').contiguous()
            nlev = _num_levels(max_level, base.shape[-1])
            rest, _ = _build_pyramid(base, nlev)
        hp, wp = base.shape[-2:]
        out = torch.empty((n, c, ho, wo), dtype=inputs.dtype, device=inputs.device)
        levels = torch.empty((n, ho, wo), dtype=inputs.dtype, device=inputs.device)
        _lib.call('gg_mipmap_warp_fwd_f32', out, levels, base, rest, nlev, grid, n, c, h, w, hp, wp,
                  pad_l, ho, wo, max_level, min_level, _PAD_MODES[padding_mode], int(antialias))
        ctx.arg_pin = None
        if antialias and LEVEL_ARG_PINS:
            ctx.arg_pin = LEVEL_ARG_PINS.pop(0)
            if ctx.arg_pin is not None:
                assert ctx.arg_pin.dtype == torch.int8 and tuple(ctx.arg_pin.shape) == (n, ho, wo), 'level_arg pin shape'
                ctx.arg_pin = ctx.arg_pin.contiguous()
        ctx.save_for_backward(grid, base, *([rest] if rest is not None else []))
        ctx.conf = (n, c, h, w, hp, wp, pad_l, ho, wo, max_level, min_level, _PAD_MODES[padding_mode], int(antialias),
                    nlev)
        ctx.mark_non_differentiable(levels)
        return out, levels

    @staticmethod
    def backward(ctx, grad_out, _grad_levels):
        grid, base, *more = ctx.saved_tensors
        rest = more[0] if more else None
        n, c, h, w, hp, wp, pad_l, ho, wo, max_level, min_level, pm, antialias, nlev = ctx.conf
        grad_out = grad_out.contiguous()
        grad_grid = torch.empty_like(grid)
        want_img = ctx.needs_input_grad[0]
        g0 = grest = None
        if want_img:
            g0 = torch.zeros_like(base)
            grest = torch.zeros_like(rest) if rest is not None else None
        _lib.call('gg_mipmap_warp_bwd_f32', grad_grid, g0, grest, grad_out, base, rest, nlev, grid, n, c, h, w, hp, wp,
                  pad_l, ho, wo, max_level, min_level, pm, antialias, ctx.arg_pin)
        grad_in = None
        if want_img:
            if grest is not None:   # fold the pyramid gradients back: g[l-1] += down2x^T(g[l])
                gp = [g0] + _level_views(grest, n, c, hp, wp, nlev)
                for lvl in range(nlev - 1, 0, -1):
                    _lib.call('gg_mip_downsample2x_bwd_f32', gp[lvl - 1], gp[lvl], n * c, hp >> (lvl - 1), wp >> (lvl - 1))
            grad_in = g0
            if pad_l or hp != h:
                # adjoint of the reflect pad: fold the border gradients back onto the image
                full = grad_in
                pad_r = hp - h - pad_l
                grad_in = _reflect_pad_adjoint(full, pad_l, pad_r)
        return grad_in, grad_grid, None, None, None, None


def _reflect_pad_adjoint(g, pad_l, pad_r):
    """Adjoint of F.pad(x, (l,r,l,r), mode='reflect') via autograd of the pad itself (rare path:
    non power-of-two inputs only)."""
    n, c, hp, wp = g.shape
    h, w = hp - pad_l - pad_r, wp - pad_l - pad_r
    x = torch.zeros((n, c, h, w), dtype=g.dtype, device=g.device, requires_grad=True)
    with torch.enable_grad():
        y = F.pad(x, (pad_l, pad_r, pad_l, pad_r), mode='reflect')
    (gx,) = torch.autograd.grad(y, x, g)
    return gx


def warp_indices(grid, height, width, max_num_levels=3.5, min_level=0.0, padding_mode='border', antialias=True):
    """The integer by-products of MipmapWarp / Warp for `grid` (N,ho,wo,2) sampling an (height, width) image:
    (ix_nw, iy_nw, floor(level), ceil(level)) as int32 (N,ho,wo) tensors, computed by the device functions the
    sampling kernels use (gg_mipmap_warp_indices_f32).  The number of stack levels the reference would build is
    `int(ceil_level.max()) + 1` (antialiased_sampling.py:52)."""
    _check(grid, 'warp_indices')
    grid = grid.contiguous()
    n, ho, wo, _ = grid.shape
    outs = [torch.empty((n, ho, wo), dtype=torch.int32, device=grid.device) for _ in range(4)]
    _lib.call('gg_mipmap_warp_indices_f32', outs[0], outs[1], outs[2], outs[3], None, grid, n, height, width, ho, wo,
              float(max_num_levels - 1.0), float(min_level), _PAD_MODES[padding_mode], int(antialias))
    return tuple(outs)


def warp_level_arg(grid, height, width, max_num_levels=3.5, min_level=0.0):
    """Which neighbour (0 left, 1 right, 2 up, 3 down) holds the maximum coordinate distance at every output pixel - the
    arg-max of MipmapWarp.get_max_coord_distance (antialiased_sampling.py:62-97; first maximum wins), as the sampling
    kernels evaluate it.  int32 (N,ho,wo)."""
    _check(grid, 'warp_level_arg')
    grid = grid.contiguous()
    n, ho, wo, _ = grid.shape
    arg = torch.empty((n, ho, wo), dtype=torch.int32, device=grid.device)
    _lib.call('gg_mipmap_warp_indices_f32', None, None, None, None, arg, grid, n, height, width, ho, wo,
              float(max_num_levels - 1.0), float(min_level), 1, 1)
    return arg


class Warp(nn.Module):
    """Spatial transform without anti-aliasing == F.grid_sample(..., align_corners=False) (:9-16)."""

    def __init__(self):
        super().__init__()

    def forward(self, inputs, grid, padding_mode='border'):
        out, _ = _MipmapWarpFn.apply(inputs, grid, 0.0, 0.0, padding_mode, False)
        return out


class MipmapWarp(nn.Module):
    """Spatial transform with mipmap anti-aliasing; analogous to grid_sample() (:19-60)."""

    def __init__(self, max_num_levels=8):
        super().__init__()
        self.max_num_levels = max_num_levels
        blur = np.array([1., 3., 3., 1.])
        blur = torch.Tensor(blur[:, None] * blur[None, :])
        blur = blur / torch.sum(blur)
        self.register_buffer('blur_filter', blur[None, None, ...])     # kept for state_dict compatibility
        self.levels_map = None
        if max_num_levels - 1.0 > MAX_LEVELS - 1:
            raise NotImplementedError(f'MipmapWarp: the fused kernel keeps up to {MAX_LEVELS} pyramid levels '
                                      f'(max_num_levels <= {MAX_LEVELS}: the reference default, :22; the heads use 3.5, '
                                      f'warping_heads.py:32,170)')

    def forward(self, inputs, grid, min_level=0.0, padding_mode='border'):
        out, levels = _MipmapWarpFn.apply(inputs, grid, float(self.max_num_levels - 1.0), float(min_level),
                                          padding_mode, True)
        self.levels_map = levels / (self.max_num_levels - 1.0)
        return out


class _BilinearDownsampleFn(Function):
    @staticmethod
    def forward(ctx, x, stride):
        _check(x, 'BilinearDownsample')
        x = x.contiguous()
        n, c, h, w = x.shape
        r = stride // 2
        oh = (h + 2 * r - 2 * stride) // stride + 1
        ow = (w + 2 * r - 2 * stride) // stride + 1
        out = torch.empty((n, c, oh, ow), dtype=x.dtype, device=x.device)
        _lib.call('gg_bilinear_downsample_f32', out, x, n * c, h, w, stride)
        ctx.conf = (n, c, h, w, stride)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        n, c, h, w, stride = ctx.conf
        gx = torch.empty((n, c, h, w), dtype=grad_out.dtype, device=grad_out.device)
        _lib.call('gg_bilinear_downsample_bwd_f32', gx, grad_out.contiguous(), n * c, h, w, stride)
        return gx, None


class BilinearDownsample(nn.Module):
    """Tent-filtered strided downsample (:241-256).  Buffers kernel_horz / kernel_vert are kept so
    reference checkpoints load; the kernel itself recomputes the taps."""

    def __init__(self, stride, channels):
        super().__init__()
        self.stride = stride
        self.channels = channels
        kernel = np.arange(1, 2 * stride + 1, 2)
        kernel = np.concatenate((kernel, kernel[::-1]))
        kernel = torch.Tensor(kernel / np.sum(kernel))
        self.register_buffer('kernel_horz', kernel[None, None, None, :].repeat((self.channels, 1, 1, 1)))
        self.register_buffer('kernel_vert', kernel[None, None, :, None].repeat((self.channels, 1, 1, 1)))

    def forward(self, input):
        return _BilinearDownsampleFn.apply(input, self.stride)
