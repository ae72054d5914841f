"""Autograd wrappers for the fused flow-side kernels of csrc/stn_ops.hip:
affine_grid, the FlowHead composition (convex upsample + identity + affine), bilinear flow
resize and the TV / identity regularisers.  Reference call sites: warping_heads.py:135,173-193,
239-250,268-277; models/losses/loss.py:4-18."""
import torch
from torch.autograd import Function

from .. import _lib


def _f32_cuda(*tensors):
    for t in tensors:
        if t is None:
            continue
        if t.device.type != 'cuda':
            raise _lib.HipLibraryError('flow ops: HIP tensors only (CPU restatement: oracle/np_ops.py)')
        if t.dtype != torch.float32:
            raise TypeError('flow ops: float32 only')


class _AffineGrid(Function):
    @staticmethod
    def forward(ctx, theta, ho, wo):
        _f32_cuda(theta)
        theta = theta.contiguous()
        n = theta.shape[0]
        grid = torch.empty((n, ho, wo, 2), dtype=torch.float32, device=theta.device)
        _lib.call('gg_affine_grid_f32', grid, theta, n, ho, wo)
        ctx.conf = (n, ho, wo)
        return grid

    @staticmethod
    def backward(ctx, grad_grid):
        n, ho, wo = ctx.conf
        gt = torch.empty((n, 2, 3), dtype=torch.float32, device=grad_grid.device)
        _lib.call('gg_affine_grid_bwd_f32', gt, grad_grid.contiguous(), n, ho, wo)
        return gt, None, None


class _SimilarityMatrix(Function):
    """params (N, 4K) -> (N, K, 2, 3), SimilarityHead.make_affine_matrix (warping_heads.py:36-56) as one launch per
    direction (gg_similarity_matrix_f32 / _bwd_f32) instead of 11 + ~25 element-wise launches on (N, K) tensors."""

    @staticmethod
    def forward(ctx, params, heads):
        _f32_cuda(params)
        params = params.contiguous()
        n = params.shape[0]
        if params.dim() != 2 or params.shape[1] != 4 * heads:
            raise ValueError(f'similarity_matrix: params {tuple(params.shape)} for {heads} head(s)')
        m = torch.empty((n, heads, 2, 3), dtype=torch.float32, device=params.device)
        _lib.call('gg_similarity_matrix_f32', m, params, n, heads)
        ctx.save_for_backward(params)
        ctx.heads = heads
        return m

    @staticmethod
    def backward(ctx, gm):
        (params,) = ctx.saved_tensors
        gp = torch.empty_like(params)
        _lib.call('gg_similarity_matrix_bwd_f32', gp, gm.contiguous(), params, params.shape[0], ctx.heads)
        return gp, None


def similarity_matrix(params, heads=1):
    """[rot | log-scale | shift_x | shift_y] (N, 4 * heads) -> similarity matrices (N, heads, 2, 3)."""
    return _SimilarityMatrix.apply(params, int(heads))


def affine_grid(theta, size, align_corners=False):
    """F.affine_grid(theta (N,2,3), size (N,C,H,W), align_corners=False)."""
    if align_corners:
        raise NotImplementedError('affine_grid: align_corners=False only (as the reference uses)')
    return _AffineGrid.apply(theta, int(size[-2]), int(size[-1]))


class _FlowCompose(Function):
    @staticmethod
    def forward(ctx, low_flow, mask, base, ds):
        _f32_cuda(low_flow, mask, base)
        low_flow = low_flow.contiguous()
        mask = mask.contiguous()
        base = base.contiguous() if base is not None else None
        n, two, hl, wl = low_flow.shape
        assert two == 2 and mask.shape == (n, 9 * ds * ds, hl, wl), (low_flow.shape, mask.shape)
        delta = torch.empty((n, ds * hl, ds * wl, 2), dtype=torch.float32, device=low_flow.device)
        flow = torch.empty_like(delta)
        _lib.call('gg_flow_compose_fwd_f32', delta, flow, low_flow, mask, base, n, hl, wl, ds)
        ctx.save_for_backward(low_flow, mask, base if base is not None else low_flow.new_empty(0))
        ctx.conf = (n, hl, wl, ds, base is not None)
        return flow, delta

    @staticmethod
    def backward(ctx, g_flow, g_delta):
        low_flow, mask, base = ctx.saved_tensors
        n, hl, wl, ds, has_base = ctx.conf
        base = base if has_base else None
        glow = torch.empty_like(low_flow)
        gmask = torch.empty_like(mask)
        gbase = torch.empty((n, 2, 3), dtype=torch.float32, device=low_flow.device) if has_base else None
        _lib.call('gg_flow_compose_bwd_f32', glow, gmask, gbase,
                  g_flow.contiguous() if g_flow is not None else None,
                  g_delta.contiguous() if g_delta is not None else None,
                  low_flow, mask, base, n, hl, wl, ds)
        return glow, gmask, gbase, None


def flow_compose(low_flow, mask, base_warp=None, ds=8):
    """low_flow (N,2,h,w) [NCHW conv output], mask (N,9*ds*ds,h,w), base_warp (N,2,3)|None
    -> (flow, delta_flow), both (N,ds*h,ds*w,2)."""
    return _FlowCompose.apply(low_flow, mask, base_warp, ds)


class _FlowResize(Function):
    @staticmethod
    def forward(ctx, flow, scale):
        _f32_cuda(flow)
        flow = flow.contiguous()
        n, hi, wi, _ = flow.shape
        ho, wo = int(hi * scale), int(wi * scale)          # floor(in * scale_factor)
        out = torch.empty((n, ho, wo, 2), dtype=torch.float32, device=flow.device)
        _lib.call('gg_flow_resize_f32', out, flow, n, hi, wi, ho, wo, scale)
        ctx.conf = (n, hi, wi, ho, wo, scale)
        return out

    @staticmethod
    def backward(ctx, g):
        n, hi, wi, ho, wo, scale = ctx.conf
        gi = torch.empty((n, hi, wi, 2), dtype=torch.float32, device=g.device)
        _lib.call('gg_flow_resize_bwd_f32', gi, g.contiguous(), n, hi, wi, ho, wo, scale)
        return gi, None


def flow_resize(flow, scale):
    """F.interpolate(flow.permute(0,3,1,2), scale_factor=scale, mode='bilinear').permute(0,2,3,1)
    (warping_heads.py:250).  scale == 1 is the identity (bit-exact in the reference as well)."""
    if float(scale) == 1.0:
        return flow
    return _FlowResize.apply(flow, float(scale))


class _FlowLosses(Function):
    @staticmethod
    def forward(ctx, delta):
        _f32_cuda(delta)
        delta = delta.contiguous()
        n, hf, wf, two = delta.shape
        assert two == 2
        losses = torch.empty(2, dtype=torch.float32, device=delta.device)
        _lib.call('gg_flow_losses_f32', losses, delta, n, hf, wf)
        ctx.save_for_backward(delta)
        return losses

    @staticmethod
    def backward(ctx, g_losses):
        (delta,) = ctx.saved_tensors
        n, hf, wf, _ = delta.shape
        gd = torch.empty_like(delta)
        _lib.call('gg_flow_losses_bwd_f32', gd, delta, g_losses.contiguous(), n, hf, wf)
        return gd


def flow_losses(delta_flow):
    """-> tensor([total_variation_loss(delta), flow_identity_loss(delta)])  (loss.py:4-18)."""
    return _FlowLosses.apply(delta_flow)
