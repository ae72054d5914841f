"""SimilarityHead / FlowHead: regress a warp from STN features and apply it with the fused HIP
sampling kernels.  Constructor arguments, parameter names (``linear``, ``flow_out``, ``mask_out``),
forward arguments and return tuples follow models/spatial_transformers/warping_heads.py:14-265.

Execution differs: affine_grid, the RAFT convex upsampling + identity + affine composition and the
anti-aliased warp are one kernel each (csrc/stn_ops.hip, csrc/mipmap_warp.hip) instead of ~40
elementwise / unfold / softmax / grid_sample launches; ``identity_flow`` is generated inside the
kernel, so nothing is pinned to a device at construction time (the reference calls .cuda() in
__init__, warping_heads.py:158).
"""
import math

import torch
import torch.nn as nn

from .antialiased_sampling import MipmapWarp, Warp
from .flow_ops import affine_grid, flow_compose, flow_resize, similarity_matrix
from ..op import conv_mfma
from ..stylegan2.networks import EqualConv2d


def _resolve_policy(warp_policy, img, num_heads):
    """-> (policy name, per-sample head index or None).  'cartesian': every image x every head;
    tensor / module scores: each image uses argmax(score) % num_heads (warping_heads.py:101-115)."""
    logits = None
    if isinstance(warp_policy, torch.Tensor):
        logits = warp_policy
    elif isinstance(warp_policy, nn.Module):
        logits = warp_policy(img)
    elif warp_policy != 'cartesian':
        raise NotImplementedError(warp_policy)
    if logits is None:
        return 'cartesian', None
    return 'assign_only', logits.max(dim=1).indices % num_heads


def check_if_warp_exceeds_image_boundaries(grid, image_bounds, img_size, split_size, threshold=0.025):
    """(N,) bool: did more than `threshold` of the output pixels sample beyond the image (or beyond the un-padded
    content described by `image_bounds` (N, 2) = (height, width) of the raw image)?  warping_heads.py:280-310; used by
    the data pre-processing application only."""
    if image_bounds is None:
        boundary_y = img_size[-2]
        boundary_x = img_size[-1]
    else:
        image_bounds = image_bounds.repeat_interleave(split_size, dim=0)
        landscape = image_bounds[:, 0] < image_bounds[:, 1]
        boundary_y = torch.where(landscape, img_size[-2] * image_bounds[:, 0] / image_bounds[:, 1],
                                 torch.tensor(img_size[-2], dtype=torch.float, device=grid.device)).round()
        boundary_x = torch.where(landscape, torch.tensor(img_size[-1], dtype=torch.float, device=grid.device),
                                 img_size[-1] * image_bounds[:, 1] / image_bounds[:, 0]).round()
    grid_x, grid_y = grid[..., 0], grid[..., 1]
    bx, by = (boundary_x - 1) / img_size[-1], (boundary_y - 1) / img_size[-2]
    if isinstance(bx, torch.Tensor):
        bx, by = bx.view(-1, 1), by.view(-1, 1)
    oob_x = grid_x.flatten(1).abs().gt(bx).float().mean(dim=1).gt(threshold)
    oob_y = grid_y.flatten(1).abs().gt(by).float().mean(dim=1).gt(threshold)
    return torch.logical_or(oob_y, oob_x)


class SimilarityHead(nn.Module):
    """Rotation, uniform scale and translation (4 parameters per head)."""

    def __init__(self, in_shape, antialias=True, num_heads=1, **kwargs):
        super().__init__()
        self.num_warp_params = 4
        self.linear = nn.Linear(in_shape, self.num_warp_params * num_heads, bias=True)
        self.linear.bias.data.zero_()          # identity warp at initialisation
        self.linear.weight.data.zero_()
        self.warper = MipmapWarp(max_num_levels=3.5) if antialias else Warp()
        self.num_heads = num_heads
        self.register_buffer('one_hot', torch.tensor([0, 0, 1], dtype=torch.float).view(1, 1, 1, 3))

    @staticmethod
    def make_affine_matrix(rot, scale, shift_x, shift_y):
        n, k = rot.size()
        rot = torch.tanh(rot) * math.pi
        scale = torch.exp(scale)
        c, s = torch.cos(rot), torch.sin(rot)
        rows = torch.stack([scale * c, -scale * s, shift_x, scale * s, scale * c, shift_y], dim=2)
        return rows.reshape(n, k, 2, 3)

    def make_3x3(self, m):
        return torch.cat([m, self.one_hot.expand(m.size(0), m.size(1), 1, 3)], 2)

    def forward(self, img, features, output_resolution=None, alpha=None, base_warp=None, stop_grad=False,
                padding_mode='border', return_out_of_bounds=False, image_bounds=None, warp_policy='cartesian',
                unfold=False):
        n = features.size(0)
        params = self.linear(features)
        policy, assignments = _resolve_policy(warp_policy, img, self.num_heads)
        if policy == 'assign_only':
            params = params.reshape(-1, self.num_warp_params, self.num_heads).permute(0, 2, 1)
            params = params.gather(1, assignments.view(n, 1, 1).repeat(1, 1, self.num_warp_params)).squeeze(1)
            split = 1
        else:
            split = self.num_heads
        if params.dtype == torch.float32 and params.is_cuda and 'similarity_matrix' not in conv_mfma.DISABLED:
            matrix = similarity_matrix(params, split)                                  # (N, split, 2, 3), one launch
        else:
            matrix = self.make_affine_matrix(*torch.split(params, split, dim=1))
        if base_warp is not None:
            if base_warp.dim() == 3:
                base_warp = base_warp.unsqueeze(1)
            matrix = base_warp @ self.make_3x3(matrix)
        if alpha is not None:
            eye = torch.eye(2, 3, device=matrix.device)[None, None]
            matrix = eye.lerp(matrix, alpha[:, None, None, None])
        res = img.size(-1) if output_resolution is None else output_resolution
        res_h = img.size(-2) if output_resolution is None else output_resolution
        if stop_grad:
            matrix = matrix.detach() + 0 * matrix
        matrix = matrix.reshape(n * split, 2, 3)
        img = img.repeat_interleave(split, dim=0) if split > 1 else img
        grid = affine_grid(matrix, (n * split, img.size(1), res_h, res))
        out = self.warper(img, grid, padding_mode=padding_mode)
        oob = None
        if return_out_of_bounds:
            oob = check_if_warp_exceeds_image_boundaries(grid, image_bounds, (n * split, img.size(1), res_h, res), split)
        if unfold:
            out = out.reshape(n, -1, out.size(1), out.size(2), out.size(3))
            matrix = matrix.reshape(n, -1, 2, 3)
            grid = grid.reshape(n, -1, res_h, res, 2)
        return out, grid, matrix, oob


class FlowHead(nn.Module):
    """Dense flow: low-resolution flow + RAFT convex upsampling mask, composed with the previous
    stage's similarity transform and applied by anti-aliased reverse sampling."""

    def __init__(self, in_shape, antialias=True, num_heads=1, flow_downsample=8, **kwargs):
        super().__init__()
        self.flow_downsample = flow_downsample
        ch = in_shape[1]
        self.flow_out = nn.Sequential(EqualConv2d(ch, ch, 3, padding=1), nn.ReLU(),
                                      EqualConv2d(ch, num_heads * 2, 3, padding=1))
        nn.init.zeros_(self.flow_out[-1].weight)   # identity warp at initialisation
        nn.init.zeros_(self.flow_out[-1].bias)
        self.mask_out = nn.Sequential(EqualConv2d(ch, ch, 3, padding=1), nn.ReLU(),
                                      EqualConv2d(ch, num_heads * 9 * flow_downsample * flow_downsample, 3, padding=1))
        self.warper = MipmapWarp(max_num_levels=3.5) if antialias else Warp()
        self.num_heads = num_heads
        self.flow_res = (flow_downsample * in_shape[2], flow_downsample * in_shape[3])

    @staticmethod
    def _head(seq, features):
        """conv3x3 -> ReLU -> conv3x3 (warping_heads.py:160-169).  The first convolution carries its bias + ReLU in the
        epilogue and the ReLU's backward + bias gradient in one pass (conv_mfma.conv3x3_bias_act with slope 0, gain 1) -
        unless a diagnostics observer is installed (conv_mfma.ACT_OBSERVER: the decision-replay test pins the nn.ReLU
        outputs through forward hooks while it is).  Like ConvLayer's and ResBlock's fused routes, this one does not call
        the child modules, so forward hooks registered on them do not fire."""
        conv0, conv1 = seq[0], seq[2]
        if (features.dtype == torch.float32 and features.is_cuda and 'head_relu' not in conv_mfma.DISABLED and
                conv_mfma.ACT_OBSERVER is None and isinstance(seq[1], nn.ReLU) and
                conv0.weight.shape[-1] == 3 and conv0.stride == 1 and conv0.padding == 1 and conv0.bias is not None and
                (features.shape[-1] * features.shape[-2]) % 4 == 0):
            y = conv_mfma.conv3x3_bias_act(features, conv0.weight, conv0.bias, 0.0, 1.0, weight_scale=conv0.scale)
            return conv1(y)
        return seq(features)

    @property
    def identity_flow(self):
        """(1, H, W, 2) identity sampling grid (reference attribute, warping_heads.py:158,173-178)."""
        dev = self.flow_out[0].weight.device
        eye = torch.eye(2, 3, device=dev).unsqueeze(0)
        return affine_grid(eye, (1, 1, self.flow_res[0], self.flow_res[1]))

    def forward(self, img, features, output_resolution=None, alpha=None, base_warp=None, stop_grad=False,
                padding_mode='border', return_out_of_bounds=False, image_bounds=None, warp_policy='cartesian',
                unfold=False):
        ds, k = self.flow_downsample, self.num_heads
        low = self._head(self.flow_out, features)              # (N, K*2, h, w)
        mask = self._head(self.mask_out, features)             # (N, K*9*ds*ds, h, w)
        n, _, h, w = low.shape
        policy, assignments = _resolve_policy(warp_policy, img, k)
        if policy == 'assign_only':
            idx = torch.arange(n, device=low.device)
            low = low.reshape(n, k, 2, h, w)[idx, assignments]
            mask = mask.reshape(n, k, 9 * ds * ds, h, w)[idx, assignments]
            split = 1
        else:
            low = low.reshape(n * k, 2, h, w)
            mask = mask.reshape(n * k, 9 * ds * ds, h, w)
            split = k
        if base_warp is not None and base_warp.dim() == 4:
            base_warp = base_warp.reshape(-1, 2, 3)
        flow, delta_flow = flow_compose(low, mask, base_warp, ds)      # (N*split, H, W, 2) each
        if alpha is not None:
            flow = self.identity_flow.lerp(flow, alpha[:, None, None, None])
        if output_resolution is not None:
            flow = flow_resize(flow, output_resolution / flow.size(2))
        if stop_grad:
            flow = flow.detach() + 0 * flow
        img = img.repeat_interleave(split, dim=0) if split > 1 else img
        out = self.warper(img, flow, padding_mode=padding_mode)
        oob = None
        if return_out_of_bounds:
            size = (img.size(0), flow.size(1), flow.size(2)) if output_resolution is None else \
                (img.size(0), img.size(1), output_resolution, output_resolution)
            oob = check_if_warp_exceeds_image_boundaries(flow, image_bounds, size, split)
        if unfold:
            out = out.reshape(out.size(0) // k, k, out.size(1), out.size(2), out.size(3))
            flow = flow.reshape(flow.size(0) // k, k, out.size(3), out.size(4), 2)
            delta_flow = delta_flow.reshape(delta_flow.size(0) // k, k, ds * h, ds * w, 2)
        return out, flow, delta_flow, oob


def apply_affine(matrix, grid):
    """[grid, 1] @ matrix^T per sample (warping_heads.py:268-277); kept for API parity - the training
    path composes inside flow_compose."""
    size = grid.size()
    g = grid.reshape(size[0], -1, 2)
    g = torch.cat([g, torch.ones(g.size(0), g.size(1), 1, device=g.device)], 2)
    return (g @ matrix.permute(0, 2, 1)).reshape(size)
