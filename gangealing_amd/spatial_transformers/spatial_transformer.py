"""SpatialTransformer / ComposedSTN / get_stn with the constructor arguments, state_dict layout and
training-path forward semantics of models/spatial_transformers/spatial_transformer.py:11-139,
388-615.  The point-transfer / flip / propagate helpers of the inference applications (:141-366,
617-726) come from point_transfer.py (mixins)."""
import math

import torch
import torch.nn as nn

from .antialiased_sampling import BilinearDownsample
from .warping_heads import SimilarityHead, FlowHead
from ..stylegan2.networks import EqualLinear, ConvLayer, ResBlock, CHANNELS
from ..op import conv_mfma
from .point_transfer import ComposedStnPointOps, SingleStnPointOps, unravel_index  # noqa: F401


def load_filtered_state_dict(module, state_dict, strict, ignore):
    """The reference drops a few derived buffers from the checkpoint and then loads NON-strictly
    (spatial_transformer.py:378-385,722-726; cluster_classifier.py:98-101), so a truncated or wrong-architecture
    checkpoint loads silently.  Same filtering here, but with strict=True (the default) a missing PARAMETER or a key
    the module does not know raises, as nn.Module.load_state_dict would; derived buffers may be absent."""
    filtered = {k: v for k, v in state_dict.items() if k not in ignore}
    result = nn.Module.load_state_dict(module, filtered, strict=False)
    if strict:
        params = {n for n, _ in module.named_parameters()}
        missing = [k for k in result.missing_keys if k in params]
        if missing or result.unexpected_keys:
            raise RuntimeError(f'Error(s) in loading state_dict for {module.__class__.__name__}: '
                               f'missing parameter(s) {missing}, unexpected key(s) {list(result.unexpected_keys)}')
    return result


def get_stn(transforms, **stn_kwargs):
    assert isinstance(transforms, (str, list))
    if isinstance(transforms, str):
        transforms = [transforms]
    if len(transforms) == 1:
        return SpatialTransformer(transform=transforms[0], **stn_kwargs)
    return ComposedSTN(transforms, **stn_kwargs)


class SpatialTransformer(SingleStnPointOps, nn.Module):
    """ResNet trunk at flow_size^2 -> warp head.  Similarity: trunk goes down to 4x4 then a linear
    layer; flow: trunk stops at flow_size/flow_downsample and feeds the RAFT-style heads."""

    def __init__(self, flow_size, supersize, channel_multiplier=0.5, blur_kernel=(1, 3, 3, 1), num_heads=1,
                 transform='similarity', flow_downsample=8):
        super().__init__()
        if supersize > flow_size:
            self.input_downsample = BilinearDownsample(supersize // flow_size, 3)
        self.input_downsample_required = supersize > flow_size
        self.stn_in_size = flow_size
        self.is_flow = transform == 'flow'
        channels = {r: (c if r <= 32 else c * channel_multiplier) for r, c in CHANNELS.items()}
        log_size = int(math.log(flow_size, 2))
        log_down = int(math.log(flow_downsample, 2))
        end_log = log_size - 4 if self.is_flow else 2
        assert end_log >= 0
        in_ch = int(channels[flow_size])
        convs = [ConvLayer(3, in_ch, 1)]
        n_down = 0
        for i in range(log_size, end_log, -1):
            down = (not self.is_flow) or (n_down < log_down)
            n_down += down
            out_ch = int(channels[2 ** (i - 1)])
            convs.append(ResBlock(in_ch, out_ch, list(blur_kernel), down))
            in_ch = out_ch
        self.convs = nn.Sequential(*convs)
        self.final_conv = ConvLayer(in_ch, CHANNELS[4], 3)
        if not self.is_flow:
            self.final_linear = EqualLinear(CHANNELS[4] * 4 * 4, CHANNELS[4], activation='fused_lrelu')
        if transform == 'similarity':
            self.warp_head = SimilarityHead(CHANNELS[4], antialias=True, num_heads=num_heads,
                                            flow_downsample=flow_downsample)
        elif transform == 'flow':
            shape = (1, in_ch, flow_size // flow_downsample, flow_size // flow_downsample)
            self.warp_head = FlowHead(shape, antialias=True, num_heads=num_heads, flow_downsample=flow_downsample)
        else:
            raise NotImplementedError(transform)

    @property
    def identity_flow(self):
        return self.warp_head.identity_flow

    def load_state_dict(self, state_dict, strict=True):
        ignore = {'warp_head.one_hot', 'input_downsample.kernel_horz', 'input_downsample.kernel_vert',
                  'warp_head.rebias'}
        return load_filtered_state_dict(self, state_dict, strict, ignore)

    def forward(self, input_img, output_resolution=None, iters=1, return_warp=False, return_flow=False,
                return_intermediates=False, return_out_of_bounds=False, intermediate_output_resolution=None,
                stop_grad=False, alpha=None, padding_mode='border', input_img_for_sampling=None, image_bounds=None,
                warp_policy='cartesian', unfold=False, base_warp=None):
        """spatial_transformer.py:471-521: iters == 1 -> single_forward, else iterated_forward."""
        common = dict(output_resolution=output_resolution, return_warp=return_warp, return_flow=return_flow,
                      stop_grad=stop_grad, alpha=alpha, padding_mode=padding_mode,
                      input_img_for_sampling=input_img_for_sampling, return_out_of_bounds=return_out_of_bounds,
                      image_bounds=image_bounds, warp_policy=warp_policy, unfold=unfold, base_warp=base_warp)
        if iters == 1:
            return self.single_forward(input_img, **common)
        return self.iterated_forward(input_img, iters=iters, return_intermediates=return_intermediates,
                                     intermediate_output_resolution=intermediate_output_resolution, **common)

    def iterated_forward(self, input_img, output_resolution=None, iters=1, return_warp=False, return_flow=False,
                         return_intermediates=False, intermediate_output_resolution=None, stop_grad=False, alpha=None,
                         padding_mode='border', input_img_for_sampling=None, return_out_of_bounds=False,
                         image_bounds=None, warp_policy='cartesian', unfold=False, base_warp=None):
        """Apply a similarity STN to its own output `iters` times, composing the warps; pixels are always sampled
        from the original source (spatial_transformer.py:523-567)."""
        assert not self.is_flow, 'iterated_forward is currently only supported for similarity STNs'
        out = input_img
        source = input_img if input_img_for_sampling is None else input_img_for_sampling
        mid = self.stn_in_size if intermediate_output_resolution is None else intermediate_output_resolution
        m = base_warp
        outs, transforms = [], []
        grid = out_of_bounds = None
        for it in range(iters):
            last = it == iters - 1
            out, grid, m, oob = self.single_forward(
                out, output_resolution=output_resolution if last else mid, return_warp=True, return_flow=True,
                return_out_of_bounds=return_out_of_bounds and last, base_warp=m, input_img_for_sampling=source,
                stop_grad=stop_grad, alpha=alpha if last else None, padding_mode=padding_mode,
                image_bounds=image_bounds, warp_policy=warp_policy, unfold=unfold and last, pack=True)
            if return_out_of_bounds and last:
                out_of_bounds = oob
            outs.append(out)
            transforms.append(m)
        if return_intermediates:
            return outs, transforms
        ret = [out] + ([grid] if return_warp else []) + ([m] if return_flow else []) + \
            ([out_of_bounds] if return_out_of_bounds else [])
        return ret[0] if len(ret) == 1 else ret

    def single_forward(self, input_img, output_resolution=None, return_warp=False, return_flow=False,
                       return_out_of_bounds=False, base_warp=None, input_img_for_sampling=None, stop_grad=False,
                       alpha=None, padding_mode='border', image_bounds=None, warp_policy='cartesian', unfold=False,
                       pack=False):
        """spatial_transformer.py:569-615.  pack=True returns everything the warp head returned."""
        regression_input = self.input_downsample(input_img) if input_img.size(-1) > self.stn_in_size else input_img
        source = input_img if input_img_for_sampling is None else input_img_for_sampling
        if self.is_flow:
            feats = self.final_conv(self.convs(regression_input))
        else:
            # the four similarity parameters place every output sample: their trunk runs with fp32-class products in
            # the split-precision mode (conv_mfma.REGRESSION_PRECISION)
            with conv_mfma.regression_precision():
                feats = self.final_conv(self.convs(regression_input))
            feats = self.final_linear(feats.view(feats.shape[0], -1))
        res = output_resolution if output_resolution is not None else self.stn_in_size
        out, grid, m, oob = self.warp_head(source, feats, output_resolution=res, base_warp=base_warp,
                                           stop_grad=stop_grad, alpha=alpha, padding_mode=padding_mode,
                                           return_out_of_bounds=return_out_of_bounds, image_bounds=image_bounds,
                                           warp_policy=warp_policy, unfold=unfold)
        if pack:
            return [out, grid, m, oob]
        ret = [out] + ([grid] if return_warp else []) + ([m] if return_flow else []) + \
            ([oob] if return_out_of_bounds else [])
        return ret[0] if len(ret) == 1 else ret


class ComposedSTN(ComposedStnPointOps, nn.Module):
    """Chains STNs by composing warps (similarity -> flow)."""

    def __init__(self, transforms, **stn_kwargs):
        super().__init__()
        if transforms != ['similarity', 'flow']:
            print('WARNING: ComposedSTN is only tested for transforms=["similarity", "flow"].')
        self.stns = nn.ModuleList([SpatialTransformer(transform=t, **stn_kwargs) for t in transforms])
        self.transforms = transforms[:]
        self.stn_in_size = stn_kwargs['flow_size']
        self.N_minus_1 = len(self.stns) - 1
        self.is_flow = 'flow' in transforms
        self.num_heads = self.stns[0].warp_head.num_heads

    @property
    def identity_flow(self):
        return self.stns[self.transforms.index('flow')].identity_flow

    def load_state_dict(self, state_dict, strict=True):
        ignore = {'warp_head.one_hot'}
        for i in range(len(self.stns)):
            ignore |= {f'stns.{i}.input_downsample.kernel_horz', f'stns.{i}.input_downsample.kernel_vert',
                       f'stns.{i}.warp_head.rebias'}
        return load_filtered_state_dict(self, state_dict, strict, ignore)

    def forward(self, input_img, return_warp=None, return_flow=False, return_sim=False, return_intermediates=False,
                output_resolution=None, unfold=False, iters=1, alpha=None, warp_policy='cartesian',
                input_img_for_sampling=None, **stn_forward_kwargs):
        out = input_img
        source = input_img if input_img_for_sampling is None else input_img_for_sampling
        warp = None
        imgs, warps = [], []
        n = source.size(0)
        last = self.N_minus_1
        for i, stn in enumerate(self.stns):
            if self.num_heads > 1 and warp_policy == 'cartesian' and i > 0:
                policy = torch.eye(self.num_heads, device=source.device).repeat(n, 1)     # one head per replica
            else:
                policy = warp_policy
            out, grid, warp = stn(out, return_warp=True, return_flow=True, input_img_for_sampling=source,
                                  base_warp=warp, output_resolution=output_resolution if i == last else self.stn_in_size,
                                  unfold=unfold if i == last else False, iters=iters if i == 0 else 1,
                                  alpha=alpha if i == last else None, warp_policy=policy, **stn_forward_kwargs)
            if self.num_heads > 1 and warp_policy == 'cartesian' and i == 0:
                source = source.repeat_interleave(self.num_heads, dim=0)
            imgs.append(out)
            warps.append(grid)
            if i == 0:
                sim_out = out
        if return_intermediates:
            return imgs, warps
        ret = [out] + ([grid] if return_warp else []) + ([warp] if return_flow else []) + ([sim_out] if return_sim else [])
        return ret[0] if len(ret) == 1 else ret
