"""StyleGAN2 generator and the conv blocks shared with the STN, built on the HIP operators.

Module / parameter / buffer names follow models/stylegan2/networks.py so that reference
state_dicts (``g_ema``, ``t``, ``t_ema``; train.py:22-28) load unchanged.  What differs is how the
layers execute on MI355X:

  * ModulatedConv2d runs ONE dense implicit-GEMM convolution with shared weights: the style scales
    the activation as it is gathered and the demodulation scales the accumulator in the epilogue
    (op/conv_mfma.py).  The reference materialises (N,Cout,Cin,k,k) weights
    and runs a per-sample grouped convolution (networks.py:243-280).
  * packed GEMM weights and the (Cout,Cin) squared-weight table are cached per weight version: the
    generator is frozen during GANgealing training (train.py:64-65), so they are built once.
  * EqualConv2d folds its runtime ``weight * scale`` into the weight-packing kernel.
The fp16 ``normalize`` branch of the reference (networks.py:237-242) is not implemented: every
GANgealing recipe runs fp32 (num_fp16_res=0, run_fp32=True; networks.py:406,572).
"""
import math
import random

import torch
from torch import nn
from torch.nn import functional as F

from ..op import FusedLeakyReLU, fused_leaky_relu, upfirdn2d
from ..op.fused_act import noise_bias_leaky_relu
from ..op.upfirdn2d import blur_noise_act, blur_noise_act_ok, upfirdn2d_add, blur_down_tap
from ..op import conv_mfma


# How ModulatedConv2d executes:
#   'shared'  (default) one dense convolution with shared weights; style and demodulation ride in the gather / epilogue
#   'grouped' the reference's own operator sequence (networks.py:233-282): per-sample weights scale * W * style
#             (* demod) materialised as an (N * Cout, Cin, k, k) tensor and a per-sample GROUPED convolution issued
#             through op.conv2d_gradfix (conv2d / conv_transpose2d with groups = N) - what an unmodified reference
#             networks.py runs on these operators (`python -m gangealing_amd.launch .../train.py`).
#             No fusion across layers applies in this form; it exists so that the literal drop-in route is tested and
#             timed (tests/test_gpu_dropin.py, bench.py extras.dropin_route).
import contextlib as _contextlib
import os as _os
MODCONV_FORM = _os.environ.get('GANGEALING_MODCONV', 'shared')


@_contextlib.contextmanager
def modconv_form(form):
    global MODCONV_FORM
    if form not in ('shared', 'grouped'):
        raise ValueError(f'modconv form {form!r}: shared | grouped')
    old, MODCONV_FORM = MODCONV_FORM, form
    try:
        yield
    finally:
        MODCONV_FORM = old


class PixelNorm(nn.Module):
    def forward(self, input):
        return input * torch.rsqrt(torch.mean(input ** 2, dim=1, keepdim=True) + 1e-8)


def make_kernel(k):
    k = torch.tensor(k, dtype=torch.float32)
    if k.ndim == 1:
        k = k[None, :] * k[:, None]
    k /= k.sum()
    return k


class Upsample(nn.Module):
    def __init__(self, kernel, factor=2):
        super().__init__()
        self.factor = factor
        self.register_buffer('kernel', make_kernel(kernel) * (factor ** 2))
        p = self.kernel.shape[0] - factor
        self.pad = ((p + 1) // 2 + factor - 1, p // 2)

    def forward(self, input):
        return upfirdn2d(input, self.kernel.type(input.dtype), up=self.factor, down=1, pad=self.pad)


class Downsample(nn.Module):
    def __init__(self, kernel, factor=2):
        super().__init__()
        self.factor = factor
        self.register_buffer('kernel', make_kernel(kernel))
        p = self.kernel.shape[0] - factor
        self.pad = ((p + 1) // 2, p // 2)

    def forward(self, input):
        return upfirdn2d(input, self.kernel.type(input.dtype), up=1, down=self.factor, pad=self.pad)


class Blur(nn.Module):
    def __init__(self, kernel, pad, upsample_factor=1):
        super().__init__()
        kernel = make_kernel(kernel)
        if upsample_factor > 1:
            kernel = kernel * (upsample_factor ** 2)
        self.register_buffer('kernel', kernel)
        self.pad = pad

    def forward(self, input):
        return upfirdn2d(input, self.kernel.type(input.dtype), pad=self.pad)


class EqualConv2d(nn.Module):
    def __init__(self, in_channel, out_channel, kernel_size, stride=1, padding=0, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_channel, in_channel, kernel_size, kernel_size))
        self.scale = 1 / math.sqrt(in_channel * kernel_size ** 2)
        self.stride = stride
        self.padding = padding
        self.bias = nn.Parameter(torch.zeros(out_channel)) if bias else None

    def forward(self, input):
        return conv_mfma.conv2d(input, self.weight, bias=self.bias, stride=self.stride, padding=self.padding,
                                weight_scale=self.scale)

    def __repr__(self):
        return (f'{self.__class__.__name__}({self.weight.shape[1]}, {self.weight.shape[0]},'
                f' {self.weight.shape[2]}, stride={self.stride}, padding={self.padding})')


class EqualLinear(nn.Module):
    def __init__(self, in_dim, out_dim, bias=True, bias_init=0, lr_mul=1, activation=None):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_dim, in_dim).div_(lr_mul))
        self.bias = nn.Parameter(torch.zeros(out_dim).fill_(bias_init)) if bias else None
        self.activation = activation
        self.scale = (1 / math.sqrt(in_dim)) * lr_mul
        self.lr_mul = lr_mul

    def _scaled(self):
        """(weight * scale, bias * lr_mul); cached while the parameters are frozen (the generator's mapping network
        and modulation layers: two element-wise launches less per layer and pass)."""
        w, b = self.weight, self.bias
        if w.requires_grad or (b is not None and b.requires_grad):
            return w * self.scale, (None if b is None else b * self.lr_mul)
        key = (w._version, w.data_ptr(), None if b is None else (b._version, b.data_ptr()))
        if getattr(self, '_scaled_cache', None) is None or self._scaled_cache[0] != key:
            # (inference_mode(False): a cache first filled under torch.inference_mode - the reference's training visuals
            # run before the first iteration, training_vis.py:128 - would hold inference tensors, which autograd refuses
            # to save for backward when the training step uses the cache)
            with torch.inference_mode(False), torch.no_grad():
                self._scaled_cache = (key, (w * self.scale).contiguous(), None if b is None else b * self.lr_mul)
        return self._scaled_cache[1], self._scaled_cache[2]

    def forward(self, input):
        weight, bias = self._scaled()
        if self.activation:
            out = F.linear(input, weight)
            return fused_leaky_relu(out, bias)
        return F.linear(input, weight, bias=bias)

    def __repr__(self):
        return f'{self.__class__.__name__}({self.weight.shape[1]}, {self.weight.shape[0]})'


class ScaledLeakyReLU(nn.Module):
    def __init__(self, negative_slope=0.2):
        super().__init__()
        self.negative_slope = negative_slope

    def forward(self, input):
        return F.leaky_relu(input, negative_slope=self.negative_slope) * math.sqrt(2)


class ModulatedConv2d(nn.Module):
    def __init__(self, in_channel, out_channel, kernel_size, style_dim, demodulate=True, upsample=False,
                 downsample=False, blur_kernel=(1, 3, 3, 1), normalize=False):
        super().__init__()
        if downsample:
            raise NotImplementedError('ModulatedConv2d(downsample=True) is not used by the generator')
        if normalize:
            raise NotImplementedError('fp16 normalize branch (networks.py:237-242) is off on every GANgealing recipe')
        self.eps = 1e-8
        self.kernel_size = kernel_size
        self.in_channel = in_channel
        self.out_channel = out_channel
        self.upsample = upsample
        self.downsample = downsample
        if upsample:
            factor = 2
            p = (len(blur_kernel) - factor) - (kernel_size - 1)
            self.blur = Blur(list(blur_kernel), pad=((p + 1) // 2 + factor - 1, p // 2 + 1), upsample_factor=factor)
        self.scale = 1 / math.sqrt(in_channel * kernel_size ** 2)
        self.padding = kernel_size // 2
        self.weight = nn.Parameter(torch.randn(1, out_channel, in_channel, kernel_size, kernel_size))
        self.modulation = EqualLinear(style_dim, in_channel, bias_init=1)
        self.demodulate = demodulate
        self._packed = None        # (weight version, device, wmat_fwd, wmat_bwd, wsq)

    def __repr__(self):
        return (f'{self.__class__.__name__}({self.in_channel}, {self.out_channel}, {self.kernel_size}, '
                f'upsample={self.upsample}, downsample={self.downsample})')

    def _weights(self):
        w = self.weight
        key = (w._version, w.device, w.data_ptr())
        if self._packed is None or self._packed[0] != key:
            if w.requires_grad and torch.is_grad_enabled():
                raise NotImplementedError('the fused modulated convolution treats the generator weights as frozen '
                                          '(train.py:64-65); call requires_grad_(False) on the generator')
            with torch.inference_mode(False), torch.no_grad():        # (see EqualLinear._scaled)
                w4 = w[0]
                k, cin, cout = self.kernel_size, self.in_channel, self.out_channel
                w4 = w4.detach().clone()          # the packs are built lazily per arithmetic mode
                wmat_fwd = conv_mfma.PackedWeight(w4, 1, cout, cin, k, 0, 0, self.scale)
                # dgrad: reduce over co.  plain conv -> flipped taps; transposed stride-2 conv -> strided correlation
                wmat_bwd = conv_mfma.PackedWeight(w4, 1, cin, cout, k, 1, 0 if self.upsample else 1, self.scale)
                wsq = (w4 * self.scale).pow(2).sum(dim=(2, 3)).contiguous()
            self._packed = (key, wmat_fwd, wmat_bwd, wsq)
        return self._packed[1:]

    def modulate(self, input, style, wsq):
        """latent slot -> (style (N, Cin), demodulation (N, Cout) or None when left to the convolution function)."""
        mod = self.modulation
        if ('style_demod' not in conv_mfma.DISABLED and input.dtype == torch.float32 and style.dtype == torch.float32 and
                mod.activation is None and
                not (torch.is_grad_enabled() and (style.requires_grad or mod.weight.requires_grad or
                                                  (mod.bias is not None and mod.bias.requires_grad)))):
            # no gradient wanted for the style: EqualLinear + demodulation in one launch (csrc/modulation.hip)
            return conv_mfma.style_demod(style, mod.weight, mod.bias, mod.scale, mod.lr_mul,
                                         wsq if self.demodulate else None, self.eps)
        if ('style_demod' not in conv_mfma.DISABLED and 'style_demod_grad' not in conv_mfma.DISABLED and
                input.dtype == torch.float32 and style.dtype == torch.float32 and mod.activation is None and
                style.dim() == 2 and not mod.weight.requires_grad and not (mod.bias is not None and mod.bias.requires_grad)):
            # a learned W+ slot through a frozen modulation layer (the latent learner's slots): the same single launch,
            # with a gradient for the latent
            return conv_mfma.style_demod_grad(style, mod.weight, mod.bias, mod._scaled()[0], mod.scale, mod.lr_mul,
                                              wsq if self.demodulate else None, self.eps)
        return mod(style), None

    def bank_entry(self, slot):
        """This layer's row of a conv_mfma.StyleBank (modulation + demodulation from W+ slot `slot`)."""
        _, _, wsq = self._weights()
        mod = self.modulation
        return (slot, mod.weight.detach(), None if mod.bias is None else mod.bias.detach(), mod.scale, mod.lr_mul,
                wsq if self.demodulate else None, self.eps)

    def forward_grouped(self, input, style):
        """The reference's formulation (networks.py:233-282) on the drop-in operators: per-sample weights, one grouped
        convolution with groups = batch."""
        from ..op import conv2d_gradfix
        n, cin, h, w = input.shape
        k, cout = self.kernel_size, self.out_channel
        weight = (self.scale * self.weight) * self.modulation(style).view(n, 1, cin, 1, 1)
        if self.demodulate:
            demod = torch.rsqrt(weight.pow(2).sum([2, 3, 4]) + self.eps)
            weight = weight * demod.view(n, cout, 1, 1, 1)
        if self.upsample:
            per_sample = weight.transpose(1, 2).reshape(n * cin, cout, k, k)
            out = conv2d_gradfix.conv_transpose2d(input.reshape(1, n * cin, h, w), per_sample, padding=0, stride=2,
                                                  groups=n)
            return self.blur(out.view(n, cout, out.shape[-2], out.shape[-1]))
        out = conv2d_gradfix.conv2d(input.reshape(1, n * cin, h, w), weight.view(n * cout, cin, k, k),
                                    padding=self.padding, groups=n)
        return out.view(n, cout, out.shape[-2], out.shape[-1])

    def forward(self, input, style, act=None, pre=None, bias=None, prelimb=None):
        """act = (noise, noise_weight, bias, negative_slope, scale): fuse StyledConv's NoiseInjection +
        FusedLeakyReLU into the convolution (callers check `can_fuse_act` first).  pre = (style vector, demodulation)
        already computed for this layer (Generator's style bank); bias: frozen per-channel bias for the epilogue.
        prelimb (up-sampling layers): `input` x this layer's style already in limb form (conv_mfma.torgb_limb)."""
        if MODCONV_FORM == 'grouped':
            if act is not None or pre is not None or bias is not None:
                raise RuntimeError('grouped (reference-form) modulated convolution: the fused paths do not apply')
            return self.forward_grouped(input, style)
        wmat_fwd, wmat_bwd, wsq = self._weights()
        if pre is not None:
            style, demod = pre
        else:
            style, demod = self.modulate(input, style, wsq)
        if self.upsample and act is not None:
            # up-sampling layer: the activation follows the blur, so it rides in the blur kernel instead
            out = conv_mfma.modulated_conv2d(input, style, wmat_fwd, wmat_bwd, wsq, self.kernel_size,
                                             upsample=True, demodulate=self.demodulate, demod=demod, prelimb=prelimb)
            noise, noise_weight, bias, negative_slope, scale = act
            return blur_noise_act(out, self.blur.kernel, self.blur.pad, noise, noise_weight, bias, negative_slope, scale)
        out = conv_mfma.modulated_conv2d(input, style, wmat_fwd, wmat_bwd, wsq, self.kernel_size,
                                         upsample=self.upsample, demodulate=self.demodulate, act=act, demod=demod,
                                         bias=bias, prelimb=prelimb if self.upsample else None)
        if self.upsample:
            out = self.blur(out)
        return out

    def can_fuse_act(self, input, style, *frozen):
        """The one-kernel StyledConv applies to 3x3 layers without upsampling when neither the style (i.e. the
        latent and the modulation layer) nor the activation parameters need a gradient."""
        if (self.kernel_size != 3 or input.dtype != torch.float32 or 'fuse_act' in conv_mfma.DISABLED
                or MODCONV_FORM == 'grouped'):
            return False
        if self.upsample:
            h, w = 2 * input.shape[-2] + 1, 2 * input.shape[-1] + 1           # transposed-conv output
            if not blur_noise_act_ok(input.new_empty((1, 1, h, w)), self.blur.kernel, self.blur.pad):
                return False
        elif (input.shape[-1] * input.shape[-2]) % 4:
            return False
        if torch.is_grad_enabled() and (style.requires_grad or self.modulation.weight.requires_grad or
                                        any(p.requires_grad for p in frozen)):
            return False
        return True


class NoiseInjection(nn.Module):
    def __init__(self):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(1))

    def forward(self, image, noise=None):
        if noise is None:
            batch, _, height, width = image.shape
            noise = image.new_empty(batch, 1, height, width).normal_()
        else:
            noise = noise.type(image.dtype)
        return image + self.weight.type(image.dtype) * noise


class ConstantInput(nn.Module):
    def __init__(self, channel, size=4):
        super().__init__()
        self.input = nn.Parameter(torch.randn(1, channel, size, size))
        self.size = size

    def forward(self, input):
        return self.input.repeat(input.shape[0], 1, 1, 1)


class StyledConv(nn.Module):
    def __init__(self, in_channel, out_channel, kernel_size, style_dim, upsample=False, blur_kernel=(1, 3, 3, 1),
                 demodulate=True, normalize=False):
        super().__init__()
        self.conv = ModulatedConv2d(in_channel, out_channel, kernel_size, style_dim, upsample=upsample,
                                    blur_kernel=blur_kernel, demodulate=demodulate, normalize=normalize)
        self.noise = NoiseInjection()
        self.activate = FusedLeakyReLU(out_channel)

    def forward(self, input, style, noise=None, pre=None, prelimb=None):
        if self.conv.can_fuse_act(input, style, self.noise.weight, self.activate.bias):
            n, _, h, w = input.shape
            if self.conv.upsample:
                h, w = 2 * h, 2 * w
            if noise is None:
                noise = input.new_empty(n, 1, h, w).normal_()
            elif noise.shape[0] != n:
                noise = noise.expand(n, -1, -1, -1)
            return self.conv(input, style, act=(noise.type(input.dtype), self.noise.weight, self.activate.bias,
                                                self.activate.negative_slope, self.activate.scale), pre=pre,
                             prelimb=prelimb)
        out = self.conv(input, style, pre=pre, prelimb=prelimb)
        n, _, h, w = out.shape
        if out.dtype == torch.float32 and (h * w) % 4 == 0 and MODCONV_FORM != 'grouped':
            # NoiseInjection + FusedLeakyReLU in one pass over the activation (csrc/fused_bias_act.hip)
            if noise is None:
                noise = out.new_empty(n, 1, h, w).normal_()
            elif noise.shape[0] != n:
                noise = noise.expand(n, -1, -1, -1)
            return noise_bias_leaky_relu(out, noise.type(out.dtype), self.noise.weight, self.activate.bias,
                                         self.activate.negative_slope, self.activate.scale)
        out = self.noise(out, noise=noise)
        return self.activate(out)


def styled_conv_with_rgb(styled, to_rgb, input, style, rgb_latent, noise=None, pre=None, pre_rgb=None, next_up=None):
    """StyledConv (no up-sampling) + the ToRGB convolution of the same resolution as ONE autograd node
    (conv_mfma._StyledConvToRGB); returns (activation, raw rgb, prelimb) or None when the pair cannot be fused (a style
    that needs a gradient, trainable generator weights, shapes off the fused path).
    next_up = (the next resolution's up-sampling ModulatedConv2d, its (style, demodulation) from the style bank): the
    ToRGB pass then also writes the activation in that layer's limb form (prelimb = (xlimb, xexp), else None).  Without
    a gradient to carry (generator pass 1) the pair is fused only for that purpose."""
    conv, rgb_conv = styled.conv, to_rgb.conv
    want_limb = (next_up is not None and next_up[1] is not None and to_rgb.epilogue_bias() is not None and
                 conv_mfma.prelimb_wanted(next_up[0].in_channel, input.shape[-1]) and input.shape[-1] == input.shape[-2])
    needs_node = torch.is_grad_enabled() and input.requires_grad
    if 'torgb_fuse' in conv_mfma.DISABLED or MODCONV_FORM == 'grouped' or not (needs_node or want_limb):
        return None
    if not conv.can_fuse_act(input, style, styled.noise.weight, styled.activate.bias) or conv.upsample:
        return None
    if rgb_latent.requires_grad or rgb_conv.modulation.weight.requires_grad or rgb_conv.weight.requires_grad:
        return None
    n, _, h, w = input.shape
    if noise is None:
        noise = input.new_empty(n, 1, h, w).normal_()
    elif noise.shape[0] != n:
        noise = noise.expand(n, -1, -1, -1)
    wmat_fwd, wmat_bwd, wsq = conv._weights()
    s_c, demod = pre if pre is not None else conv.modulate(input, style, wsq)
    rgb_fwd, _, _ = rgb_conv._weights()
    s_rgb = pre_rgb[0] if pre_rgb is not None else rgb_conv.modulate(input, rgb_latent, None)[0]
    act = (noise.type(input.dtype), styled.noise.weight, styled.activate.bias, styled.activate.negative_slope,
           styled.activate.scale)
    w_rgb = rgb_conv.weight.detach().reshape(3, rgb_conv.in_channel).contiguous()
    return conv_mfma.styled_conv_torgb(input, s_c, wmat_fwd, wmat_bwd, wsq, conv.demodulate, act, demod, s_rgb, rgb_fwd,
                                       w_rgb, rgb_conv.scale, to_rgb.epilogue_bias(),
                                       next_style=next_up[1][0] if want_limb else None)


class ToRGB(nn.Module):
    def __init__(self, in_channel, style_dim, upsample=True, blur_kernel=(1, 3, 3, 1), normalize=False):
        super().__init__()
        if upsample:
            self.upsample = Upsample(list(blur_kernel))
        self.conv = ModulatedConv2d(in_channel, 3, 1, style_dim, demodulate=False, normalize=normalize)
        self.bias = nn.Parameter(torch.zeros(1, 3, 1, 1))

    def epilogue_bias(self):
        """The bias as a (3,) vector for the 1x1 convolution's epilogue, or None when it needs a gradient (then
        finish() adds it as the reference does) or the fusion is switched off."""
        if ('torgb_bias' in conv_mfma.DISABLED or (self.bias.requires_grad and torch.is_grad_enabled())
                or MODCONV_FORM == 'grouped'):
            return None
        return self.bias.detach().reshape(-1)

    def forward(self, input, style, skip=None, pre=None):
        bias = self.epilogue_bias() if input.dtype == torch.float32 else None
        return self.finish(self.conv(input, style, pre=pre, bias=bias), skip, bias_done=bias is not None)

    def finish(self, rgb, skip=None, bias_done=False):
        """bias + up-sampled skip connection around the raw modulated 1x1 convolution output (networks.py:366-371);
        the bias rides in the convolution's epilogue and the skip addition in the up-sampling kernel where possible."""
        out = rgb if bias_done else rgb + self.bias.type(rgb.dtype)
        if skip is not None:
            up = self.upsample
            if (out.dtype == torch.float32 and skip.dtype == torch.float32 and 'torgb_bias' not in conv_mfma.DISABLED
                    and MODCONV_FORM != 'grouped'):
                out = upfirdn2d_add(skip, up.kernel, out, up=up.factor, down=1, pad=up.pad)
            else:
                out = out.float() + up(skip)
        return out


class ConvLayer(nn.Sequential):
    def __init__(self, in_channel, out_channel, kernel_size, downsample=False, blur_kernel=(1, 3, 3, 1), bias=True,
                 activate=True):
        layers = []
        if downsample:
            factor = 2
            p = (len(blur_kernel) - factor) + (kernel_size - 1)
            layers.append(Blur(list(blur_kernel), pad=((p + 1) // 2, p // 2)))
            stride = 2
            self.padding = 0
        else:
            stride = 1
            self.padding = kernel_size // 2
        layers.append(EqualConv2d(in_channel, out_channel, kernel_size, padding=self.padding, stride=stride,
                                  bias=bias and not activate))
        if activate:
            layers.append(FusedLeakyReLU(out_channel) if bias else ScaledLeakyReLU(0.2))
        super().__init__(*layers)

    def forward(self, input):
        # 3x3 / stride 1 conv followed by FusedLeakyReLU: bias + activation ride in the convolution's epilogue
        if (len(self) == 2 and isinstance(self[0], EqualConv2d) and isinstance(self[1], FusedLeakyReLU) and
                self[0].weight.shape[-1] == 3 and self[0].stride == 1 and self[0].padding == 1 and
                self[0].bias is None and input.dtype == torch.float32 and input.is_cuda):
            conv, act = self[0], self[1]
            return conv_mfma.conv3x3_bias_act(input, conv.weight, act.bias, act.negative_slope, act.scale,
                                              weight_scale=conv.scale)
        # Blur followed by a 1x1 / stride-2 convolution (ResBlock.skip, networks.py:383-386): the convolution reads only
        # every second blurred pixel, so the blur is evaluated at those positions only (upfirdn2d with down = 2: the
        # same taps on the same inputs, a quarter of the outputs and of the intermediate tensor) and the 1x1
        # convolution becomes a dense stride-1 one
        if (len(self) == 2 and isinstance(self[0], Blur) and isinstance(self[1], EqualConv2d) and
                self[1].weight.shape[-1] == 1 and self[1].stride == 2 and self[1].padding == 0 and
                input.dtype == torch.float32 and input.is_cuda and 'skip_down' not in conv_mfma.DISABLED):
            blur, conv = self[0], self[1]
            x = upfirdn2d(input, blur.kernel, up=1, down=2, pad=blur.pad)
            return conv_mfma.conv2d(x, conv.weight, bias=conv.bias, stride=1, padding=0, weight_scale=conv.scale)
        return super().forward(input)


class ResBlock(nn.Module):
    def __init__(self, in_channel, out_channel, blur_kernel=(1, 3, 3, 1), downsample=True):
        super().__init__()
        self.conv1 = ConvLayer(in_channel, in_channel, 3)
        self.conv2 = ConvLayer(in_channel, out_channel, 3, downsample=downsample)
        self.skip = ConvLayer(in_channel, out_channel, 1, downsample=downsample, activate=False, bias=False)

    def forward(self, input):
        # (not in the exact-product fp32 mode: that mode keeps the reference's operation ORDER as well as its products, so
        # that its un-pinned branch decisions are the reference's wherever float32 rounding decides them -
        # tests/test_gpu_stn_decisions.py: with the fold, one image whose mip level sits on the clamp at 0 takes the other
        # side and the pinned similarity-stage distance goes from 1.3e-5 to 9e-4, profiles/r06_h_similarity_stage_dice.txt)
        if (input.dtype == torch.float32 and input.is_cuda and 'resblock_fold' not in conv_mfma.DISABLED
                and conv_mfma.PRECISION != 'fp32' and self._foldable()):
            return self._forward_folded(input)
        out = self.conv2(self.conv1(input))
        return conv_mfma.add_scale(out, self.skip(input), 1.0 / math.sqrt(2))

    def _foldable(self):
        c2, sk = self.conv2, self.skip
        if isinstance(c2[0], Blur):
            return (len(c2) == 3 and isinstance(c2[1], EqualConv2d) and isinstance(c2[2], FusedLeakyReLU) and
                    c2[1].bias is None and len(sk) == 2 and isinstance(sk[0], Blur) and isinstance(sk[1], EqualConv2d) and
                    sk[1].weight.shape[-1] == 1 and sk[1].stride == 2 and sk[1].padding == 0 and sk[1].bias is None and
                    tuple(sk[0].kernel.shape) == (4, 4))
        return (len(c2) == 2 and isinstance(c2[0], EqualConv2d) and isinstance(c2[1], FusedLeakyReLU) and
                c2[0].bias is None and c2[0].weight.shape[-1] == 3 and c2[0].stride == 1 and c2[0].padding == 1 and
                len(sk) == 1 and isinstance(sk[0], EqualConv2d) and sk[0].bias is None and sk[0].stride == 1)

    def _forward_folded(self, input):
        """(conv2(conv1(x)) + skip(x)) / sqrt(2) (networks.py:388-393) with the 1 / sqrt(2) folded into the two branches -
        conv2's activation gain (sqrt(2) * 1 / sqrt(2)) and the skip convolution's equalised-lr weight scale - so that the
        merge is a plain sum and its backward hands the incoming gradient to both branches unscaled (one full-size
        multiply less per block); with down-sampling the skip branch's blur is a tap node (blur_down_tap) whose backward
        ADDS its adjoint into the gradient conv1 produced for the shared input (one full-size add less per block)."""
        s = 1.0 / math.sqrt(2)
        c2, sk = self.conv2, self.skip
        if isinstance(c2[0], Blur):
            x, xs = blur_down_tap(input, sk[0].kernel, sk[0].pad)
            c1 = self.conv1
            if ('conv_blur_bwd' not in conv_mfma.DISABLED and conv_mfma.ACT_OBSERVER is None and len(c1) == 2 and
                    isinstance(c1[0], EqualConv2d) and isinstance(c1[1], FusedLeakyReLU) and c1[0].bias is None and
                    c1[0].weight.shape[-1] == 3 and c1[0].stride == 1 and c1[0].padding == 1 and
                    tuple(c2[0].kernel.shape) == (4, 4) and min(x.shape[-2:]) >= 24 and (x.shape[-1] * x.shape[-2]) % 4 == 0):
                # conv1 + its activation + the Blur as one node: the Blur's adjoint carries the activation's backward
                y = conv_mfma.conv3x3_bias_act_blur(x, c1[0].weight, c1[1].bias, c2[0].kernel, c2[0].pad,
                                                    c1[1].negative_slope, c1[1].scale, weight_scale=c1[0].scale)
            else:
                y = c2[0](c1(x))
            if 'conv_s2_act' in conv_mfma.DISABLED or conv_mfma.ACT_OBSERVER is not None:
                y = conv_mfma.conv2d(y, c2[1].weight, bias=None, stride=2, padding=0, weight_scale=c2[1].scale)
                y = fused_leaky_relu(y, c2[2].bias, c2[2].negative_slope, c2[2].scale * s)
            else:           # convolution + FusedLeakyReLU as one node, the activation inside the library
                y = conv_mfma.conv3x3s2_bias_act(y, c2[1].weight, c2[2].bias, c2[2].negative_slope, c2[2].scale * s,
                                                 weight_scale=c2[1].scale)
            sconv, sin = sk[1], xs
        else:
            y = conv_mfma.conv3x3_bias_act(self.conv1(input), c2[0].weight, c2[1].bias, c2[1].negative_slope,
                                           c2[1].scale * s, weight_scale=c2[0].scale)
            sconv, sin = sk[0], input
        if sconv.weight.shape[-1] == 1 and sconv.padding == 0 and 'conv_residual' not in conv_mfma.DISABLED:
            # the merge (y + skip) rides in the 1x1 skip convolution's epilogue / split-K reduce pass
            return conv_mfma.conv2d(sin, sconv.weight, bias=None, stride=1, padding=0, weight_scale=sconv.scale * s,
                                    residual=y)
        skip = conv_mfma.conv2d(sin, sconv.weight, bias=None, stride=1, padding=sconv.padding, weight_scale=sconv.scale * s)
        return conv_mfma.add_scale(y, skip, 1.0)


CHANNELS = {4: 512, 8: 512, 16: 512, 32: 512, 64: 256, 128: 128, 256: 64, 512: 32, 1024: 16}


class Generator(nn.Module):
    def __init__(self, size, style_dim, n_mlp, channel_multiplier=2, blur_kernel=(1, 3, 3, 1), lr_mlp=0.01,
                 num_fp16_res=0, run_fp32=True):
        super().__init__()
        if num_fp16_res != 0 and not run_fp32:
            raise NotImplementedError('fp16 generator layers are not part of the GANgealing recipes')
        self.size = size
        self.style_dim = style_dim
        self.style = nn.Sequential(PixelNorm(), *[
            EqualLinear(style_dim, style_dim, lr_mul=lr_mlp, activation='fused_lrelu') for _ in range(n_mlp)])
        self.channels = {r: (c if r <= 32 else c * channel_multiplier) for r, c in CHANNELS.items()}
        self.input = ConstantInput(self.channels[4])
        self.conv1 = StyledConv(self.channels[4], self.channels[4], 3, style_dim, blur_kernel=blur_kernel)
        self.to_rgb1 = ToRGB(self.channels[4], style_dim, upsample=False)
        self.log_size = int(math.log(size, 2))
        self.num_layers = (self.log_size - 2) * 2 + 1
        self.convs = nn.ModuleList()
        self.upsamples = nn.ModuleList()
        self.to_rgbs = nn.ModuleList()
        self.noises = nn.Module()
        for layer_idx in range(self.num_layers):
            res = (layer_idx + 5) // 2
            self.noises.register_buffer(f'noise_{layer_idx}', torch.randn(1, 1, 2 ** res, 2 ** res))
        in_channel = self.channels[4]
        for i in range(3, self.log_size + 1):
            out_channel = self.channels[2 ** i]
            self.convs.append(StyledConv(in_channel, out_channel, 3, style_dim, upsample=True, blur_kernel=blur_kernel))
            self.convs.append(StyledConv(out_channel, out_channel, 3, style_dim, blur_kernel=blur_kernel))
            self.to_rgbs.append(ToRGB(out_channel, style_dim))
            in_channel = out_channel
        self.n_latent = self.log_size * 2 - 2
        self.num_fp16_res = num_fp16_res
        self.run_fp32 = run_fp32

    def make_noise(self, batch_size=1):
        device = self.input.input.device
        noises = [torch.randn(batch_size, 1, 4, 4, device=device)]
        for i in range(3, self.log_size + 1):
            noises += [torch.randn(batch_size, 1, 2 ** i, 2 ** i, device=device) for _ in range(2)]
        return noises

    def batch_latent(self, n_latent):
        return self.style(torch.randn(n_latent, self.style_dim, device=self.input.input.device))

    def mean_latent(self, n_latent):
        return self.batch_latent(n_latent).mean(dim=0, keepdim=True)

    def get_latent(self, input):
        return self.style(input)

    def forward(self, styles, mapping_only=False, return_latents=False, inject_index=None, truncation=1,
                truncation_latent=None, input_is_latent=False, noise=None, randomize_noise=True, grad_latents=None):
        """Reference signature (networks.py:514-525) plus `grad_latents`: when given, only the first
        `grad_latents` W+ slots can carry a gradient (the DirectionInterpolator only writes the first
        `inject` slots, latent_learner.py:64-67; the rest is a copy of w that needs no gradient), so the
        style-gradient reductions of all later layers are skipped.  None = reference behaviour."""
        if not input_is_latent:
            styles = [self.style(s) for s in styles]
            if mapping_only:
                return styles
        if noise is None:
            if randomize_noise and 'noise_bank' not in conv_mfma.DISABLED:
                noise = self._noise_bank(styles[0].shape[0], styles[0].device, styles[0].dtype)
            else:
                noise = [None] * self.num_layers if randomize_noise else \
                    [getattr(self.noises, f'noise_{i}') for i in range(self.num_layers)]
        if truncation < 1:
            styles = [truncation_latent + truncation * (styles[0] - truncation_latent), styles[0]]
        if len(styles) < 2 or inject_index == self.n_latent:
            latent = styles[0] if styles[0].ndim >= 3 else styles[0].unsqueeze(1).repeat(1, self.n_latent, 1)
        else:
            if inject_index is None:
                inject_index = random.randint(1, self.n_latent - 1)
            latent = torch.cat([styles[0].unsqueeze(1).repeat(1, inject_index, 1),
                                styles[1].unsqueeze(1).repeat(1, self.n_latent - inject_index, 1)], 1)
        if grad_latents is not None and latent.requires_grad:
            # one `unbind` (backward: one `stack`) instead of a `select` per slot, whose backward writes a full-size zero
            # tensor and adds it into the latent's gradient once per slot
            lat = list(latent[:, :grad_latents].unbind(1)) + list(latent[:, grad_latents:].detach().unbind(1))
        else:
            lat = list(latent.unbind(1))
        first_free = 0
        if latent.requires_grad and torch.is_grad_enabled():
            first_free = self.n_latent if grad_latents is None else grad_latents
        pre = self._styles(latent, first_free)          # layer -> (style, demodulation) for W+ slots >= first_free
        out = self.conv1(self.input(latent), lat[0], noise=noise[0], pre=pre.get(self.conv1.conv))
        skip = self.to_rgb1(out, lat[1], pre=pre.get(self.to_rgb1.conv))
        i = 1
        prelimb = None                      # the next up-sampling layer's operand in limb form (conv_mfma.torgb_limb)
        ups = list(self.convs[::2])
        for k, (conv_up, conv, n_up, n_conv, to_rgb) in enumerate(zip(ups, self.convs[1::2], noise[1::2], noise[2::2],
                                                                      self.to_rgbs)):
            out = conv_up(out, lat[i], noise=n_up, pre=pre.get(conv_up.conv), prelimb=prelimb)
            prelimb = None
            last = to_rgb is self.to_rgbs[-1]
            # every resolution but the last: the activation feeds ToRGB AND the next up-sampling layer -> one node
            next_up = None if last else (ups[k + 1].conv, pre.get(ups[k + 1].conv))
            pair = None if last else styled_conv_with_rgb(conv, to_rgb, out, lat[i + 1], lat[i + 2], noise=n_conv,
                                                          pre=pre.get(conv.conv), pre_rgb=pre.get(to_rgb.conv),
                                                          next_up=next_up)
            if pair is not None:
                out, rgb, prelimb = pair
                skip = to_rgb.finish(rgb, skip, bias_done=to_rgb.epilogue_bias() is not None)
            else:
                out = conv(out, lat[i + 1], noise=n_conv, pre=pre.get(conv.conv))
                skip = to_rgb(out, lat[i + 2], skip, pre=pre.get(to_rgb.conv))
            i += 2
        return (skip, latent) if return_latents else (skip, None)

    def _layer_slots(self):
        """[(ModulatedConv2d, W+ slot)] in execution order (networks.py:560-583)."""
        order = [(self.conv1.conv, 0), (self.to_rgb1.conv, 1)]
        i = 1
        for conv_up, conv, to_rgb in zip(self.convs[::2], self.convs[1::2], self.to_rgbs):
            order += [(conv_up.conv, i), (conv.conv, i + 1), (to_rgb.conv, i + 2)]
            i += 2
        return order

    def _styles(self, latent, first_free):
        """Modulation + demodulation vectors of every layer whose W+ slot needs no gradient (slot >= first_free), in
        two launches (conv_mfma.StyleBank) instead of two per layer.  {} when the bank does not apply."""
        if ('style_bank' in conv_mfma.DISABLED or 'style_demod' in conv_mfma.DISABLED or latent.dtype != torch.float32
                or not latent.is_cuda or latent.dim() != 3 or first_free >= self.n_latent or MODCONV_FORM == 'grouped'):
            return {}
        layers = [(m, slot) for m, slot in self._layer_slots() if slot >= first_free]
        frozen = all(not (m.modulation.weight.requires_grad or m.weight.requires_grad or
                          (m.modulation.bias is not None and m.modulation.bias.requires_grad)) for m, _ in layers)
        if not layers or (torch.is_grad_enabled() and not frozen) or any(m.modulation.activation for m, _ in layers):
            return {}
        key = (latent.device,) + tuple((m.weight._version, m.modulation.weight._version, m.weight.data_ptr())
                                       for m, _ in layers)
        banks = self.__dict__.setdefault('_banks', {})           # one bank per first_free (pass 1: 0, pass 2: inject)
        cache = banks.get(first_free)
        if cache is None or cache[0] != key:
            with torch.inference_mode(False), torch.no_grad():
                bank = conv_mfma.StyleBank([m.bank_entry(slot) for m, slot in layers], latent.device)
            cache = banks[first_free] = (key, bank)
        lat = latent.detach()
        if not lat.is_contiguous():
            lat = lat.contiguous()
        vecs = cache[1].run(lat)
        return {m: v for (m, _), v in zip(layers, vecs)}

    def _noise_bank(self, batch, device, dtype):
        """Fresh N(0,1) noise images for all layers from ONE generator launch (the reference draws one tensor per
        layer, networks.py:294)."""
        sizes = [2 ** ((i + 5) // 2) for i in range(self.num_layers)]
        flat = torch.empty(batch * sum(r * r for r in sizes), device=device, dtype=dtype).normal_()
        out, off = [], 0
        for r in sizes:
            out.append(flat[off:off + batch * r * r].view(batch, 1, r, r))
            off += batch * r * r
        return out
