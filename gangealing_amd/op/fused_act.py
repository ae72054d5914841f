"""FusedLeakyReLU / fused_leaky_relu - same names, signatures and parameter layout as
models/stylegan2/op/fused_act.py:74-97, running csrc/fused_bias_act.hip.

Differences from the reference that do not change results: the backward computes grad_input and
the bias gradient in ONE kernel (gg_fused_lrelu_bwd); the reference launches the activation
kernel and then a separate torch reduction (:27-38).  As in the reference the saved tensor is the
OUTPUT (sign reference), and second-order gradients are supported (:43-49).
"""
import torch
from torch import nn
from torch.autograd import Function

from .. import _lib

_SUFFIX = {torch.float32: 'f32', torch.float64: 'f64', torch.float16: 'f16'}


def _bias_act(x, bias, ref, act, grad, alpha, scale):
    """Mirror of fused.fused_bias_act(input, bias, refer, act, grad, alpha, scale)
    (fused_bias_act.cpp:11-21); empty tensors / None mean "absent"."""
    if x.dtype not in _SUFFIX:
        raise TypeError(f'fused_bias_act: unsupported dtype {x.dtype}')
    x = x.contiguous()
    bias = None if bias is None or bias.numel() == 0 else bias.to(x.dtype).contiguous()
    ref = None if ref is None or ref.numel() == 0 else ref.contiguous()
    out = torch.empty_like(x)
    step_b = 1
    for d in x.shape[2:]:
        step_b *= d
    _lib.call('gg_fused_bias_act_' + _SUFFIX[x.dtype], out, x, bias, ref, act, grad, alpha, scale,
              x.numel(), step_b, 0 if bias is None else bias.numel())
    return out


def _observe(site, out):
    """Diagnostics hook of the decision-replay tests (conv_mfma.ACT_OBSERVER); nothing is installed otherwise."""
    from . import conv_mfma
    if conv_mfma.ACT_OBSERVER is not None:
        conv_mfma.ACT_OBSERVER(site, out)


class FusedLeakyReLUFunctionBackward(Function):
    @staticmethod
    def forward(ctx, grad_output, out, negative_slope, scale):
        ctx.save_for_backward(out)
        ctx.negative_slope = negative_slope
        ctx.scale = scale
        grad_output = grad_output.contiguous()
        n, c = out.shape[0], out.shape[1]
        hw = out.numel() // max(n * c, 1)
        grad_input = torch.empty_like(grad_output)
        # half tensors: the kernel accumulates the bias gradient in an fp32 buffer
        grad_bias = torch.empty(c, dtype=torch.float32 if out.dtype == torch.float16 else out.dtype, device=out.device)
        _lib.call('gg_fused_lrelu_bwd_' + _SUFFIX[out.dtype], grad_input, grad_bias, grad_output, out,
                  negative_slope, scale, n, c, hw)
        return grad_input, grad_bias.to(out.dtype)

    @staticmethod
    def backward(ctx, gradgrad_input, gradgrad_bias):
        (out,) = ctx.saved_tensors
        gradgrad_out = _bias_act(gradgrad_input, gradgrad_bias, out, 3, 1, ctx.negative_slope, ctx.scale)
        return gradgrad_out, None, None, None


class FusedLeakyReLUFunction(Function):
    @staticmethod
    def forward(ctx, input, bias, negative_slope, scale):
        out = _bias_act(input, bias, None, 3, 0, negative_slope, scale)
        _observe('fused_leaky_relu', out)
        ctx.save_for_backward(out)
        ctx.negative_slope = negative_slope
        ctx.scale = scale
        ctx.bias_ref = bias          # the parameter itself: looked up in the gradient-slot registry in backward
        return out

    @staticmethod
    def backward(ctx, grad_output):
        (out,) = ctx.saved_tensors
        if not ctx.needs_input_grad[1] and not torch.is_grad_enabled():
            # frozen bias, first-order only: skip the bias reduction (atomics) altogether
            grad_output = grad_output.contiguous()
            n, c = out.shape[0], out.shape[1]
            grad_input = torch.empty_like(grad_output)
            _lib.call('gg_fused_lrelu_bwd_' + _SUFFIX[out.dtype], grad_input, None, grad_output, out,
                      ctx.negative_slope, ctx.scale, n, c, out.numel() // max(n * c, 1))
            return grad_input, None, None, None
        if ctx.needs_input_grad[1] and not torch.is_grad_enabled() and out.dtype == torch.float32:
            # inside the trainer's backward (conv_mfma.grad_slots): the bias gradient is added straight into the
            # parameter's slot of the flat gradient arena - no temporary and no AccumulateGrad add
            from . import conv_mfma
            slot = conv_mfma._slot_for(ctx.bias_ref) if ctx.bias_ref is not None else None
            if slot is not None:
                grad_output = grad_output.contiguous()
                n, c = out.shape[0], out.shape[1]
                grad_input = torch.empty_like(grad_output)
                _lib.call('gg_fused_lrelu_bwd_acc_f32', grad_input, slot, grad_output, out, ctx.negative_slope,
                          ctx.scale, n, c, out.numel() // max(n * c, 1), 1)
                return grad_input, None, None, None
        grad_input, grad_bias = FusedLeakyReLUFunctionBackward.apply(grad_output, out, ctx.negative_slope, ctx.scale)
        return grad_input, grad_bias, None, None


class NoiseBiasLeakyReLUFunction(Function):
    """lrelu(x + noise_weight * noise + bias) * scale in one kernel (StyledConv's NoiseInjection +
    FusedLeakyReLU, networks.py:291-298,344-350).  Gradients: input always; bias / noise weight only if
    they require grad (they do not on the frozen generator)."""

    @staticmethod
    def forward(ctx, input, noise, noise_weight, bias, negative_slope, scale):
        input = input.contiguous()
        n, c = input.shape[0], input.shape[1]
        hw = input.numel() // max(n * c, 1)
        out = torch.empty_like(input)
        _lib.call('gg_noise_bias_act_f32', out, input, noise.contiguous(), noise_weight.contiguous(),
                  bias.contiguous(), negative_slope, scale, n, c, hw)
        _observe('noise_bias_leaky_relu', out)
        ctx.save_for_backward(out, noise)
        ctx.conf = (negative_slope, scale)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        out, noise = ctx.saved_tensors
        negative_slope, scale = ctx.conf
        grad_output = grad_output.contiguous()
        n, c = out.shape[0], out.shape[1]
        hw = out.numel() // max(n * c, 1)
        need_bias, need_nw = ctx.needs_input_grad[3], ctx.needs_input_grad[2]
        grad_input = torch.empty_like(grad_output)
        grad_bias = torch.empty(c, dtype=out.dtype, device=out.device) if need_bias else None
        _lib.call('gg_fused_lrelu_bwd_f32', grad_input, grad_bias, grad_output, out, negative_slope, scale, n, c, hw)
        grad_nw = (grad_input.sum(dim=1, keepdim=True) * noise).sum().reshape(1) if need_nw else None
        return grad_input, None, grad_nw, grad_bias, None, None


def noise_bias_leaky_relu(input, noise, noise_weight, bias, negative_slope=0.2, scale=2 ** 0.5):
    return NoiseBiasLeakyReLUFunction.apply(input, noise, noise_weight, bias, negative_slope, scale)


class FusedLeakyReLU(nn.Module):
    def __init__(self, channel, negative_slope=0.2, scale=2 ** 0.5):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel))
        self.negative_slope = negative_slope
        self.scale = scale

    def forward(self, input):
        return fused_leaky_relu(input, self.bias.type(input.dtype), self.negative_slope, self.scale)


def fused_leaky_relu(input, bias, negative_slope=0.2, scale=2 ** 0.5):
    if input.device.type != 'cuda':
        raise _lib.HipLibraryError('fused_leaky_relu: HIP tensors only (CPU restatement: oracle/np_ops.fused_leaky_relu)')
    return FusedLeakyReLUFunction.apply(input, bias, negative_slope, scale)
