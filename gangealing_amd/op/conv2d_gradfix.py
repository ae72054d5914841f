"""Drop-in for models/stylegan2/op/conv2d_gradfix.py: same module-level API
(conv2d, conv_transpose2d, no_weight_gradients, enabled, weight_gradients_disabled).

In the reference these are pass-throughs to F.conv2d / F.conv_transpose2d on every supported
torch version (could_use_op is True only for torch 1.7/1.8, conv2d_gradfix.py:78-92).  Here they
run the hand-written implicit-GEMM MFMA convolution (csrc/conv_mfma.hip), including the
per-sample grouped form that ModulatedConv2d issues (groups = batch, networks.py:263,272,278).
"""
import contextlib

from . import conv_mfma

enabled = True
weight_gradients_disabled = False


@contextlib.contextmanager
def no_weight_gradients():
    global weight_gradients_disabled
    old = weight_gradients_disabled
    weight_gradients_disabled = True
    try:
        yield
    finally:                      # (restored when the body raises too; the reference, conv2d_gradfix.py:13-19, is not)
        weight_gradients_disabled = old


def conv2d(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
    if weight_gradients_disabled:
        weight = weight.detach()
    return conv_mfma.conv2d(input, weight, bias=bias, stride=stride, padding=padding, dilation=dilation,
                            groups=groups)


def conv_transpose2d(input, weight, bias=None, stride=1, padding=0, output_padding=0, groups=1, dilation=1):
    if weight_gradients_disabled:
        weight = weight.detach()
    return conv_mfma.conv_transpose2d(input, weight, bias=bias, stride=stride, padding=padding,
                                      output_padding=output_padding, groups=groups, dilation=dilation)
