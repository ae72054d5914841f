"""torch.library registration of the boundary operators (namespace ``gangealing``): the same HIP entry points as
the Python modules of this package, visible to the dispatcher - ``torch.ops.gangealing.upfirdn2d(...)`` etc. - with
fake (meta) implementations for shape inference / torch.compile tracing and autograd formulas registered through
``torch.library.register_autograd``.  (north_star: "exposed through PyTorch-ROCm custom ops")

    import gangealing_amd.op.library          # registers on import
    y = torch.ops.gangealing.upfirdn2d(x, k, 2, 1, 2, 1)

The module-level functions (op/upfirdn2d.py, op/fused_act.py, splat2d_cuda/functional.py, antialiased_sampling.py)
remain the drop-in API with the reference's signatures; these ops are aliases of the same kernels (all of them
differentiable where the reference's operator is; splat2d's backward raises, as the reference's does)."""
import torch

import importlib

from .. import _lib  # noqa: F401  (the HIP library must be loadable for these ops to run)

# (the package re-exports the FUNCTIONS `upfirdn2d` / `fused_leaky_relu` under the submodule names)
_fa = importlib.import_module(__package__ + '.fused_act')
_up = importlib.import_module(__package__ + '.upfirdn2d')

_lib_def = torch.library.Library('gangealing', 'FRAGMENT')


def _define(schema):
    try:
        _lib_def.define(schema)
    except RuntimeError:           # already defined (re-import)
        pass


# ---- upfirdn2d(input (N,C,H,W), kernel (kh,kw), up, down, pad0, pad1) -------------------------------------------
_define('upfirdn2d(Tensor input, Tensor kernel, int up, int down, int pad0, int pad1) -> Tensor')


def _upfirdn2d_impl(input, kernel, up, down, pad0, pad1):
    return _up._launch(input, kernel, up, up, down, down, pad0, pad1, pad0, pad1)


def _upfirdn2d_fake(input, kernel, up, down, pad0, pad1):
    n, c, h, w = input.shape
    kh, kw = kernel.shape
    oh, ow = _up._out_size(h, w, kh, kw, up, up, down, down, pad0, pad1, pad0, pad1)
    return input.new_empty((n, c, max(oh, 0), max(ow, 0)))


def _upfirdn2d_setup(ctx, inputs, output):
    input, kernel, up, down, pad0, pad1 = inputs
    kh, kw = kernel.shape
    _, _, h, w = input.shape
    oh, ow = output.shape[-2:]
    ctx.save_for_backward(kernel)
    ctx.conf = (up, down, kw - pad0 - 1, w * up - ow * down + pad0 - up + 1, kh - pad0 - 1,
                h * up - oh * down + pad0 - up + 1)


def _upfirdn2d_backward(ctx, grad):
    (kernel,) = ctx.saved_tensors
    up, down, gx0, gx1, gy0, gy1 = ctx.conf
    # the adjoint is the same operator with up <-> down and flipped taps (upfirdn2d.py:113-118); the general
    # (asymmetric) adjoint padding goes through the Function of op/upfirdn2d.py
    g = _up.UpFirDn2d.apply(grad, _up._flipped(kernel), (down, down), (up, up), (gx0, gx1, gy0, gy1))
    return g, None, None, None, None, None


# ---- fused_bias_act(input, bias, negative_slope, scale) ---------------------------------------------------------
_define('fused_leaky_relu(Tensor input, Tensor bias, float negative_slope, float scale) -> Tensor')


def _flr_impl(input, bias, negative_slope, scale):
    return _fa._bias_act(input, bias, None, 3, 0, negative_slope, scale)


def _flr_fake(input, bias, negative_slope, scale):
    return torch.empty_like(input)


def _flr_setup(ctx, inputs, output):
    ctx.save_for_backward(output)
    ctx.conf = (inputs[2], inputs[3])


def _flr_backward(ctx, grad):
    (out,) = ctx.saved_tensors
    gi, gb = _fa.FusedLeakyReLUFunctionBackward.apply(grad, out, *ctx.conf)
    return gi, gb, None, None


# ---- splat2d(input, coordinates, values, sigma, soft_normalize) -------------------------------------------------
_define('splat2d(Tensor input, Tensor coordinates, Tensor values, Tensor sigma, bool soft_normalize) -> Tensor')


def _splat_impl(input, coordinates, values, sigma, soft_normalize):
    from ..splat2d_cuda.functional import Splat2DFunction
    return Splat2DFunction.forward(None, input, coordinates, values, sigma, soft_normalize)


def _splat_fake(input, coordinates, values, sigma, soft_normalize):
    return torch.empty_like(input)


# ---- mipmap_warp(inputs, grid, max_num_levels, min_level, padding_mode, antialias) -> (out, levels) --------------
_define('mipmap_warp(Tensor inputs, Tensor grid, float max_level, float min_level, str padding_mode, bool antialias)'
        ' -> (Tensor, Tensor)')


def _warp_impl(inputs, grid, max_level, min_level, padding_mode, antialias):
    from ..spatial_transformers.antialiased_sampling import _MipmapWarpFn
    with torch.no_grad():
        return _MipmapWarpFn.apply(inputs, grid, max_level, min_level, padding_mode, antialias)


def _warp_setup(ctx, inputs, output):
    img, grid, max_level, min_level, padding_mode, antialias = inputs
    ctx.save_for_backward(img, grid)
    ctx.conf = (max_level, min_level, padding_mode, antialias)


def _warp_backward(ctx, grad_out, _grad_levels):
    """Gradients w.r.t. the image and the sampling grid: the formula of _MipmapWarpFn (the op's forward ran without a
    graph, so the Function is re-applied here - the alias operator pays one extra forward; the module-level MipmapWarp
    is the training path)."""
    from ..spatial_transformers.antialiased_sampling import _MipmapWarpFn
    img, grid = ctx.saved_tensors
    need = ctx.needs_input_grad[:2]
    with torch.enable_grad():
        a = img.detach().requires_grad_(need[0])
        b = grid.detach().requires_grad_(need[1])
        out, _ = _MipmapWarpFn.apply(a, b, *ctx.conf)
        wanted = [t for t, n in ((a, need[0]), (b, need[1])) if n]
        grads = list(torch.autograd.grad(out, wanted, grad_out)) if wanted else []
    g_img = grads.pop(0) if need[0] else None
    g_grid = grads.pop(0) if need[1] else None
    return g_img, g_grid, None, None, None, None


def _splat_setup(ctx, inputs, output):
    pass


def _splat_backward(ctx, grad):
    raise NotImplementedError('splat2d has no backward (as the reference: utils/splat2d_cuda/functional.py:66-68)')


def _warp_fake(inputs, grid, max_level, min_level, padding_mode, antialias):
    n, c = inputs.shape[:2]
    ho, wo = grid.shape[1:3]
    return inputs.new_empty((n, c, ho, wo)), inputs.new_empty((n, ho, wo))


# ---- conv2d / conv_transpose2d (conv2d_gradfix.py:22-75): the implicit-GEMM MFMA convolutions -------------------
# (the autograd formula is conv_mfma.conv2d_backward: data and weight gradients are further HIP launches)
_define('conv2d(Tensor input, Tensor weight, Tensor? bias, int stride, int padding, int groups) -> Tensor')
_define('conv_transpose2d(Tensor input, Tensor weight, Tensor? bias, int stride, int padding, int output_padding, '
        'int groups) -> Tensor')


def _conv2d_impl(input, weight, bias, stride, padding, groups):
    from . import conv_mfma
    with torch.no_grad():
        return conv_mfma.conv2d(input, weight, bias, stride, padding, 1, groups)


def _conv_conf(weight, bias, stride, padding, groups, transposed, output_padding):
    k = weight.shape[-1]
    if transposed:
        cin_g, cout_g = weight.shape[0] // groups, weight.shape[1]
    else:
        cin_g, cout_g = weight.shape[1], weight.shape[0] // groups
    return (stride, padding, groups, transposed, output_padding, bias is not None, cin_g, cout_g, k, 1.0)


def _conv2d_setup(ctx, inputs, output):
    input, weight, bias, stride, padding, groups = inputs
    ctx.save_for_backward(input, weight)
    ctx.conf = _conv_conf(weight, bias, stride, padding, groups, False, 0)


def _convT_setup(ctx, inputs, output):
    input, weight, bias, stride, padding, output_padding, groups = inputs
    ctx.save_for_backward(input, weight)
    ctx.conf = _conv_conf(weight, bias, stride, padding, groups, True, output_padding)


def _conv2d_backward(ctx, grad):
    from . import conv_mfma
    x, weight = ctx.saved_tensors
    dx, dw, db = conv_mfma.conv2d_backward(x.contiguous(), weight.contiguous(), grad, ctx.conf, ctx.needs_input_grad[:3])
    return (dx, dw, db) + (None,) * (4 if ctx.conf[3] else 3)


def _conv2d_fake(input, weight, bias, stride, padding, groups):
    n, _, h, w = input.shape
    k = weight.shape[-1]
    return input.new_empty((n, weight.shape[0], (h + 2 * padding - k) // stride + 1, (w + 2 * padding - k) // stride + 1))


def _convT_impl(input, weight, bias, stride, padding, output_padding, groups):
    from . import conv_mfma
    with torch.no_grad():
        return conv_mfma.conv_transpose2d(input, weight, bias, stride, padding, output_padding, groups)


def _convT_fake(input, weight, bias, stride, padding, output_padding, groups):
    n, _, h, w = input.shape
    k = weight.shape[-1]
    return input.new_empty((n, weight.shape[1] * groups, (h - 1) * stride - 2 * padding + k + output_padding,
                            (w - 1) * stride - 2 * padding + k + output_padding))


def _register():
    for name, impl, fake in (('upfirdn2d', _upfirdn2d_impl, _upfirdn2d_fake), ('fused_leaky_relu', _flr_impl, _flr_fake),
                             ('splat2d', _splat_impl, _splat_fake), ('mipmap_warp', _warp_impl, _warp_fake)):
        try:
            _lib_def.impl(name, impl, 'CUDA')
            torch.library.register_fake(f'gangealing::{name}', fake, lib=_lib_def)
        except RuntimeError:
            pass
    for name, impl, fake in (('conv2d', _conv2d_impl, _conv2d_fake), ('conv_transpose2d', _convT_impl, _convT_fake)):
        try:
            _lib_def.impl(name, impl, 'CUDA')
            torch.library.register_fake(f'gangealing::{name}', fake, lib=_lib_def)
            torch.library.register_autograd(f'gangealing::{name}', _conv2d_backward,
                                            setup_context=_conv2d_setup if name == 'conv2d' else _convT_setup,
                                            lib=_lib_def)
        except RuntimeError:
            pass
    try:
        torch.library.register_autograd('gangealing::upfirdn2d', _upfirdn2d_backward, setup_context=_upfirdn2d_setup,
                                        lib=_lib_def)
        torch.library.register_autograd('gangealing::fused_leaky_relu', _flr_backward, setup_context=_flr_setup,
                                        lib=_lib_def)
        torch.library.register_autograd('gangealing::mipmap_warp', _warp_backward, setup_context=_warp_setup,
                                        lib=_lib_def)
        torch.library.register_autograd('gangealing::splat2d', _splat_backward, setup_context=_splat_setup,
                                        lib=_lib_def)
    except RuntimeError:
        pass


_register()
