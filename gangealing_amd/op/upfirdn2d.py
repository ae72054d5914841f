"""upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)) - same signature and semantics as
models/stylegan2/op/upfirdn2d.py:147-158, running gg_upfirdn2d_f32/f64/f16 (csrc/upfirdn2d.hip).

The gradient of upfirdn2d is upfirdn2d with up<->down, flipped taps and the g_pad of
upfirdn2d.py:113-118, so a single autograd Function whose backward re-applies itself gives
first- and second-order gradients (the reference needs two Function classes, :21-144).
No gradient flows to `kernel` (as in the reference).
"""
import torch
from torch.autograd import Function

from .. import _lib

_SUFFIX = {torch.float32: 'f32', torch.float64: 'f64', torch.float16: 'f16'}


def _out_size(in_h, in_w, kh, kw, up_x, up_y, down_x, down_y, px0, px1, py0, py1):
    out_h = (in_h * up_y + py0 + py1 - kh) // down_y + 1
    out_w = (in_w * up_x + px0 + px1 - kw) // down_x + 1
    return out_h, out_w


def _launch(x, kernel, up_x, up_y, down_x, down_y, px0, px1, py0, py1):
    if x.dtype not in _SUFFIX:
        raise TypeError(f'upfirdn2d: unsupported dtype {x.dtype} (float32 / float64 / float16)')
    n, c, in_h, in_w = x.shape
    kh, kw = kernel.shape
    out_h, out_w = _out_size(in_h, in_w, kh, kw, up_x, up_y, down_x, down_y, px0, px1, py0, py1)
    x = x.contiguous()
    # half tensors: fp32 taps and accumulation inside the kernel
    kernel = kernel.to(dtype=torch.float32 if x.dtype == torch.float16 else x.dtype).contiguous()
    out = torch.empty((n, c, max(out_h, 0), max(out_w, 0)), dtype=x.dtype, device=x.device)
    if out.numel():
        name = f'upfirdn2d<{kh}x{kw},up{up_x},down{down_x}>'
        prof = _profiler(name)
        start = prof.begin() if prof is not None else None
        _lib.call('gg_upfirdn2d_' + _SUFFIX[x.dtype], out, x, kernel, n * c, in_h, in_w, kh, kw,
                  up_x, up_y, down_x, down_y, px0, px1, py0, py1)
        if prof is not None:      # algorithmic bytes: every input and output element once
            prof.end(start, x.element_size() * (x.numel() + out.numel()), name, 'byte')
    return out


def _profiler(name):
    """bench.py's per-kernel timing (conv_mfma.LaunchProfiler with every=True) when it wants launches of `name`."""
    from . import conv_mfma
    prof = conv_mfma.PROFILER
    return prof if (prof is not None and prof.every and (prof.only is None or prof.only == name)) else None


_FLIPPED = {}


def _flipped(kernel):
    """torch.flip(kernel, [0, 1]), cached per (storage, version): the FIR taps are module buffers."""
    if kernel.is_inference():               # made under torch.inference_mode: no version counter to key the cache on
        return torch.flip(kernel, [0, 1]).contiguous()
    key = (kernel.data_ptr(), kernel._version, tuple(kernel.shape), kernel.dtype)
    hit = _FLIPPED.get(key)
    if hit is None:
        if len(_FLIPPED) > 256:
            _FLIPPED.clear()
        with torch.inference_mode(False):      # a cached tensor must be usable by autograd later (see EqualLinear._scaled)
            hit = _FLIPPED[key] = (torch.flip(kernel, [0, 1]).contiguous(), kernel)    # keep `kernel` alive: stable address
    return hit[0]


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
        grad_input = UpFirDn2d.apply(grad_output, _flipped(kernel), down, up, ctx.g_pad)
        return grad_input, None, None, None, None


class UpFirDn2dAdd(Function):
    """upfirdn2d(input) + addend in one kernel (gg_upfirdn2d_add_f32): ToRGB's `out + self.upsample(skip)`
    (networks.py:369-371).  The addend's gradient is the incoming gradient itself."""

    @staticmethod
    def forward(ctx, input, kernel, up, down, pad, addend):
        px0, px1, py0, py1 = pad
        kh, kw = kernel.shape
        n, c, in_h, in_w = input.shape
        out_h, out_w = _out_size(in_h, in_w, kh, kw, up, up, down, down, px0, px1, py0, py1)
        if tuple(addend.shape) != (n, c, out_h, out_w):
            raise ValueError(f'upfirdn2d_add: addend {tuple(addend.shape)} vs output {(n, c, out_h, out_w)}')
        ctx.save_for_backward(kernel)
        ctx.conf = (up, down)
        ctx.g_pad = (kw - px0 - 1, in_w * up - out_w * down + px0 - up + 1,
                     kh - py0 - 1, in_h * up - out_h * down + py0 - up + 1)
        out = torch.empty_like(addend, memory_format=torch.contiguous_format)
        _lib.call('gg_upfirdn2d_add_f32', out, input.contiguous(), kernel.to(input.dtype).contiguous(),
                  addend.contiguous(), n * c, in_h, in_w, kh, kw, up, up, down, down, px0, px1, py0, py1)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        (kernel,) = ctx.saved_tensors
        up, down = ctx.conf
        grad_input = None
        if ctx.needs_input_grad[0]:
            grad_input = UpFirDn2d.apply(grad_output, _flipped(kernel), (down, down), (up, up), ctx.g_pad)
        return grad_input, None, None, None, None, (grad_output if ctx.needs_input_grad[5] else None)


class _BlurDownTap(Function):
    """x -> (x, upfirdn2d(x, kernel, down=2, pad)) as ONE autograd node: ResBlock's input feeds conv1 AND the blurred,
    decimated skip branch (networks.py:383-393).  The backward receives the gradient conv1 produced for x and ADDS the
    adjoint of the decimating blur into it in the same pass (gg_upfirdn2d_add_f32) - instead of writing the adjoint to a
    tensor of its own that autograd then sums with the other branch's (a full-size write + read + add launch less)."""

    @staticmethod
    def forward(ctx, x, kernel, pad):
        x = x.contiguous()
        p0, p1 = pad
        kh, kw = kernel.shape
        _, _, in_h, in_w = x.shape
        out_h, out_w = _out_size(in_h, in_w, kh, kw, 1, 1, 2, 2, p0, p1, p0, p1)
        ctx.save_for_backward(kernel)
        ctx.g_pad = (kw - p0 - 1, in_w - out_w * 2 + p0, kh - p0 - 1, in_h - out_h * 2 + p0)
        ctx.in_hw = (in_h, in_w)
        ctx.set_materialize_grads(False)
        return x, _launch(x, kernel, 1, 1, 2, 2, p0, p1, p0, p1)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_x, g_down):
        (kernel,) = ctx.saved_tensors
        if g_down is None:
            return g_x, None, None
        gx0, gx1, gy0, gy1 = ctx.g_pad
        flipped = _flipped(kernel)
        if g_x is None or g_x.dtype != torch.float32:
            gd = _launch(g_down, flipped, 2, 2, 1, 1, gx0, gx1, gy0, gy1)
            return (gd if g_x is None else g_x + gd), None, None
        g_down, g_x = g_down.contiguous(), g_x.contiguous()
        n, c, h, w = g_down.shape
        in_h, in_w = ctx.in_hw
        if tuple(g_x.shape[-2:]) != (in_h, in_w):
            raise ValueError(f'blur_down_tap backward: gradient {tuple(g_x.shape)} vs input {(in_h, in_w)}')
        out = torch.empty_like(g_x)
        prof = _profiler('upfirdn2d_add<4x4,up2,down1>')
        start = prof.begin() if prof is not None else None
        _lib.call('gg_upfirdn2d_add_f32', out, g_down, flipped.to(g_down.dtype).contiguous(), g_x, n * c, h, w,
                  kernel.shape[0], kernel.shape[1], 2, 2, 1, 1, gx0, gx1, gy0, gy1)
        if prof is not None:
            prof.end(start, 4 * (g_down.numel() + 2 * out.numel()), 'upfirdn2d_add<4x4,up2,down1>', 'byte')
        return out, None, None


def blur_down_tap(x, kernel, pad):
    """-> (x, upfirdn2d(x, kernel, up=1, down=2, pad)); see _BlurDownTap."""
    if x.device.type != 'cuda':
        raise _lib.HipLibraryError('blur_down_tap: HIP tensors only')
    return _BlurDownTap.apply(x, kernel.to(x.dtype), (int(pad[0]), int(pad[1])))


def upfirdn2d_add(input, kernel, addend, up=1, down=1, pad=(0, 0)):
    """upfirdn2d(input, kernel, up, down, pad) + addend (float32)."""
    if input.dtype != torch.float32 or addend.dtype != torch.float32:
        return upfirdn2d(input, kernel, up, down, pad) + addend
    return UpFirDn2dAdd.apply(input, kernel, up, down, (pad[0], pad[1], pad[0], pad[1]), addend)


def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):
    if input.device.type != 'cuda':
        raise _lib.HipLibraryError('upfirdn2d: HIP tensors only (the CPU restatement is oracle/np_ops.upfirdn2d)')
    return UpFirDn2d.apply(input, kernel, (up, up), (down, down), (pad[0], pad[1], pad[0], pad[1]))


class _BlurNoiseAct(Function):
    """lrelu(blur4x4(x) + noise_weight * noise + bias) * scale in one kernel, and its backward (the adjoint blur
    of the leaky-ReLU-masked gradient) in one kernel - the tail of the generator's up-sampling StyledConv
    (networks.py:268-298, 344-350) when noise weight and bias are frozen."""

    @staticmethod
    def forward(ctx, x, kernel, pad, noise, noise_weight, bias, negative_slope, scale):
        x = x.contiguous()
        kernel = kernel.to(x.dtype).contiguous()
        n, c, in_h, in_w = x.shape
        p0, p1 = pad
        out = torch.empty((n, c, in_h + p0 + p1 - 3, in_w + p0 + p1 - 3), dtype=x.dtype, device=x.device)
        from . import conv_mfma
        # the backward needs only the SIGN of `out` (fused_act.py:33-38): when a gradient will be asked for, the kernel
        # also leaves it as one bit per element (in its own 61-column x 16-row tiling, see gg_blur4_fused_bits_f32), and the
        # adjoint blur reads that instead of the fp32 tensor (a third less traffic; bitwise the same gradient).  Not with a diagnostics
        # observer installed: it may edit `out` (decision replay), and the backward must then see the edited tensor.
        bits = None
        if (ctx.needs_input_grad[0] and x.dtype == torch.float32 and 'blur_bits' not in conv_mfma.DISABLED
                and conv_mfma.ACT_OBSERVER is None):
            bits = torch.empty((n * c, blur_bits_words(out.shape[2], out.shape[3])), dtype=torch.int32, device=x.device)
        prof = _profiler('blur4_fused<noise+bias+lrelu>')
        start = prof.begin() if prof is not None else None
        if bits is not None:
            rc = _lib.call('gg_blur4_fused_bits_f32', out, x, kernel, n, c, in_h, in_w, p0, p1, p0, p1, noise.contiguous(),
                           noise_weight.contiguous(), bias.contiguous(), bits, negative_slope, scale,
                           allow=(_lib.NOT_SERVED,))
            if rc != 0:
                bits = None
        if bits is None:
            _lib.call('gg_blur4_fused_f32', out, x, kernel, n, c, in_h, in_w, p0, p1, p0, p1, noise.contiguous(),
                      noise_weight.contiguous(), bias.contiguous(), None, negative_slope, scale)
        if prof is not None:
            prof.end(start, 4 * (x.numel() + out.numel() + noise.numel()), 'blur4_fused<noise+bias+lrelu>', 'byte')
        conv_mfma.observe_activation('blur_noise_act', out)
        ctx.has_bits = bits is not None
        ctx.save_for_backward(kernel, bits if bits is not None else out)
        # adjoint padding (reference upfirdn2d.py:113-118 with up = down = 1)
        ctx.g_pad = (4 - p0 - 1, in_w - out.shape[3] + p0)
        ctx.conf = (negative_slope, scale)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        kernel, out = ctx.saved_tensors
        negative_slope, scale = ctx.conf
        grad_output = grad_output.contiguous()
        n, c, h, w = grad_output.shape
        g0, g1 = ctx.g_pad
        dx = torch.empty((n, c, h + g0 + g1 - 3, w + g0 + g1 - 3), dtype=grad_output.dtype, device=grad_output.device)
        prof = _profiler('blur4_fused<lrelu mask>')
        start = prof.begin() if prof is not None else None
        if ctx.has_bits:
            _lib.call('gg_blur4_fused_bits_f32', dx, grad_output, _flipped(kernel), n, c, h, w,
                      g0, g1, g0, g1, None, None, None, out, negative_slope, scale)
        else:
            _lib.call('gg_blur4_fused_f32', dx, grad_output, _flipped(kernel), n, c, h, w,
                      g0, g1, g0, g1, None, None, None, out, negative_slope, scale)
        if prof is not None:
            prof.end(start, 4 * ((1 if ctx.has_bits else 2) * grad_output.numel() + dx.numel()) + (out.numel() * 4 if ctx.has_bits else 0),
                     'blur4_fused<lrelu mask>', 'byte')
        return dx, None, None, None, None, None, None, None


def blur_bits_words(h, w):
    """uint32 words per plane of the blur tail's sign plane for an (h, w) output (gg_blur4_bits_words)."""
    return 2 * ((w + 60) // 61) * ((h + 15) // 16) * 16


def blur_bits_unpack(bits, h, w):
    """(planes, blur_bits_words(h, w)) int32 -> (planes, h, w) bool (tests)."""
    planes = bits.shape[0]
    strips, chunks = (w + 60) // 61, (h + 15) // 16
    words = (bits.to(torch.int64) & 0xFFFFFFFF).reshape(planes, chunks, strips, 16, 2)
    sh = torch.arange(32, device=bits.device, dtype=torch.int64)
    b = ((words.unsqueeze(-1) >> sh) & 1).reshape(planes, chunks, strips, 16, 64)[..., :61]     # [p][cy][sx][row][col]
    return b.permute(0, 1, 3, 2, 4).reshape(planes, chunks * 16, strips * 61)[:, :h, :w].bool()


def blur_noise_act_ok(x, kernel, pad):
    """Shapes the fused 4x4-blur kernel takes (the 64x64-tile kernel of csrc/upfirdn2d.hip)."""
    return (x.dtype == torch.float32 and tuple(kernel.shape) == (4, 4) and
            x.shape[-2] + pad[0] + pad[1] - 3 >= 24 and x.shape[-1] + pad[0] + pad[1] - 3 >= 24)


def blur_noise_act(x, kernel, pad, noise, noise_weight, bias, negative_slope=0.2, scale=2 ** 0.5):
    if noise_weight.requires_grad or bias.requires_grad:
        raise NotImplementedError('blur_noise_act: noise weight / bias gradients are not produced (frozen generator)')
    return _BlurNoiseAct.apply(x, kernel, (int(pad[0]), int(pad[1])), noise, noise_weight, bias, negative_slope, scale)
