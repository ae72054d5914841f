"""Autograd wrappers around the implicit-GEMM MFMA convolution (csrc/conv_mfma.hip).

conv2d / conv_transpose2d here are what op/conv2d_gradfix.py exposes under the reference's
names; `modulated_conv2d` is the shared-weight form of ModulatedConv2d.forward
(models/stylegan2/networks.py:233-282) used by our generator.
"""
import os

import torch
from torch.autograd import Function

from .. import _lib


class LaunchProfiler:
    """Optional per-launch timing with HIP events recorded on the stream the kernel is launched on (torch's current
    stream).  Used by bench.py for the roofline entry; None (the default) costs nothing.

    every=False: only the launches of the dominant kernel instantiation of config C2 (the predicate in conv_forward
    mirrors the library's dispatch and was checked against rocprofv3 traces).  every=True: every convolution and FIR
    launch, keyed by the kernel the library reports it launched (gg_last_conv_kernel) - how bench.py finds out which
    kernel dominates another workload (C4: the transposed tile / the 513^2 blur; C5: the STN / VGG tiles).  only=name
    keeps the records of that kernel alone (the timed region of those workloads)."""

    def __init__(self, every=False, only=None, names=None):
        self.records = []          # (start_event, end_event, work, kernel name, unit of work)
        self.every = bool(every) or only is not None
        self.only = only
        # launch signature -> kernel name, learned while every launch is timed (the survey); with only=name the timed
        # region then records events for the matching launches alone - an event pair around EVERY launch costs host time
        # that a launch-bound step (small batches) cannot hide
        self.names = {} if names is None else names

    def wants(self, sig):
        """Record this launch?  (survey: yes; only=name: when the survey saw this signature run that kernel)"""
        return self.only is None or self.names.get(sig) == self.only

    def begin(self):
        start = torch.cuda.Event(enable_timing=True)
        start.record()
        return start

    def end(self, start, work, name='dominant', unit='flop'):
        if self.only is not None and name != self.only:
            return
        end = torch.cuda.Event(enable_timing=True)
        end.record()
        self.records.append((start, end, float(work), name, unit))

    def summary(self):
        torch.cuda.synchronize()
        ms = [r[0].elapsed_time(r[1]) for r in self.records]
        return dict(launches=len(ms), total_ms=sum(ms), total_flops=float(sum(r[2] for r in self.records)))

    def by_kernel(self):
        """{kernel name: dict(launches, ms, work, unit)} over the recorded launches."""
        torch.cuda.synchronize()
        out = {}
        for start, end, work, name, unit in self.records:
            ent = out.setdefault(name, dict(launches=0, ms=0.0, work=0.0, unit=unit))
            ent['launches'] += 1
            ent['ms'] += start.elapsed_time(end)
            ent['work'] += work
        return out


def last_amax_written():
    return bool(_lib.load().gg_last_amax_written())


def last_conv_kernel():
    lib = _lib.load()
    return lib.gg_last_conv_kernel().decode() if hasattr(lib, 'gg_last_conv_kernel') else 'unknown'


PROFILER = None


def _prof_begin(sig):
    """(profiler, start event) when the installed profiler times launches of signature `sig`, else (None, None)."""
    p = PROFILER
    if p is None or not p.every or not p.wants(sig):
        return None, None
    return p, p.begin()


def _prof_end(p, start, sig, work, name, unit='flop'):
    if p is not None:
        p.names[sig] = name
        p.end(start, work, name, unit)

# Arithmetic of the implicit-GEMM convolutions:
#   'fp32'   v_mfma_f32_32x32x2_f32 - exact fp32 products (the parity mode, default)
#   'bf16x3' / 'bf16x6'  bf16 matrix pipe with 2 / 3 bf16 limbs per fp32 operand (3 / 6 MFMAs per tile step,
#            fp32 accumulation): ~2^-16 / ~2^-23 relative error per product.  Layers whose input-channel
#            count is not a multiple of 32 (3-channel stems) or with <= 32 outputs stay on the fp32 kernel.
#   'bf16'   one bf16 limb per operand (one MFMA per tile step, fp32 accumulation, fp32 activations in HBM): plain
#            bf16 matrix arithmetic, ~2^-9 relative error per product - the arithmetic BASELINE.json's benchmark
#            configuration names ("bf16").  Not a parity mode: activations agree with the reference to ~1e-2.
#   'fp16x3' two 16-bit limbs per operand and 3 MFMA products like bf16x3, but the limbs of the FORWARD convolutions are
#            IEEE binary16 (11-bit mantissa: two limbs carry 22 bits, ~2^-21 per product - fp32-class activations at the
#            bf16x3 price; binary16's narrow exponent is safe there because activations are O(1) and the weights are
#            pre-scaled in the pack), while every gradient convolution keeps bf16 limbs (fp32's exponent range: gradients
#            of any magnitude).  CPU emulation on the reference STN (scripts/study_bf16x3_stn.py): warped-output error
#            2.6e-5 (bf16x3: 3.3e-4, bf16x6: 2.5e-5, the reference's own fp32 vs fp64: 1.7e-5).
import os as _os
PRECISION = _os.environ.get('GANGEALING_CONV_PRECISION', 'fp32')
_LIMBS = {'fp32': 0, 'bf16': 1, 'bf16x3': 2, 'bf16x6': 3, 'fp16x3': 2}
_F16_FORWARD = frozenset(['fp16x3'])
# Round 4: the DATA-gradient convolutions of fp16x3 run on binary16 limbs too.  What kept them on bf16 limbs in round 3
# was binary16's exponent range (gradients sit at 1e-8 .. 1e-2); the kernels now carry a block exponent per tile that is
# taken from the data (csrc/conv_mfma.hip, BlockExp), so any magnitude is staged at full limb precision.  Weight
# gradients keep bf16 limbs.  GANGEALING_F16_GRADS=0 restores round 3's arithmetic (A/B).
_F16_GRAD = frozenset(['fp16x3']) if _os.environ.get('GANGEALING_F16_GRADS', '1') != '0' else frozenset()


# ... except on the generic (re-gathering) kernel: it is instruction-issue bound (~350 VALU / scalar instructions per 24
# MFMAs), so the block-exponent bookkeeping per gathered slab costs it 9-12 % (profiles/r04_*_ab*: 0.738 -> 0.828 ms
# on the 257^2 -> 128^2 data gradient) where the patch / transposed tiles hide it under their MFMAs.  Its gradient
# launches (stride-2 correlations = data gradients of the generator's up-convolutions, 1x1, images narrower than 16)
# therefore stay on bf16 limbs unless GANGEALING_F16_GRADS=all.
_F16_GRAD_GENERIC = _os.environ.get('GANGEALING_F16_GRADS', '1') == 'all'


def limb_code(grad=False, generic=False):
    """Format code of the split-precision entry points for the current mode: limb count, + 16 when the limbs are
    binary16 (the fp16x3 mode; `grad` = the launch is a data-gradient convolution, `generic` = a shape the library serves
    with the generic re-gathering kernel)."""
    limbs = _LIMBS[PRECISION]
    if grad and generic and not _F16_GRAD_GENERIC:
        return limbs
    if PRECISION in (_F16_GRAD if grad else _F16_FORWARD):
        # + 32 (round 6): the operand is a gradient - the block exponent's unscaled band starts at 2^5, not 2^-3
        return limbs | 16 | (GRAD_OPERAND if grad else 0)
    return limbs


GRAD_OPERAND = 32


_S2_PATCH = _os.environ.get('GG_S2_PATCH', '1') != '0'      # (the library reads the same switch)


def _generic_shape(k, stride, pad, mode, w, h=None):
    """True when csrc/conv_mfma.hip::conv_dispatch sends this split-precision launch to conv_split_kernel (neither the
    3x3 / stride-1 patch tile, the transposed 3x3 / stride-2 tile nor the 3x3 / stride-2 patch tile of
    conv_s2_patch.hip serves it).  A performance hint only: either limb format is valid on every kernel."""
    pow2 = (w & (w - 1)) == 0
    if mode == 1:
        return not (k == 3 and pad <= 1 and w >= 4 and pow2)
    if k == 3 and stride == 2 and pad == 0 and _S2_PATCH:
        ow, oh = (w - 3) // 2 + 1, ((w if h is None else h) - 3) // 2 + 1
        return not (ow >= 32 and ow % 32 == 0 and oh >= 4 and oh % 4 == 0)
    return not (k == 3 and stride == 1 and pad == 1 and w >= 16 and pow2)


def set_precision(mode):
    global PRECISION
    if mode not in _LIMBS:
        raise ValueError(f'unknown conv precision {mode!r}; choose from {sorted(_LIMBS)}')
    PRECISION = mode


# The similarity STN regresses FOUR numbers (rotation, scale, shift) that place every output sample: a relative error
# of 4e-5 in them - what two bf16 limbs leave after the 14 convolutions of its trunk - moves the samples near the image
# corners by ~1e-3 pixel, i.e. 3e-4 on a textured image of amplitude 2, above north_star's 1e-4.  Every layer of the
# trunk contributes about equally (scripts/study_bf16x3_stn.py: an emulation of the limb arithmetic on the reference
# STN; upgrading only the <= 8^2 or <= 32^2 layers leaves 1.6e-4 / 1.2e-4), so in the bf16x3 mode the FORWARD pass of
# that trunk (12.6 GFLOP per image, 3 % of the step's convolution work) runs with three limbs (fp32-class products:
# 4.8e-5 in the same study); its backward stays on two limbs, like everything else.  GANGEALING_SIM_PRECISION overrides.
REGRESSION_PRECISION = {'bf16x3': _os.environ.get('GANGEALING_SIM_PRECISION', 'bf16x6')}


class forward_precision:
    """Context manager: the convolutions issued inside run in `mode` (None: unchanged).  Backward passes read the
    global mode when THEY run, so this only affects forward launches."""

    def __init__(self, mode):
        if mode is not None and mode not in _LIMBS:
            raise ValueError(f'unknown conv precision {mode!r}')
        self.mode = mode

    def __enter__(self):
        global PRECISION
        self.prev = PRECISION
        if self.mode is not None:
            PRECISION = self.mode
        return self

    def __exit__(self, *exc):
        global PRECISION
        PRECISION = self.prev
        return False


def regression_precision():
    """forward_precision for the similarity STN's regression trunk under the current global mode."""
    return forward_precision(REGRESSION_PRECISION.get(PRECISION))


class PackedWeight:
    """GEMM-layout view(s) of one convolution weight, built lazily per arithmetic mode.
    (cout_g, cin_g) are those OF THE CONVOLUTION BEING RUN (reduction channels = cin_g)."""

    def __init__(self, weight, groups, cout_g, cin_g, k, transpose_io, flip, scale=1.0):
        self.weight, self.groups, self.cout_g, self.cin_g, self.k = weight, groups, cout_g, cin_g, k
        self.transpose_io, self.flip, self.scale = int(transpose_io), int(flip), float(scale)
        self._fp32 = None
        self._split = {}

    def fp32(self):
        if self._fp32 is None:
            self._fp32 = pack_weight(self.weight, self.groups, self.cout_g, self.cin_g, self.k, self.transpose_io,
                                     self.flip, self.scale)
        return self._fp32

    def split_ok(self):
        return self.cin_g % 32 == 0 and self.cout_g > 32

    def split(self, limbs):
        """limbs: format code (limb_code): 1 | 2 | 3 bf16 limbs, 18 = two binary16 limbs (pre-scaled weights); the
        gradient-operand bit (32) of a launch code does not concern the pack."""
        limbs &= ~GRAD_OPERAND
        if limbs not in self._split:
            n = self.groups * self.cout_g * self.cin_g * self.k * self.k
            buf = torch.empty((limbs & 15, n), dtype=torch.int16, device=self.weight.device)
            _lib.call('gg_conv_pack_weight_split', buf, self.weight.contiguous(), self.groups, self.cout_g,
                      self.cin_g, self.k, self.k, self.transpose_io, self.flip, self.scale, limbs)
            self._split[limbs] = (buf, n)
        return self._split[limbs]


def _pair_eq(v, name):
    if isinstance(v, (tuple, list)):
        if len(v) != 2 or v[0] != v[1]:
            raise NotImplementedError(f'conv_mfma: {name} must be square, got {v}')
        return int(v[0])
    return int(v)


_FROZEN_PACKS = {}


# Trainable weights: None = pack on every use.  A trainer that updates its parameters in place once per step can
# switch the registry on (enable_pack_registry) and call repack_trainable() after the optimizer: every pack that was
# used so far is then rebuilt by ONE launch (gg_conv_pack_weights_many) instead of one launch per layer and use.
# Entries are validated against the weight's version counter, so any other in-place update falls back to packing.
class _PackRegistry:
    def __init__(self):
        self.entries = {}          # key -> [PackedWeight, weight version the packs correspond to]
        self.signature = None      # which layouts the job table covers
        self.jobs = None
        self.njobs = 0


TRAINABLE_PACKS = None


def enable_pack_registry(on=True):
    global TRAINABLE_PACKS
    TRAINABLE_PACKS = _PackRegistry() if on else None


def repack_trainable():
    """Rebuild every registered pack from the current parameter values (one launch)."""
    reg = TRAINABLE_PACKS
    if reg is None or not reg.entries or 'pack_registry' in DISABLED:
        return
    signature = tuple((key, ent[0]._fp32 is not None, tuple(sorted(ent[0]._split))) for key, ent in reg.entries.items())
    if signature != reg.signature:
        import numpy as np
        dt = np.dtype([('dst', '<u8'), ('src', '<u8'), ('total', '<i8'), ('limb_stride', '<i8'), ('cout_g', '<i4'),
                       ('cin_g', '<i4'), ('kh', '<i4'), ('kw', '<i4'), ('transpose_io', '<i4'), ('flip', '<i4'),
                       ('limbs', '<i4'), ('scale', '<f4')])
        assert dt.itemsize == 64
        rows, device = [], None
        for ent in reg.entries.values():
            pw = ent[0]
            device = pw.weight.device
            total = pw.groups * pw.cout_g * pw.cin_g * pw.k * pw.k
            tail = (pw.cout_g, pw.cin_g, pw.k, pw.k, pw.transpose_io, pw.flip)
            if pw._fp32 is not None:
                rows.append((pw._fp32.data_ptr(), pw.weight.data_ptr(), total, 0) + tail + (0, pw.scale))
            for limbs, (buf, n) in pw._split.items():
                rows.append((buf.data_ptr(), pw.weight.data_ptr(), total, n) + tail + (limbs, pw.scale))
        reg.njobs = len(rows)
        reg.jobs = None
        if rows:
            arr = np.array(rows, dtype=dt)
            reg.jobs = torch.from_numpy(arr.view(np.uint8).reshape(-1).copy()).to(device)
        reg.signature = signature
    if reg.jobs is not None:
        _lib.call('gg_conv_pack_weights_many', reg.jobs, reg.njobs)
    for ent in reg.entries.values():
        ent[1] = ent[0].weight._version


def mark_trainable_packs_current():
    """After a captured graph replayed the optimizer + re-pack launches: the registered packs hold the current
    parameter values again; record the parameters' (host-bumped) version counters."""
    reg = TRAINABLE_PACKS
    if reg is not None:
        for ent in reg.entries.values():
            ent[1] = ent[0].weight._version


def packed(weight, groups, cout_g, cin_g, k, transpose_io, flip, scale=1.0):
    """PackedWeight for `weight`; weights that do not require grad (frozen VGG / generator) keep their
    packs across steps, keyed by storage + version so an in-place update invalidates them."""
    if groups > 1 and not isinstance(weight, torch.nn.Parameter):
        # per-sample weights of the reference-form modulated convolution (groups = batch): a fresh (N*Cout, Cin, k, k)
        # tensor per call - nothing to cache (a cache entry would pin 151 MB per layer and pass)
        return PackedWeight(weight.detach(), groups, cout_g, cin_g, k, transpose_io, flip, scale)
    if weight.requires_grad:
        reg = TRAINABLE_PACKS
        if reg is None or 'pack_registry' in DISABLED or not weight.is_leaf:
            return PackedWeight(weight, groups, cout_g, cin_g, k, transpose_io, flip, scale)
        key = (weight.data_ptr(), groups, cout_g, cin_g, k, int(transpose_io), int(flip), float(scale))
        ent = reg.entries.get(key)
        if ent is None or ent[1] != weight._version:
            ent = reg.entries[key] = [PackedWeight(weight.detach(), groups, cout_g, cin_g, k, transpose_io, flip, scale),
                                      weight._version]
        return ent[0]
    if weight.is_inference():
        # a tensor made under torch.inference_mode (the reference's training visuals, training_vis.py:18-19, compute
        # `scale * weight * style` there): no version counter to key a cache on, and nothing to keep - pack per call
        return PackedWeight(weight, groups, cout_g, cin_g, k, transpose_io, flip, scale)
    key = (weight.data_ptr(), weight._version, groups, cout_g, cin_g, k, int(transpose_io), int(flip), float(scale))
    pw = _FROZEN_PACKS.get(key)
    if pw is None:
        if len(_FROZEN_PACKS) > 512:
            _FROZEN_PACKS.clear()
        pw = _FROZEN_PACKS[key] = PackedWeight(weight.detach(), groups, cout_g, cin_g, k, transpose_io, flip, scale)
    return pw


def pack_weight(weight, groups, cout_g, cin_g, k, transpose_io, flip, scale=1.0):
    """-> wmat (groups, cin_g*k*k, cout_g): GEMM layout consumed by gg_conv2d_f32.  `cin_g` is the
    reduction-channel count and `cout_g` the output-channel count OF THE CONVOLUTION BEING RUN."""
    wmat = torch.empty((groups, cin_g * k * k, cout_g), dtype=torch.float32, device=weight.device)
    _lib.call('gg_conv_pack_weight_f32', wmat, weight.contiguous(), groups, cout_g, cin_g, k, k,
              int(transpose_io), int(flip), scale)
    return wmat


def conv_forward(x, wmat, batch, groups, cin_g, cout_g, k, stride, pad, mode, in_scale=None, out_scale=None,
                 bias=None, out_hw=None, act=None, grad=False, want_sign_bits=None, amax_out=None, prelimb=None,
                 residual=None):
    """act = (noise (N,1,OH,OW), noise_weight (1,), act_bias (Cout,), alpha, gain): the StyledConv tail
    lrelu(y + noise_weight*noise + act_bias)*gain fused behind a 3x3/stride-1/pad-1 convolution
    (gg_modconv3x3_act_f32).  grad: this launch is a gradient convolution (data gradient): bf16 limbs in every
    split-precision mode.  want_sign_bits (with act; True / False, not None): the call returns (y, bits) - bits = the
    1-bit sign plane of y (int32 (N, OH*OW, Cout/32), gg_modconv3x3_act_bits_f32) for the layer's masked data gradient when
    asked for with True, None when not asked for or when the kernel that served the launch does not write it.
    amax_out (with act; zeroed float32 (N,)): receives max |y[n]| when the launch's own epilogue wrote y
    (gg_modconv3x3_act_amax_f32; ask last_amax_written()).  prelimb = (xlimb, xexp) (transposed 3x3 / stride 2 only): the
    operand in limb form with the style already applied (torgb_limb) - x / in_scale are then the fallback if the limb-form
    tile does not serve the shape."""
    sign_bits = None
    h, w = x.shape[-2], x.shape[-1]
    if mode == 0:
        oh, ow = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    else:
        oh, ow = (h - 1) * stride - 2 * pad + k, (w - 1) * stride - 2 * pad + k
        if out_hw is not None:
            oh, ow = out_hw
    y = torch.empty((batch, groups * cout_g, oh, ow), dtype=torch.float32, device=x.device)
    if y.numel():
        code = limb_code(grad, grad and _generic_shape(k, stride, pad, mode, w, h))
        limbs = code & 15
        pack_code = code & ~GRAD_OPERAND
        use_split = limbs > 0 and isinstance(wmat, PackedWeight) and wmat.split_ok()
        prof = sig = None
        if PROFILER is not None and PROFILER.every:
            sig = (k, stride, pad, mode, batch, groups, cin_g, cout_g, h, w, in_scale is None, act is None, code)
            prof = PROFILER if PROFILER.wants(sig) else None
        if prof is None and PROFILER is not None and not PROFILER.every and k == 3 and mode == 0 and cout_g > 64:
            if use_split:
                # exactly the launches that run conv3x3_patch_kernel<2, true, 256> (csrc/conv_mfma.hip: patch_geometry
                # + the 256-pixel-tile rule): the generator's style-scaled 3x3 stride-1 layers that fill the chip
                tiles256 = (batch * oh * ow + 255) // 256 * ((cout_g + 127) // 128) * groups
                pow2 = ow >= 16 and (ow & (ow - 1)) == 0
                if (limbs in (1, 2) and stride == 1 and pad == 1 and in_scale is not None and pow2 and tiles256 >= 512
                        and cin_g > 64 and oh % (256 // min(ow, 64)) == 0):
                    prof = PROFILER
                # bf16x6: the same layers run conv3x3_patch_kernel<3, true, 128>
                if (limbs == 3 and stride == 1 and pad == 1 and in_scale is not None and pow2 and tiles256 >= 512
                        and oh % (128 // min(ow, 64)) == 0):
                    prof = PROFILER
            elif limbs == 0 and cout_g > 64 and batch * oh * ow >= 4096:
                prof = PROFILER          # fp32 mode: conv_igemm_kernel<3,0,2,2,2,2,*> launches without split-K
        if prof is not None:
            start = prof.begin()
        if residual is not None and (act is not None or tuple(residual.shape) != tuple(y.shape)
                                     or residual.dtype != torch.float32):
            raise NotImplementedError('conv_forward: residual = a float32 tensor of the output shape, without activation')
        if act is not None and (k, stride, pad, mode) != (3, 1, 1, 0):
            # any other geometry (ResBlock's 3x3 / stride-2 convolution): bias + leaky ReLU inside the library when the
            # split-precision kernels serve the launch (gg_conv2d_split_act_f32), else convolution + activation pass
            noise, noise_weight, act_bias, alpha, gain = act
            if noise is not None or bias is not None or want_sign_bits is not None:
                raise NotImplementedError('conv_forward: noise / epilogue bias / sign plane need the 3x3 stride-1 layer')
            rc = _lib.NOT_SERVED
            if use_split and (oh * ow) % 4 == 0:
                wbuf, stride_l = wmat.split(code)
                rc = _lib.call('gg_conv2d_split_act_f32', y, x, wbuf, stride_l, code, in_scale, out_scale, act_bias, alpha,
                               gain, batch, groups, cin_g, cout_g, h, w, k, stride, pad, mode, oh if mode == 1 else 0,
                               ow if mode == 1 else 0, allow=(_lib.NOT_SERVED,))
            if rc != 0:
                y = conv_forward(x, wmat, batch, groups, cin_g, cout_g, k, stride, pad, mode, in_scale, out_scale,
                                 out_hw=out_hw, grad=grad)
                hw = oh * ow
                _lib.call('gg_fused_bias_act_f32', y, y, act_bias, None, 3, 0, alpha, gain, y.numel(), hw,
                          0 if act_bias is None else act_bias.numel())
        elif act is not None:
            if (k, stride, pad, mode, groups) != (3, 1, 1, 0, 1) or bias is not None or (oh * ow) % 4:
                raise NotImplementedError('conv_forward: the fused activation needs a 3x3 stride-1 pad-1 single-group conv')
            noise, noise_weight, act_bias, alpha, gain = act       # noise / noise_weight / act_bias may be None
            wbuf, stride_l = wmat.split(code) if use_split else (None, 0)
            wm = None if use_split else (wmat.fp32() if isinstance(wmat, PackedWeight) else wmat)
            # the sign plane comes from the binary16-limb patch tile, or (round 6) from the few-input-channel fp32 kernel
            # that serves the perceptual trunk's RGB stem
            few_in = not use_split and cin_g <= 4 and w % 4 == 0 and noise is None and 'fewin_bits' not in DISABLED
            if (want_sign_bits and ((use_split and pack_code == 18) or few_in) and cout_g % 32 == 0
                    and 'sign_bits' not in DISABLED and ACT_OBSERVER is None):
                sign_bits = torch.empty((batch, oh * ow, cout_g // 32), dtype=torch.int32, device=x.device)
            if amax_out is not None and use_split and pack_code == 18:
                _lib.call('gg_modconv3x3_act_amax_f32', y, x, wm, wbuf, stride_l, code, in_scale, out_scale,
                          noise, noise_weight, act_bias, alpha, gain, batch, cin_g, cout_g, h, w, sign_bits, amax_out)
                if sign_bits is not None and not _lib.load().gg_last_sign_bits_written():
                    sign_bits = None
            elif sign_bits is not None:
                _lib.call('gg_modconv3x3_act_bits_f32', y, x, wm, wbuf, stride_l, code if use_split else 0, in_scale,
                          out_scale, noise, noise_weight, act_bias, alpha, gain, batch, cin_g, cout_g, h, w, sign_bits)
                if not _lib.load().gg_last_sign_bits_written():
                    sign_bits = None          # split-K / another kernel served the shape: the backward keeps y
            else:
                _lib.call('gg_modconv3x3_act_f32', y, x, wm, wbuf, stride_l, code if use_split else 0, in_scale, out_scale,
                          noise, noise_weight, act_bias, alpha, gain, batch, cin_g, cout_g, h, w)
        elif use_split:
            wbuf, stride_l = wmat.split(code)
            served = False
            if prelimb is not None and mode == 1 and k == 3 and stride == 2 and pack_code == 18 and groups == 1:
                # the operand already exists style-scaled, split into limbs, channel-fastest (torgb_limb)
                rc = _lib.call('gg_convT3x3s2_prelimb_f32', y, prelimb[0], prelimb[1], wbuf, stride_l, out_scale, bias,
                               batch, cin_g, cout_g, h, w, pad, oh, ow, allow=(_lib.NOT_SERVED,))
                served = rc == 0
            if not served and residual is not None and k == 1 and mode == 0 and 'conv_residual' not in DISABLED:
                # ResBlock's skip convolution with the residual merge in its epilogue / split-K reduce pass
                rc = _lib.call('gg_conv1x1_split_residual_f32', y, x, wbuf, stride_l, code, in_scale, out_scale, bias,
                               residual.contiguous(), batch, groups, cin_g, cout_g, h, w, stride, allow=(_lib.NOT_SERVED,))
                if rc == 0:
                    served, residual = True, None
            if not served:
                _lib.call('gg_conv2d_split_f32', y, x, wbuf, stride_l, code, in_scale, out_scale, bias, batch, groups,
                          cin_g, cout_g, h, w, k, stride, pad, mode, oh if mode == 1 else 0, ow if mode == 1 else 0)
        else:
            wm = wmat.fp32() if isinstance(wmat, PackedWeight) else wmat
            _lib.call('gg_conv2d_f32', y, x, wm, in_scale, out_scale, bias, batch, groups, cin_g, cout_g, h, w,
                      k, stride, pad, mode, oh if mode == 1 else 0, ow if mode == 1 else 0)
        if residual is not None:             # not merged inside the convolution: one pass of its own
            _lib.call('gg_add_scale_f32', y, y, residual.contiguous(), 1.0, y.numel())
        if prof is not None:
            # algorithmic FLOPs: a transposed stride-2 convolution does its multiply-adds at the INPUT positions
            pos = oh * ow if mode == 0 else h * w
            name = last_conv_kernel()          # what the library's dispatcher actually launched for this call
            if prof.every:
                prof.names[sig] = name
            prof.end(start, 2.0 * batch * groups * cout_g * cin_g * k * k * pos, name)
    return y if want_sign_bits is None else (y, sign_bits)


# Gradient slots: weight storage address -> (the tensor its gradient is ACCUMULATED into, weak reference to the
# parameter).  The slot is a view of the trainer's flat, zero-initialised gradient arena; train_step.FlatArena
# registers the trainable conv weights.  Inside a `with grad_slots():` block (the trainer wraps its own backward in
# one) the backward of a registered weight adds the weight gradient straight into the slot
# (gg_conv2d_wgrad_acc_f32) and hands autograd no gradient: no per-layer memset, no temporary and no AccumulateGrad
# add - two tiny launches less per layer.  Outside such a block (torch.autograd.grad, a diagnostic backward, a
# second loss) the registry is ignored and the gradient is returned to autograd as usual.
GRAD_SLOTS = {}
_SLOTS_ACTIVE = False


class grad_slots:
    """Context manager: weight gradients of registered parameters accumulate into their arena slots.

    Enter it ONLY around the `.backward()` call whose gradients are meant for the arena (the trainer's
    `total.backward()`).  Inside the block the backward of a registered parameter ADDS into the arena and hands autograd
    None, so a `torch.autograd.grad(..., inputs=[registered parameter])` issued there would see no gradient for it while
    the arena changes: run diagnostics, penalties on parameter gradients and second losses outside the block."""

    def __enter__(self):
        global _SLOTS_ACTIVE
        self.prev = _SLOTS_ACTIVE
        _SLOTS_ACTIVE = True
        return self

    def __exit__(self, *exc):
        global _SLOTS_ACTIVE
        _SLOTS_ACTIVE = self.prev
        return False


def register_grad_slot(param, slot):
    import weakref
    GRAD_SLOTS[param.data_ptr()] = (slot, weakref.ref(param))


def _slot_for(weight):
    """The registered accumulation slot of `weight`, or None (registry off / not registered / the address now
    belongs to some other tensor)."""
    if not _SLOTS_ACTIVE or 'slots' in DISABLED:
        return None
    ent = GRAD_SLOTS.get(weight.data_ptr())
    if ent is None:
        return None
    slot, ref = ent
    owner = ref()
    if owner is None or owner.data_ptr() != weight.data_ptr() or owner.shape != weight.shape or \
            slot.shape != weight.shape or not slot.is_contiguous():
        return None
    return slot
# developer A/B switches (comma separated names in GG_DISABLE): slots, style_demod, fuse_act, wgrad_rows, lpips_tail,
# pack_registry, mask_dgrad, torgb_fuse
DISABLED = frozenset(filter(None, os.environ.get('GG_DISABLE', '').split(',')))

# Diagnostics hook (tests/test_gpu_act_masks.py, tests/test_gpu_lpips_masks.py): a callable (site, y) handed the OUTPUT of
# every leaky-ReLU / ReLU layer right after it was produced - the tensor the backward takes its branch decisions from.
# An observer may edit y.data (decision replay); while one is installed the layers keep no sign plane, so that the
# edited tensor is what their backward reads.  None (always, outside those tests) costs nothing.
ACT_OBSERVER = None


def observe_activation(site, y):
    if ACT_OBSERVER is not None:
        ACT_OBSERVER(site, y)
    return y
# opt-in paths (GG_ENABLE): mask_wgrad - leaky-ReLU backward inside BOTH gradient kernels of a trainable conv+act layer
# (gg_conv3x3_masked_wgrad_f32).  Measured 1 % SLOWER than the separate 5 TB/s mask pass on the STN shapes (the masked
# kernels read a second tensor), so it is off by default; the frozen layers use the masked data gradient only.
ENABLED = frozenset(filter(None, os.environ.get('GG_ENABLE', '').split(',')))


def conv_wgrad(x, dy, batch, groups, cin_g, cout_g, k, stride, pad, scale=1.0, into=None):
    """-> (groups*cout_g, cin_g, k, k) gradient of a mode-0 convolution's weight; `into`: add it to this tensor
    instead (returns None).  The library picks the kernel (row-streaming 3x3 / RGB-stem reduction / generic) and keeps
    the K-split partial sums in its per-stream scratch; they are added in a fixed order (no float atomics)."""
    h, w = x.shape[-2], x.shape[-1]
    oh, ow = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    limbs = _LIMBS[PRECISION]
    split = bool(limbs and (oh * ow) % 32 == 0 and ow % 4 == 0 and cout_g >= 32 and cin_g * k * k >= 32
                 and dy.data_ptr() % 16 == 0)
    stem = k == 1 and stride == 1 and pad == 0 and groups == 1 and cin_g <= 4 and (h * w) % 64 == 0   # RGB stem
    rows = split and k == 3 and stride == 1 and pad == 1 and (w % 32 == 0 or (w == 16 and h % 2 == 0))
    dw = into if into is not None else torch.empty((groups * cout_g, cin_g, k, k), dtype=torch.float32, device=x.device)
    sig = ('wgrad', batch, groups, cin_g, cout_g, h, w, k, stride, pad, limbs)
    prof, start = _prof_begin(sig)
    try:
        return _conv_wgrad(x, dy, batch, groups, cin_g, cout_g, k, stride, pad, scale, into, h, w, limbs, split, stem,
                           rows, dw)
    finally:
        _prof_end(prof, start, sig, 2.0 * batch * groups * cout_g * cin_g * k * k * oh * ow,
                  f'conv_wgrad<k{k},s{stride},{"rows" if rows else ("stem" if stem else "generic")},limbs{limbs if (split or stem) else 0}>')


def _conv_wgrad(x, dy, batch, groups, cin_g, cout_g, k, stride, pad, scale, into, h, w, limbs, split, stem, rows, dw):
    if 'wgrad_rows' in DISABLED and (stem or rows):
        # A/B switch: the generic kernels (the row-streaming kernel is what gg_conv2d_wgrad_ws_f32 would choose)
        if into is not None:
            _lib.call('gg_conv2d_wgrad_acc_f32', into, x, dy, batch, groups, cin_g, cout_g, h, w, k, stride, pad, scale,
                      limbs if split else 0)
        elif split:
            _lib.call('gg_conv2d_wgrad_split_f32', dw, x, dy, batch, groups, cin_g, cout_g, h, w, k, stride, pad, scale,
                      limbs)
        else:
            _lib.call('gg_conv2d_wgrad_f32', dw, x, dy, batch, groups, cin_g, cout_g, h, w, k, stride, pad, scale)
        return None if into is not None else dw
    _lib.call('gg_conv2d_wgrad_ws_f32', dw, x, dy, batch, groups, cin_g, cout_g, h, w, k, stride, pad, scale,
              limbs if (split or stem) else 0, 1 if into is not None else 0, None, 0)
    return None if into is not None else dw


class _Conv2d(Function):
    """F.conv2d / F.conv_transpose2d semantics (square 1x1 / 3x3 kernels, stride 1 or 2, dilation 1)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, groups, transposed, output_padding, wscale=1.0, residual=None):
        """residual (plain convolutions only): a tensor of the output's shape added to it (ResBlock's merge); its
        gradient is the incoming gradient."""
        if x.dtype != torch.float32 or weight.dtype != torch.float32:
            raise TypeError('conv_mfma: float32 only')
        if residual is not None and transposed:
            raise NotImplementedError('conv_mfma: residual with a transposed convolution')
        x = x.contiguous()
        weight = weight.contiguous()
        batch, cin = x.shape[0], x.shape[1]
        k = weight.shape[-1]
        if weight.shape[-2] != k or k not in (1, 3):
            raise NotImplementedError(f'conv_mfma: kernel {tuple(weight.shape[-2:])} not supported (1x1 / 3x3)')
        if stride not in (1, 2):
            raise NotImplementedError(f'conv_mfma: stride {stride} not supported')
        cin_g = cin // groups
        if not transposed:
            cout_g = weight.shape[0] // groups
            assert weight.shape[1] == cin_g, 'weight / input channel mismatch'
            wmat = packed(weight, groups, cout_g, cin_g, k, 0, 0, wscale)
            y = conv_forward(x, wmat, batch, groups, cin_g, cout_g, k, stride, padding, 0, bias=bias, residual=residual)
        else:
            cout_g = weight.shape[1]
            assert weight.shape[0] == cin, 'weight / input channel mismatch'
            if stride == 1:      # transposed stride-1 == correlation with flipped taps
                wmat = packed(weight, groups, cout_g, cin_g, k, 1, 1, wscale)
                y = conv_forward(x, wmat, batch, groups, cin_g, cout_g, k, 1, k - 1 - padding, 0, bias=bias)
            else:
                wmat = packed(weight, groups, cout_g, cin_g, k, 1, 0, wscale)
                h, w = x.shape[-2:]
                oh = (h - 1) * 2 - 2 * padding + k + output_padding
                ow = (w - 1) * 2 - 2 * padding + k + output_padding
                y = conv_forward(x, wmat, batch, groups, cin_g, cout_g, k, 2, padding, 1, bias=bias, out_hw=(oh, ow))
        ctx.save_for_backward(x, weight)
        ctx.conf = (stride, padding, groups, transposed, output_padding, bias is not None, cin_g, cout_g, k, wscale)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dx, dw, db = conv2d_backward(x, weight, dy, ctx.conf, ctx.needs_input_grad[:3])
        return dx, dw, db, None, None, None, None, None, None, (dy if len(ctx.needs_input_grad) > 9 and ctx.needs_input_grad[9] else None)


def conv2d_backward(x, weight, dy, conf, needs):
    """Data / weight / bias gradients of _Conv2d (also the autograd formula of torch.ops.gangealing.conv2d /
    conv_transpose2d, op/library.py).  conf as saved by _Conv2d.forward; needs = (dx, dw, db) wanted."""
    stride, padding, groups, transposed, output_padding, has_bias, cin_g, cout_g, k, wscale = conf
    dy = dy.contiguous()
    batch = x.shape[0]
    h, w = x.shape[-2:]
    dx = dw = db = None
    if needs[0]:
        if not transposed:
            if stride == 1:
                wm = packed(weight, groups, cin_g, cout_g, k, 1, 1, wscale)
                dx = conv_forward(dy, wm, batch, groups, cout_g, cin_g, k, 1, k - 1 - padding, 0, grad=True)
            else:
                wm = packed(weight, groups, cin_g, cout_g, k, 1, 0, wscale)
                dx = conv_forward(dy, wm, batch, groups, cout_g, cin_g, k, 2, padding, 1, out_hw=(h, w), grad=True)
        else:
            if stride == 1:
                wm = packed(weight, groups, cin_g, cout_g, k, 0, 0, wscale)
                dx = conv_forward(dy, wm, batch, groups, cout_g, cin_g, k, 1, padding, 0, grad=True)
            else:
                wm = packed(weight, groups, cin_g, cout_g, k, 0, 0, wscale)
                dx = conv_forward(dy, wm, batch, groups, cout_g, cin_g, k, 2, padding, 0, grad=True)
                dx = dx[..., :h, :w].contiguous() if dx.shape[-2:] != (h, w) else dx
    if needs[1]:
        slot = _slot_for(weight)
        if not transposed:
            dw = conv_wgrad(x, dy, batch, groups, cin_g, cout_g, k, stride, padding, wscale, into=slot)
        else:
            # dW[ci,co,ky,kx] = sum x[ci,i] * dy[co, i*s + k - p]: a mode-0 weight gradient with roles swapped
            dw = conv_wgrad(dy, x, batch, groups, cout_g, cin_g, k, stride, padding, wscale, into=slot)
    if has_bias and needs[2]:
        db = dy.sum(dim=(0, 2, 3))
    return dx, dw, db


class _Conv3x3BiasAct(Function):
    """leaky_relu(conv3x3(x, weight * wscale, pad 1) + bias, alpha) * gain with the bias + activation in the
    convolution's epilogue: the STN trunk's EqualConv2d + FusedLeakyReLU pair (networks.py:602-640,
    fused_act.py:74-97) and, with alpha = 0 / gain = 1, the VGG16 backbone's conv + ReLU.  The backward masks the
    incoming gradient with the saved OUTPUT (as fused_act.py:27-38), then runs the data / weight gradients."""

    @staticmethod
    def forward(ctx, x, weight, bias, alpha, gain, wscale):
        x = x.contiguous()
        weight = weight.contiguous()
        n, cin = x.shape[0], x.shape[1]
        cout = weight.shape[0]
        wmat = packed(weight, 1, cout, cin, 3, 0, 0, wscale)
        y, ctx.sign_bits = conv_forward(x, wmat, n, 1, cin, cout, 3, 1, 1, 0,
                                        act=(None, None, None if bias is None else bias.contiguous(), alpha, gain),
                                        want_sign_bits=bool(ctx.needs_input_grad[0]))
        observe_activation('conv3x3_bias_act', y)
        ctx.save_for_backward(x, weight, y)
        ctx.conf = (alpha, gain, wscale, bias is not None)
        ctx.bias_ref = bias          # (the parameter itself: looked up in the gradient-slot registry in backward)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        alpha, gain, wscale, has_bias = ctx.conf
        dy = dy.contiguous()
        n, cout, h, w = y.shape
        cin = x.shape[1]
        need_db = has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[0] and not ctx.needs_input_grad[1] and not need_db:
            # frozen layer (VGG backbone): only the data gradient is wanted - mask the gradient inside the conv
            wm = packed(weight, 1, cin, cout, 3, 1, 1, wscale)
            dx = masked_dgrad(dy, y, alpha, gain, wm, n, cout, cin, h, w, sign_bits=ctx.sign_bits)
            if dx is None and ctx.sign_bits is not None and cin <= 4 and cout % 32 == 0:
                # few-input-channel layer (the perceptual trunk's RGB stem): the streaming few-output-channel data
                # gradient reads the activation's backward from the sign plane (gg_conv3x3_fewout_masked_bits_f32)
                dx = torch.empty((n, cin, h, w), dtype=torch.float32, device=dy.device)
                sig = ('masked_dgrad_fewout', n, cout, cin, h, w)
                prof, start = _prof_begin(sig)
                rc = _lib.call('gg_conv3x3_fewout_masked_bits_f32', dx, dy, ctx.sign_bits, alpha, gain, wm.fp32(), n, cout,
                               cin, h, w, allow=(_lib.NOT_SERVED,))
                if rc == 0:
                    _prof_end(prof, start, sig, 2.0 * n * cin * cout * 9 * h * w, last_conv_kernel())
                else:
                    dx = None
            if dx is not None:
                return dx, None, None, None, None, None
        slot = None
        if ctx.needs_input_grad[1]:
            slot = _slot_for(weight)
        if ctx.needs_input_grad[1] and _LIMBS[PRECISION] == 2 and w % 32 == 0 and 'mask_wgrad' in ENABLED \
                and 'wgrad_rows' not in DISABLED:
            # trainable layer on the row-streaming wgrad kernel: the leaky-ReLU backward rides in the loaders of both
            # gradient kernels (and the bias gradient in the wgrad kernel), so the masked gradient is never written
            dx = None
            if ctx.needs_input_grad[0]:
                dx = masked_dgrad(dy, y, alpha, gain, packed(weight, 1, cin, cout, 3, 1, 1, wscale), n, cout, cin, h, w,
                                  sign_bits=ctx.sign_bits)
            if dx is not None or not ctx.needs_input_grad[0]:
                db = torch.zeros(cout, dtype=torch.float32, device=dy.device) if need_db else None
                dw = slot if slot is not None else torch.empty((cout, cin, 3, 3), dtype=torch.float32, device=dy.device)
                rc = _lib.call('gg_conv3x3_masked_wgrad_f32', dw, db, x, dy, y, alpha, gain, n, cin, cout, h, w, wscale,
                               2, 1 if slot is not None else 0, None, 0, allow=(_lib.NOT_SERVED,))
                if rc == 0:
                    return dx, (None if slot is not None else dw), db, None, None, None
        g = torch.empty_like(dy)
        bslot = _slot_for(ctx.bias_ref) if (need_db and ctx.bias_ref is not None) else None
        if bslot is not None:      # the bias gradient is added straight into its arena slot (no temporary, no add)
            _lib.call('gg_fused_lrelu_bwd_acc_f32', g, bslot, dy, y, alpha, gain, n, cout, h * w, 1)
            db = None
        else:
            db = torch.empty(cout, dtype=torch.float32, device=dy.device) if need_db else None
            _lib.call('gg_fused_lrelu_bwd_f32', g, db, dy, y, alpha, gain, n, cout, h * w)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            wm = packed(weight, 1, cin, cout, 3, 1, 1, wscale)
            dx = conv_forward(g, wm, n, 1, cout, cin, 3, 1, 1, 0, grad=True)
        if ctx.needs_input_grad[1]:
            dw = conv_wgrad(x, g, n, 1, cin, cout, 3, 1, 1, wscale, into=slot)
        return dx, dw, db, None, None, None


class _Conv3x3BiasActBlur(Function):
    """blur4x4(leaky_relu(conv3x3(x, weight * wscale, pad 1) + bias, alpha) * gain): ResBlock's conv1 and the Blur in front
    of its down-sampling convolution (networks.py:375-386) as ONE node.  The forward is the two launches it always was; in
    the backward the Blur's adjoint applies the activation's backward in its epilogue and leaves the bias gradient
    (gg_blur4_act_bwd_f32) - the masked gradient is produced once instead of being written blurred, re-read with the saved
    output and written again by a separate pass - and the two gradient convolutions follow as in _Conv3x3BiasAct."""

    @staticmethod
    def forward(ctx, x, weight, bias, alpha, gain, wscale, kernel, pad):
        from .upfirdn2d import _launch
        x = x.contiguous()
        weight = weight.contiguous()
        n, cin = x.shape[0], x.shape[1]
        cout = weight.shape[0]
        wmat = packed(weight, 1, cout, cin, 3, 0, 0, wscale)
        y = conv_forward(x, wmat, n, 1, cin, cout, 3, 1, 1, 0, act=(None, None, bias.contiguous(), alpha, gain))
        p0, p1 = pad
        b = _launch(y, kernel, 1, 1, 1, 1, p0, p1, p0, p1)
        ctx.save_for_backward(x, weight, y, kernel)
        ctx.conf = (alpha, gain, wscale, p0, p1)
        ctx.bias_ref = bias
        return b

    @staticmethod
    def backward(ctx, db):
        from .upfirdn2d import _flipped, _launch
        x, weight, y, kernel = ctx.saved_tensors
        alpha, gain, wscale, p0, p1 = ctx.conf
        db = db.contiguous()
        n, cout, h, w = y.shape
        cin = x.shape[1]
        bh, bw = db.shape[-2:]
        g0, g1 = 4 - p0 - 1, w - bw + p0                      # the adjoint's padding (upfirdn2d.py:113-118, up = down = 1)
        need_db = ctx.needs_input_grad[2]
        bslot = _slot_for(ctx.bias_ref) if need_db else None
        dbias = bslot if bslot is not None else (torch.empty(cout, dtype=torch.float32, device=db.device) if need_db else None)
        g = torch.empty_like(y)
        rc = _lib.call('gg_blur4_act_bwd_f32', g, db, _flipped(kernel), n, cout, bh, bw, g0, g1, g0, g1, y, alpha, gain, dbias,
                       1 if bslot is not None else 0, allow=(_lib.NOT_SERVED,))
        if rc != 0:                                           # small planes: adjoint blur, then the activation's backward
            dy = _launch(db, _flipped(kernel), 1, 1, 1, 1, g0, g1, g0, g1)
            if bslot is not None:
                _lib.call('gg_fused_lrelu_bwd_acc_f32', g, bslot, dy, y, alpha, gain, n, cout, h * w, 1)
            else:
                _lib.call('gg_fused_lrelu_bwd_f32', g, dbias, dy, y, alpha, gain, n, cout, h * w)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = conv_forward(g, packed(weight, 1, cin, cout, 3, 1, 1, wscale), n, 1, cout, cin, 3, 1, 1, 0, grad=True)
        if ctx.needs_input_grad[1]:
            dw = conv_wgrad(x, g, n, 1, cin, cout, 3, 1, 1, wscale, into=_slot_for(weight))
        return dx, dw, (None if bslot is not None else dbias), None, None, None, None, None


def conv3x3_bias_act_blur(input, weight, bias, kernel, pad, negative_slope=0.2, scale=2 ** 0.5, weight_scale=1.0):
    """blur4x4(conv3x3 + bias + leaky ReLU) with the Blur's backward carrying the activation's (see _Conv3x3BiasActBlur)."""
    return _Conv3x3BiasActBlur.apply(input, weight, bias, float(negative_slope), float(scale), float(weight_scale),
                                     kernel.to(input.dtype), (int(pad[0]), int(pad[1])))


def conv3x3_bias_act(input, weight, bias=None, negative_slope=0.2, scale=2 ** 0.5, weight_scale=1.0):
    """3x3 / stride 1 / pad 1 convolution + bias + leaky ReLU (* scale) in one kernel where the shape allows
    (gg_modconv3x3_act_f32; the library falls back to conv + activation pass internally otherwise)."""
    if input.dtype != torch.float32 or weight.shape[-1] != 3 or weight.shape[-2] != 3:
        raise NotImplementedError('conv3x3_bias_act: float32 3x3 kernels only')
    if (input.shape[-1] * input.shape[-2]) % 4 or 'fuse_act' in DISABLED:
        out = conv2d(input, weight, None, 1, 1, weight_scale=weight_scale)
        from .fused_act import fused_leaky_relu
        if bias is None:
            bias = out.new_zeros(weight.shape[0])
        return fused_leaky_relu(out, bias, negative_slope, scale)
    return _Conv3x3BiasAct.apply(input, weight, bias, float(negative_slope), float(scale), float(weight_scale))


def conv2d(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1, weight_scale=1.0, residual=None):
    """weight_scale folds EqualConv2d's runtime `weight * scale` (networks.py:98,112) into the weight
    packing kernel (and into the weight gradient), saving an elementwise pass per call.
    residual: conv2d(...) + residual with the sum inside the convolution where the kernel allows (1x1, split precision)."""
    if _pair_eq(dilation, 'dilation') != 1:
        raise NotImplementedError('conv_mfma: dilation != 1')
    return _Conv2d.apply(input, weight, bias, _pair_eq(stride, 'stride'), _pair_eq(padding, 'padding'), groups,
                         False, 0, float(weight_scale), residual)


class _ConvBiasActS2(Function):
    """leaky_relu(conv3x3(x, weight * wscale, stride 2, pad 0) + bias, alpha) * gain: ResBlock's down-sampling convolution
    with its FusedLeakyReLU (networks.py:375-386) as one node - the activation inside the library in the forward
    (gg_conv2d_split_act_f32), its backward + bias gradient in one pass in front of the two gradient convolutions."""

    @staticmethod
    def forward(ctx, x, weight, bias, alpha, gain, wscale):
        x = x.contiguous()
        weight = weight.contiguous()
        n, cin = x.shape[0], x.shape[1]
        cout = weight.shape[0]
        wmat = packed(weight, 1, cout, cin, 3, 0, 0, wscale)
        y = conv_forward(x, wmat, n, 1, cin, cout, 3, 2, 0, 0, act=(None, None, bias.contiguous(), alpha, gain))
        observe_activation('fused_leaky_relu', y)
        ctx.save_for_backward(x, weight, y)
        ctx.conf = (alpha, gain, wscale, cin, cout)
        ctx.bias_ref = bias
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        alpha, gain, wscale, cin, cout = ctx.conf
        dy = dy.contiguous()
        n, _, oh, ow = y.shape
        need_db = ctx.needs_input_grad[2]
        g = torch.empty_like(dy)
        bslot = _slot_for(ctx.bias_ref) if (need_db and ctx.bias_ref is not None) else None
        if bslot is not None:
            _lib.call('gg_fused_lrelu_bwd_acc_f32', g, bslot, dy, y, alpha, gain, n, cout, oh * ow, 1)
            db = None
        else:
            db = torch.empty(cout, dtype=torch.float32, device=dy.device) if need_db else None
            _lib.call('gg_fused_lrelu_bwd_f32', g, db, dy, y, alpha, gain, n, cout, oh * ow)
        dx, dw, _ = conv2d_backward(x, weight, g, (2, 0, 1, False, 0, False, cin, cout, 3, wscale),
                                    (ctx.needs_input_grad[0], ctx.needs_input_grad[1], False))
        return dx, dw, db, None, None, None


def conv3x3s2_bias_act(input, weight, bias, negative_slope=0.2, scale=2 ** 0.5, weight_scale=1.0):
    """3x3 / stride 2 / pad 0 convolution + bias + leaky ReLU (* scale)."""
    return _ConvBiasActS2.apply(input, weight, bias, float(negative_slope), float(scale), float(weight_scale))


def conv_transpose2d(input, weight, bias=None, stride=1, padding=0, output_padding=0, groups=1, dilation=1):
    if _pair_eq(dilation, 'dilation') != 1:
        raise NotImplementedError('conv_mfma: dilation != 1')
    return _Conv2d.apply(input, weight, bias, _pair_eq(stride, 'stride'), _pair_eq(padding, 'padding'), groups,
                         True, _pair_eq(output_padding, 'output_padding'))


def plane_dot(a, b):
    """(N,C,H,W) x (N,C,H,W) -> (N,C): per-plane dot products."""
    n, c = a.shape[0], a.shape[1]
    out = torch.empty((n, c), dtype=torch.float32, device=a.device)
    _lib.call('gg_plane_dot_f32', out, a.contiguous(), b.contiguous(), n * c, a.numel() // max(n * c, 1))
    return out


def prelimb_wanted(cin, res):
    """Write the next up-sampling convolution's operand in limb form (round 6, csrc/conv_t_c16.hip: PRELIMB)?  The
    binary16 forward arithmetic, a shape the 16-channel-chunk transposed tile serves and the fused ToRGB pass measured
    (profiles/r06_c_prelimb_probe.txt: input resolutions 32 - 128; 256 for the 512^2 generators), no diagnostics hook that
    might edit the activation afterwards."""
    return (PRECISION in _F16_FORWARD and 'prelimb' not in DISABLED and ACT_OBSERVER is None and cin % 32 == 0
            and cin <= 1024 and res >= 32 and (res & (res - 1)) == 0)


def torgb_limb(y, rgb_wmat, rgb_style, rgb_bias, next_style, amax):
    """-> (rgb (N,3,H,W), xlimb, xexp) or None: ToRGB of y (modulated 1x1, no demodulation, bias in the epilogue) and y's
    limb form for the next layer's transposed convolution with style `next_style`, in one pass (gg_torgb_limb_f32)."""
    n, cin, h, w = y.shape
    if rgb_bias is None or not isinstance(rgb_wmat, PackedWeight):
        return None
    rgb = torch.empty((n, 3, h, w), dtype=torch.float32, device=y.device)
    xlimb = torch.empty(n * cin * h * w * 2, dtype=torch.int16, device=y.device)
    xexp = torch.empty(n, dtype=torch.int32, device=y.device)
    sig = ('torgb_limb', n, cin, h, w)
    prof, start = _prof_begin(sig)
    rc = _lib.call('gg_torgb_limb_f32', rgb, xlimb, xexp, y, rgb_wmat.fp32(), rgb_style.contiguous(), rgb_bias.contiguous(),
                   next_style.contiguous(), amax, n, cin, h * w, allow=(_lib.NOT_SERVED,))
    if rc != 0:
        return None
    _prof_end(prof, start, sig, 4.0 * (2 * y.numel() + rgb.numel()), 'torgb_limb', 'byte')
    return rgb, xlimb, xexp


class _ModulatedConv(Function):
    """y[n,co] = demod[n,co] * conv(W*scale, style[n,ci] * x[n,ci])  - one dense conv with shared weights.

    Algebraically identical to the reference's per-sample grouped convolution (networks.py:243-280;
    SURVEY.md Appendix C.1) but with no (N,Cout,Cin,k,k) weight tensor and no per-sample weight gradient:
    the style gradient falls out of per-plane dot products.  The generator weights are frozen on this
    path (train.py:64-65), so no weight gradient is produced.
      wmat_fwd / wmat_bwd: packed GEMM weights for the forward conv and its dgrad (cached by the caller)
      wsq: (Cout,Cin) sum over taps of (W*scale)^2, for the demodulation
    """

    @staticmethod
    def forward(ctx, x, style, wmat_fwd, wmat_bwd, wsq, k, upsample, demodulate, demod_pre=None, bias=None,
                prelimb=None):
        """bias: (Cout,) frozen per-channel bias added in the convolution's epilogue (ToRGB's bias, networks.py:366).
        prelimb: (xlimb, xexp) of x * style (torgb_limb) for an up-sampling layer."""
        x = x.contiguous()
        style = style.contiguous()
        n, cin, h, w = x.shape
        cout = wmat_fwd.cout_g
        demod = None
        if demodulate:
            demod = demod_pre if demod_pre is not None else torch.rsqrt((style * style) @ wsq.t() + 1e-8)
        if upsample:
            y = conv_forward(x, wmat_fwd, n, 1, cin, cout, k, 2, 0, 1, in_scale=style, out_scale=demod, bias=bias,
                             prelimb=prelimb)
        else:
            y = conv_forward(x, wmat_fwd, n, 1, cin, cout, k, 1, k // 2, 0, in_scale=style, out_scale=demod, bias=bias)
        ctx.save_for_backward(x, style, demod if demod is not None else style.new_empty(0), y, wsq)
        ctx.wmat_bwd = wmat_bwd
        ctx.conf = (k, upsample, demodulate)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, style, demod, y, wsq = ctx.saved_tensors
        wmat_bwd = ctx.wmat_bwd
        k, upsample, demodulate = ctx.conf
        dy = dy.contiguous()
        n, cin, h, w = x.shape
        cout = y.shape[1]
        dscale = demod if demodulate else None
        need_style = ctx.needs_input_grad[1]
        dx = dstyle = None
        if ctx.needs_input_grad[0] or need_style:
            # dx~ = dgrad(dy * demod); dx = dx~ * style.  When the style gradient is needed dx~ is kept
            # un-scaled for the per-plane dot product; otherwise the style scale rides in the epilogue.
            osc = None if need_style else style
            if upsample:
                dxt = conv_forward(dy, wmat_bwd, n, 1, cout, cin, k, 2, 0, 0, in_scale=dscale, out_scale=osc, grad=True)
                if dxt.shape[-2:] != (h, w):
                    dxt = dxt[..., :h, :w].contiguous()
            else:
                dxt = conv_forward(dy, wmat_bwd, n, 1, cout, cin, k, 1, k - 1 - k // 2, 0, in_scale=dscale,
                                   out_scale=osc, grad=True)
            if need_style:
                dstyle = plane_dot(dxt, x)
                if demodulate:
                    # + the demodulation branch: d demod / d style = -demod^3 * style * wsq, with d loss / d demod =
                    # <dy, y> / demod - one launch (csrc/modulation.hip) instead of nine tiny torch kernels per layer
                    dot_y = plane_dot(dy, y)
                    if 'style_grad' in DISABLED or cout > 1024:
                        dsq = (dot_y / demod) * (-0.5) * demod * demod * demod
                        dstyle = dstyle + 2.0 * style * (dsq @ wsq)
                    else:
                        dot_x = dstyle
                        dstyle = torch.empty_like(dot_x)
                        _lib.call('gg_modconv_style_grad_f32', dstyle, dot_x, dot_y, demod.contiguous(), style,
                                  wsq.contiguous(), n, cin, cout)
                dx = dxt * style.view(n, cin, 1, 1) if ctx.needs_input_grad[0] else None
            else:
                dx = dxt
        return dx, dstyle, None, None, None, None, None, None, None, None, None


def masked_dgrad(dy, y_act, alpha, gain, wmat_bwd, n, cin, cout, h, w, in_scale=None, out_scale=None, sign_bits=None):
    """Data gradient of a 3x3 conv + leaky-ReLU layer with the activation's backward applied while the gradient is
    gathered (gg_conv3x3_masked_dgrad_f32): no separate masked-gradient tensor.  `cin` = reduction channels (the
    layer's output channels), `cout` = the layer's input channels.  None when the shape is not served.
    sign_bits: the 1-bit plane the layer's forward wrote (conv_forward(want_sign_bits=True)); the gather then reads one
    word per pixel and 32 channels instead of the 32 saved outputs - bitwise the same gradient."""
    limbs = limb_code(grad=True) & ~GRAD_OPERAND          # (the masked entry points imply the gradient-operand rule)
    if limbs not in (1, 2, 18) or 'mask_dgrad' in DISABLED or not isinstance(wmat_bwd, PackedWeight) or \
            not wmat_bwd.split_ok():
        return None
    wbuf, stride_l = wmat_bwd.split(limbs)
    dx = torch.empty((n, cout, h, w), dtype=torch.float32, device=dy.device)
    sig = ('masked_dgrad', n, cin, cout, h, w, in_scale is None, sign_bits is not None, limbs)
    prof, start = _prof_begin(sig)
    work = 2.0 * n * cin * cout * 9 * h * w
    if sign_bits is not None and limbs == 18:
        rc = _lib.call('gg_conv3x3_masked_dgrad_bits_f32', dx, dy, sign_bits, alpha, gain, wbuf, stride_l, limbs,
                       in_scale, out_scale, n, cin, cout, h, w, allow=(_lib.NOT_SERVED,))
        if rc == 0:
            _prof_end(prof, start, sig, work, last_conv_kernel() + '+lrelu-mask(sign plane)')
            return dx
    rc = _lib.call('gg_conv3x3_masked_dgrad_f32', dx, dy, y_act, alpha, gain, wbuf, stride_l, limbs, in_scale, out_scale,
                   n, cin, cout, h, w, allow=(_lib.NOT_SERVED,))
    if rc == 0:
        _prof_end(prof, start, sig, work, last_conv_kernel() + '+lrelu-mask(fp32 output)')
    return dx if rc == 0 else None


class _ModulatedConvAct(Function):
    """StyledConv without upsampling, in one kernel: lrelu(demod * conv(W, style * x) + nw * noise + b) * gain
    (networks.py:243-298, 344-350).  Used when no gradient is wanted for the style, the noise weight or the
    activation bias (frozen generator beyond the learned W+ slots): the backward is the leaky-ReLU mask
    (sign reference = the saved OUTPUT, as in fused_act.py:27-38) followed by the data gradient."""

    @staticmethod
    def forward(ctx, x, style, wmat_fwd, wmat_bwd, wsq, demodulate, noise, noise_weight, act_bias, alpha, gain,
                demod_pre=None):
        x = x.contiguous()
        style = style.contiguous()
        n, cin, h, w = x.shape
        cout = wmat_fwd.cout_g
        demod = None
        if demodulate:
            demod = demod_pre if demod_pre is not None else torch.rsqrt((style * style) @ wsq.t() + 1e-8)
        y, ctx.sign_bits = conv_forward(x, wmat_fwd, n, 1, cin, cout, 3, 1, 1, 0, in_scale=style, out_scale=demod,
                                        act=(noise.contiguous(), noise_weight.contiguous(), act_bias.contiguous(), alpha,
                                             gain), want_sign_bits=bool(ctx.needs_input_grad[0]))
        observe_activation('styled_conv', y)
        ctx.save_for_backward(style, demod if demod is not None else style.new_empty(0), y)
        ctx.wmat_bwd = wmat_bwd
        ctx.conf = (demodulate, alpha, gain, cin)
        return y

    @staticmethod
    def backward(ctx, dy):
        style, demod, y = ctx.saved_tensors
        demodulate, alpha, gain, cin = ctx.conf
        if not ctx.needs_input_grad[0]:
            return (None,) * 12
        dy = dy.contiguous()
        n, cout, h, w = y.shape
        dx = masked_dgrad(dy, y, alpha, gain, ctx.wmat_bwd, n, cout, cin, h, w, demod if demodulate else None, style,
                          sign_bits=ctx.sign_bits)
        if dx is None:
            g = torch.empty_like(dy)
            _lib.call('gg_fused_lrelu_bwd_f32', g, None, dy, y, alpha, gain, n, cout, h * w)
            dx = conv_forward(g, ctx.wmat_bwd, n, 1, cout, cin, 3, 1, 1, 0, in_scale=demod if demodulate else None,
                              out_scale=style, grad=True)
        return (dx,) + (None,) * 11


class _StyledConvToRGB(Function):
    """(y, rgb) = one-kernel StyledConv (see _ModulatedConvAct) followed by the ToRGB layer's modulated 1x1 convolution
    on y (networks.py:352-372).  For resolutions where y also feeds the next up-sampling layer: the backward receives
    the next layer's data gradient and the RGB gradient together and adds the ToRGB branch into the former in one
    read-modify-write pass (gg_torgb_dgrad_add_f32) instead of materialising a second full-size gradient and letting
    autograd add the two (a 1x1 dgrad conv + a 3-pass element-wise add per resolution)."""

    @staticmethod
    def forward(ctx, x, style, wmat_fwd, wmat_bwd, wsq, demodulate, noise, noise_weight, act_bias, alpha, gain,
                demod_pre, rgb_style, rgb_wmat, rgb_weight, rgb_scale, rgb_bias=None, next_style=None):
        y, rgb, ctx.sign_bits, demod, rgb_style, limb = styled_conv_torgb_launches(
            x, style, wmat_fwd, wsq, demodulate, noise, noise_weight, act_bias, alpha, gain, demod_pre, rgb_style, rgb_wmat,
            rgb_bias, next_style, bool(ctx.needs_input_grad[0]))
        ctx.save_for_backward(style.contiguous(), demod if demod is not None else style.new_empty(0), y, rgb_style,
                              rgb_weight)
        ctx.wmat_bwd = wmat_bwd
        ctx.conf = (demodulate, alpha, gain, x.shape[1], rgb_scale)
        if limb is None:
            limb = (y.new_empty(0, dtype=torch.int16), y.new_empty(0, dtype=torch.int32))
        ctx.mark_non_differentiable(*limb)
        # (otherwise autograd hands backward() ZERO tensors for the limb-form outputs: a 270 MB fill per layer)
        ctx.set_materialize_grads(False)
        return (y, rgb) + limb

    @staticmethod
    def backward(ctx, gy, grgb, _gl=None, _ge=None):
        style, demod, y, rgb_style, rgb_weight = ctx.saved_tensors
        demodulate, alpha, gain, cin, rgb_scale = ctx.conf
        if not ctx.needs_input_grad[0]:
            return (None,) * 18
        n, cout, h, w = y.shape
        if gy is None:
            g = torch.zeros_like(y)
        else:
            g = gy.contiguous()              # produced for this node only (y's other consumer is internal): updated in place
        if grgb is not None:
            _lib.call('gg_torgb_dgrad_add_f32', g, grgb.contiguous(), rgb_weight, rgb_style, rgb_scale, n, cout, h * w)
        dx = masked_dgrad(g, y, alpha, gain, ctx.wmat_bwd, n, cout, cin, h, w, demod if demodulate else None, style,
                          sign_bits=ctx.sign_bits)
        if dx is None:
            gm = torch.empty_like(g)
            _lib.call('gg_fused_lrelu_bwd_f32', gm, None, g, y, alpha, gain, n, cout, h * w)
            dx = conv_forward(gm, ctx.wmat_bwd, n, 1, cout, cin, 3, 1, 1, 0, in_scale=demod if demodulate else None,
                              out_scale=style, grad=True)
        return (dx,) + (None,) * 17


def styled_conv_torgb_launches(x, style, wmat_fwd, wsq, demodulate, noise, noise_weight, act_bias, alpha, gain,
                               demod_pre, rgb_style, rgb_wmat, rgb_bias, next_style, want_sign_bits):
    """The launches of one resolution's second StyledConv + its ToRGB layer (no autograd here: _StyledConvToRGB wraps
    them when a gradient is wanted, the no-grad generator pass calls them directly).
    -> (y, rgb, sign plane or None, demodulation, rgb style, (xlimb, xexp) or None).
    next_style: the style vectors of the NEXT resolution's up-sampling convolution - when given (prelimb_wanted) the
    convolution's epilogue also leaves max |y| per image and the ToRGB pass writes y's limb form for that layer."""
    x = x.contiguous()
    style = style.contiguous()
    n, cin, h, w = x.shape
    cout = wmat_fwd.cout_g
    demod = None
    if demodulate:
        demod = demod_pre if demod_pre is not None else torch.rsqrt((style * style) @ wsq.t() + 1e-8)
    amax = torch.zeros(n, dtype=torch.float32, device=x.device) if next_style is not None else None
    y, sign_bits = conv_forward(x, wmat_fwd, n, 1, cin, cout, 3, 1, 1, 0, in_scale=style, out_scale=demod,
                                act=(noise.contiguous(), noise_weight.contiguous(), act_bias.contiguous(), alpha, gain),
                                want_sign_bits=want_sign_bits, amax_out=amax)
    have_amax = amax is not None and last_amax_written()
    observe_activation('styled_conv', y)
    rgb_style = rgb_style.contiguous()
    limb = None
    if have_amax:
        res = torgb_limb(y, rgb_wmat, rgb_style, rgb_bias, next_style, amax)
        if res is not None:
            rgb, limb = res[0], (res[1], res[2])
    if limb is None:
        rgb = conv_forward(y, rgb_wmat, n, 1, cout, 3, 1, 1, 0, 0, in_scale=rgb_style, bias=rgb_bias)
    return y, rgb, sign_bits, demod, rgb_style, limb


def styled_conv_torgb(x, style, wmat_fwd, wmat_bwd, wsq, demodulate, act, demod, rgb_style, rgb_wmat, rgb_weight,
                      rgb_scale, rgb_bias=None, next_style=None):
    """-> (y, rgb, (xlimb, xexp) or None)"""
    noise, noise_weight, act_bias, alpha, gain = act
    if not (torch.is_grad_enabled() and x.requires_grad):
        y, rgb, _, _, _, limb = styled_conv_torgb_launches(x, style, wmat_fwd, wsq, demodulate, noise, noise_weight,
                                                           act_bias, alpha, gain, demod, rgb_style, rgb_wmat, rgb_bias,
                                                           next_style, False)
        return y, rgb, limb
    y, rgb, xlimb, xexp = _StyledConvToRGB.apply(x, style, wmat_fwd, wmat_bwd, wsq, demodulate, noise, noise_weight,
                                                 act_bias, alpha, gain, demod, rgb_style, rgb_wmat, rgb_weight,
                                                 float(rgb_scale), rgb_bias, next_style)
    return y, rgb, ((xlimb, xexp) if xlimb.numel() else None)


def style_demod(latent, weight, bias, w_scale, b_scale, wsq=None, eps=1e-8):
    """(style, demod) of one modulated layer from its W+ slot in a single launch (gg_style_demod_f32):
    style = EqualLinear(latent) (networks.py:214-216), demod = rsqrt(style^2 @ wsq^T + eps) (:244-249; None when
    wsq is None).  No autograd: for layers whose latent / modulation weights need no gradient."""
    if latent.dim() != 2 or latent.stride(1) != 1:
        latent = latent.contiguous()
    n, style_dim = latent.shape
    cin = weight.shape[0]
    style = torch.empty((n, cin), dtype=torch.float32, device=latent.device)
    demod = None
    cout = 0
    if wsq is not None:
        cout = wsq.shape[0]
        demod = torch.empty((n, cout), dtype=torch.float32, device=latent.device)
    _lib.call('gg_style_demod_f32', style, demod, _lib.Strided(latent), latent.stride(0), weight.contiguous(),
              None if bias is None else bias.contiguous(), wsq, n, style_dim, cin, cout, w_scale, b_scale, eps)
    return style, demod


class _StyleDemodGrad(Function):
    """style_demod for a W+ slot that needs a gradient (the latent learner's slots) while the modulation layer itself is
    frozen: the same single launch forward (instead of F.linear + mul + mm + add + rsqrt), one matrix product backward.
    The demodulation is returned non-differentiable: _ModulatedConv's backward already carries d demod / d style in the
    style gradient it returns (gg_modconv_style_grad_f32)."""

    @staticmethod
    def forward(ctx, latent, weight, bias, w_scaled, w_scale, b_scale, wsq, eps):
        style, demod = style_demod(latent, weight, bias, w_scale, b_scale, wsq, eps)
        ctx.save_for_backward(w_scaled)
        if demod is None:
            demod = style.new_empty(0)
        ctx.mark_non_differentiable(demod)
        return style, demod

    @staticmethod
    def backward(ctx, dstyle, _ddemod):
        (w_scaled,) = ctx.saved_tensors
        return torch.mm(dstyle.contiguous(), w_scaled), None, None, None, None, None, None, None


def style_demod_grad(latent, weight, bias, w_scaled, w_scale, b_scale, wsq=None, eps=1e-8):
    """(style, demod) as style_demod, differentiable w.r.t. `latent` (w_scaled = weight * w_scale, cached by the caller)."""
    style, demod = _StyleDemodGrad.apply(latent, weight, bias, w_scaled, w_scale, b_scale, wsq, eps)
    return style, (demod if wsq is not None else None)


def modulated_conv2d(x, style, wmat_fwd, wmat_bwd, wsq, k, upsample=False, demodulate=True, act=None, demod=None,
                     bias=None, prelimb=None):
    """act = (noise, noise_weight, act_bias, alpha, gain) fuses the StyledConv tail (3x3, no upsampling, and no
    gradient wanted for style / noise weight / bias).  demod: precomputed demodulation (style_demod)."""
    if act is not None:
        noise, noise_weight, act_bias, alpha, gain = act
        if k != 3 or upsample or style.requires_grad or noise_weight.requires_grad or act_bias.requires_grad:
            raise NotImplementedError('modulated_conv2d: fused activation not applicable to this layer')
        return _ModulatedConvAct.apply(x, style, wmat_fwd, wmat_bwd, wsq, demodulate, noise, noise_weight, act_bias,
                                       alpha, gain, demod)
    if bias is not None and bias.requires_grad and torch.is_grad_enabled():
        raise NotImplementedError('modulated_conv2d: the epilogue bias is for frozen layers')
    return _ModulatedConv.apply(x, style, wmat_fwd, wmat_bwd, wsq, k, upsample, demodulate, demod, bias, prelimb)


class _AddScale(Function):
    """(a + b) * scale in one pass (ResBlock's residual merge, networks.py:392-393); both inputs receive g * scale."""

    @staticmethod
    def forward(ctx, a, b, scale):
        a, b = a.contiguous(), b.contiguous()
        out = torch.empty_like(a)
        _lib.call('gg_add_scale_f32', out, a, b, scale, a.numel())
        ctx.scale = scale
        return out

    @staticmethod
    def backward(ctx, g):
        gs = g if ctx.scale == 1.0 else g * ctx.scale
        return gs, gs, None


def add_scale(a, b, scale):
    if a.shape != b.shape or a.dtype != torch.float32 or b.dtype != torch.float32 or 'add_scale' in DISABLED:
        return (a + b) * scale
    return _AddScale.apply(a, b, float(scale))


class StyleBank:
    """Job tables for gg_style_bank_f32: the modulation (+ demodulation) vectors of a list of frozen layers in two
    launches.  entries: [(slot, mod_weight (cin, D), mod_bias (cin,) or None, w_scale, b_scale, wsq (cout, cin) or None,
    eps)].  `run(latent)` -> [(style (N, cin), demod (N, cout) or None)] as views of one freshly allocated buffer."""

    def __init__(self, entries, device):
        import numpy as np
        dt = np.dtype([('m', '<u8'), ('bias', '<u8'), ('out_off', '<i8'), ('in_off', '<i8'), ('kdim', '<i4'),
                       ('rows', '<i4'), ('scale', '<f4'), ('bias_scale', '<f4'), ('eps', '<f4'), ('pad', '<i4')])
        assert dt.itemsize == 56
        self.keep = entries                      # keeps the weight tensors (and their addresses) alive
        self.layout = []                         # per entry: (style offset, cin, demod offset or None, cout)
        style_rows, demod_rows = [], []
        off = 0                                  # offsets per sample; multiplied by N at run time (jobs are per N)
        for slot, w, b, w_scale, b_scale, wsq, eps in entries:
            cin, dim = w.shape
            if dim > 512 or cin > 512:
                raise ValueError('StyleBank: reduction lengths up to 512')
            s_off = off
            off += cin
            d_off = None
            if wsq is not None:
                d_off = off
                off += wsq.shape[0]
            self.layout.append((s_off, cin, d_off, None if wsq is None else wsq.shape[0]))
            style_rows.append((w.data_ptr(), 0 if b is None else b.data_ptr(), s_off, slot, dim, cin, w_scale, b_scale,
                               0.0, 0))
            if wsq is not None:
                demod_rows.append((wsq.data_ptr(), 0, d_off, s_off, cin, wsq.shape[0], 1.0, 0.0, eps, 0))
        self.per_sample = off
        self.dt, self.device = dt, device
        self.style_rows, self.demod_rows = style_rows, demod_rows
        self.max_style = max(r[5] for r in style_rows)
        self.max_demod = max((r[5] for r in demod_rows), default=0)
        self.tables = {}                         # batch -> (style jobs, demod jobs) device tensors

    def _tables(self, n):
        import numpy as np
        t = self.tables.get(n)
        if t is None:
            def build(rows, in_is_offset):
                if not rows:
                    return None
                arr = np.array([(m, b, o * n, (i * n if in_is_offset else i), k, r, sc, bs, eps, pad)
                                for (m, b, o, i, k, r, sc, bs, eps, pad) in rows], dtype=self.dt)
                return torch.from_numpy(arr.view(np.uint8).reshape(-1).copy()).to(self.device)
            t = self.tables[n] = (build(self.style_rows, False), build(self.demod_rows, True))
        return t

    def run(self, latent):
        """latent: (N, n_latent, D) contiguous float32."""
        n, n_latent, dim = latent.shape
        sj, dj = self._tables(n)
        out = torch.empty(self.per_sample * n, dtype=torch.float32, device=latent.device)
        _lib.call('gg_style_bank_f32', out, latent, n_latent * dim, dim, sj, len(self.style_rows), self.max_style, dj,
                  len(self.demod_rows), self.max_demod, n)
        res = []
        for s_off, cin, d_off, cout in self.layout:
            style = out[s_off * n:(s_off + cin) * n].view(n, cin)
            demod = None if d_off is None else out[d_off * n:(d_off + cout) * n].view(n, cout)
            res.append((style, demod))
        return res

