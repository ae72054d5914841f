#!/usr/bin/env python
"""VERDICT r05 item 2 as a measurement: what would the transposed 16-channel-chunk tile gain if its operand arrived
"written once in MFMA-ready form" (style applied, two binary16 limbs, channel-fastest 32-byte rows) instead of fp32 NCHW
that every consumer tile re-scales, re-reduces (amax) and re-splits?

For each up-convolution shape of config C2 (batch 16, fp16x3): the shipped kernel (fp32 operand + style vector), and the
same kernel with the PRELIMB loader (csrc/conv_t_c16.hip: four 16-byte loads + four ds_write_b128 per patch pixel and
chunk, nothing else) on an operand converted beforehand by gg_debug_limb_convert (timed separately: a producer epilogue
would do that work on registers it already holds).  Outputs must be BITWISE equal (same limbs, E = 0).
MODE=shipped|prelimb (default both) restricts the run to one variant (for rocprofv3 --pmc)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('GANGEALING_CONV_PRECISION', 'fp16x3')
from gangealing_amd import _lib                         # noqa: E402
from gangealing_amd.op import conv_mfma as cm           # noqa: E402

ITERS = int(os.environ.get('ITERS', 30))
MODE = os.environ.get('MODE', 'both')
SHAPES = [(512, 512, 32), (512, 256, 64), (256, 128, 128)]
if os.environ.get('CASE'):
    SHAPES = [s for s in SHAPES if str(s[2]) == os.environ['CASE']]


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    dev = torch.device('cuda:0')
    lib = _lib.load()
    lib.gg_debug_limb_convert.restype = ctypes.c_int
    lib.gg_debug_limb_convert.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                          ctypes.c_longlong, ctypes.c_void_p]
    lib.gg_debug_set_prelimb.restype = ctypes.c_int
    lib.gg_debug_set_prelimb.argtypes = [ctypes.c_void_p]
    cm.set_precision('fp16x3')
    n = 16
    g = torch.Generator().manual_seed(0)
    print(f'# batch {n}, fp16x3, {ITERS} launches each; transposed 3x3 / stride 2 (G up-convolutions, networks.py:254-266)')
    for cin, cout, res in SHAPES:
        x = torch.randn(n, cin, res, res, generator=g).to(dev)
        style = (torch.randn(n, cin, generator=g) * 0.3 + 1.0).to(dev)
        w = (torch.randn(cin, cout, 3, 3, generator=g) / (3 * cin ** 0.5)).to(dev)
        pw = cm.PackedWeight(w, 1, cout, cin, 3, 1, 0, 1.0)
        flop = 2.0 * n * cin * cout * 9 * res * res
        xl = torch.empty(n * cin * res * res * 2, dtype=torch.int16, device=dev)
        stream = torch.cuda.current_stream().cuda_stream

        def shipped():
            return cm.conv_forward(x, pw, n, 1, cin, cout, 3, 2, 0, 1, in_scale=style)

        def convert():
            assert lib.gg_debug_limb_convert(xl.data_ptr(), x.data_ptr(), style.data_ptr(), n * cin, res * res, stream) == 0

        def prelimb():
            lib.gg_debug_set_prelimb(xl.data_ptr())
            return cm.conv_forward(x, pw, n, 1, cin, cout, 3, 2, 0, 1, in_scale=None)

        row = f'upconv {res:3d}->{2 * res + 1:3d}  {cin:3d}->{cout:3d}'
        if MODE in ('both', 'shipped'):
            y0 = shipped()
            name = cm.last_conv_kernel()
            t0 = timed(shipped, ITERS)
            row += f'  shipped {t0:7.4f} ms {flop / t0 / 1e9:6.1f} TF/s [{name}]'
        if MODE in ('both', 'prelimb'):
            convert()
            y1 = prelimb()
            tc = timed(convert, ITERS)
            t1 = timed(prelimb, ITERS)
            row += f'  prelimb {t1:7.4f} ms {flop / t1 / 1e9:6.1f} TF/s (+ conversion pass {tc:6.4f} ms, not a producer cost)'
        if MODE == 'both':
            same = bool(torch.equal(y0, y1))
            row += f'  bitwise-equal {same}  speed-up {t0 / t1:5.3f}x'
            assert same, float((y0 - y1).abs().max())
        print(row, flush=True)


if __name__ == '__main__':
    main()
