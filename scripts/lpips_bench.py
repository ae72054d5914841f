#!/usr/bin/env python
"""Time the fused perceptual-distance tail (forward / backward) on the five VGG16 taps of config C2 (batch 16 pairs)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from gangealing_amd import _lib  # noqa: E402

dev = torch.device('cuda:0')
n = 16
tf = tb = 0.0
for (c, h) in [(64, 128), (128, 64), (256, 32), (512, 16), (512, 8)]:
    f = torch.randn(2 * n, c, h, h, device=dev).relu_()
    lin = torch.rand(c, device=dev)
    out = torch.empty(n, device=dev)
    df = torch.zeros_like(f)
    go = torch.ones(n, device=dev)

    def fwd():
        _lib.call('gg_lpips_tail_fwd_f32', out, f, lin, n, c, h * h, 1e-10)

    def bwd():
        _lib.call('gg_lpips_tail_bwd_f32', df, f, lin, go, n, c, h * h, 1e-10, 1)
    res = []
    for fn in (fwd, bwd):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            fn()
        e.record()
        torch.cuda.synchronize()
        res.append(s.elapsed_time(e) / 20 * 1e3)
    tf += res[0]
    tb += res[1]
    print(f'tap {c:3d} x {h:3d}^2  fwd {res[0]:7.1f} us  bwd {res[1]:7.1f} us   ({f.numel() * 4 / 1e6:.0f} MB of features)')
print(f'total fwd {tf:.0f} us  bwd {tb:.0f} us')
