#!/usr/bin/env python
"""Attribution probe for the stride-2 family: the G up-convolutions (transposed 3x3 / stride 2) and their data gradients
(3x3 / stride-2 correlations) on random and on all-zero operands (same instruction stream; a power-bound kernel speeds
up on zeros, a schedule- or memory-bound one does not).  Run it once per library build (GANGEALING_HIP_LIB): the
measurement builds drop the weight loads / activation loads / epilogue of the transposed kernel."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from gangealing_amd.op import conv_mfma as cm  # noqa: E402

dev = torch.device('cuda:0')
N = 16
ITERS = int(os.environ.get('ITERS', 100))
cm.set_precision(os.environ.get('GANGEALING_CONV_PRECISION', 'fp16x3'))


def timeit(fn):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(ITERS):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / ITERS


LAYERS = (('upconv 32->65', 512, 512, 32, 1), ('upconv 64->129', 512, 256, 64, 1),
          ('upconv 128->257', 256, 128, 128, 1), ('s2 corr 129->64', 256, 512, 129, 0),
          ('s2 corr 257->128', 128, 256, 257, 0))
if os.environ.get('CONVT_ONLY'):                # substring filter (counter passes)
    LAYERS = tuple(l for l in LAYERS if os.environ['CONVT_ONLY'] in l[0])
for (name, cin, cout, h, mode) in LAYERS:
    g = torch.Generator(device='cpu').manual_seed(1)
    x0 = torch.randn(N, cin, h, h, generator=g).to(dev)
    w0 = (torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).to(dev)
    s_in = torch.ones(N, cin, device=dev)
    s_out = torch.ones(N, cout, device=dev)
    for data, x, w in (('random', x0, w0), ('zeros', torch.zeros_like(x0), torch.zeros_like(w0))):
        wm = cm.PackedWeight(w, 1, cout, cin, 3, 0, 0)
        y = cm.conv_forward(x, wm, N, 1, cin, cout, 3, 2, 0, mode, in_scale=s_in, out_scale=s_out)
        oh = y.shape[-1]
        flops = 2.0 * N * cin * cout * 9 * (oh * oh if mode == 0 else h * h)
        t = timeit(lambda: cm.conv_forward(x, wm, N, 1, cin, cout, 3, 2, 0, mode, in_scale=s_in, out_scale=s_out))
        print(f'{cm.PRECISION:7s} {name:18s} {cin:4d}->{cout:4d} {data:7s} {t:7.4f} ms {flops / t / 1e9:7.1f} TF/s', flush=True)
