#!/usr/bin/env python
"""Is the dominant convolution kernel limited by its schedule or by the chip's power budget?

The same launch (same instruction stream, same addresses) is timed on operands that toggle fewer and fewer bits of the
matrix pipe's inputs: random fp32 data; data that fits ONE 16-bit limb (the low limb planes are all zero: two of the
three MFMA products multiply zeros); one operand zero; both zero.  A schedule-bound kernel takes the same time on all
of them.  A power-bound one speeds up as the data gets quieter - the clock the power manager grants rises.
Usage: python scripts/power_probe.py [out.json]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from gangealing_amd.op import conv_mfma as cm  # noqa: E402

dev = torch.device('cuda:0')
N = 16
ITERS = int(os.environ.get('ITERS', 200))


def timeit(fn):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(ITERS):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / ITERS


def one_limb(t, mode):
    return (t.half() if mode == 'fp16x3' else t.bfloat16()).float()


rows = []
for mode in os.environ.get('PROBE_MODES', 'bf16x3,fp16x3,bf16').split(','):
    cm.set_precision(mode)
    for (cin, cout, h) in ((512, 512, 64), (128, 128, 256)):
        g = torch.Generator(device='cpu').manual_seed(1)
        x0 = torch.randn(N, cin, h, h, generator=g).to(dev)
        w0 = (torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).to(dev)
        s_in = torch.ones(N, cin, device=dev)
        s_out = torch.ones(N, cout, device=dev)
        flops = 2.0 * N * cin * cout * 9 * h * h
        cases = (('random fp32', x0, w0),
                 ('one-limb values (low limb planes zero)', one_limb(x0, mode), one_limb(w0, mode)),
                 ('x random, w zero', x0, torch.zeros_like(w0)),
                 ('x zero, w random', torch.zeros_like(x0), w0),
                 ('all zero', torch.zeros_like(x0), torch.zeros_like(w0)))
        if os.environ.get('PROBE_CASES'):           # e.g. "0,4" under the counter passes
            cases = tuple(cases[int(i)] for i in os.environ['PROBE_CASES'].split(','))
        for data, x, w in cases:
            wm = cm.PackedWeight(w, 1, cout, cin, 3, 0, 0)
            t = timeit(lambda: cm.conv_forward(x, wm, N, 1, cin, cout, 3, 1, 1, 0, in_scale=s_in, out_scale=s_out))
            rows.append(dict(mode=mode, layer=f'{cin}->{cout} @{h}^2', data=data, ms=round(t, 4),
                             tflops=round(flops / t / 1e9, 1)))
            print(f'{mode:7s} {cin:4d}->{cout:4d} @{h:3d}^2  {data:42s} {t:7.4f} ms {flops / t / 1e9:7.1f} TF/s', flush=True)
if len(sys.argv) > 1:
    json.dump(dict(device=torch.cuda.get_device_name(0), iters=ITERS, rows=rows), open(sys.argv[1], 'w'), indent=1)
