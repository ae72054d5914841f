#!/usr/bin/env python
"""Per-layer micro-benchmark of the implicit-GEMM convolution at config C2 shapes (batch 16).
Prints ms and TFLOP/s (algorithmic) per layer for forward / dgrad / wgrad."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from gangealing_amd.op import conv_mfma as cm  # noqa: E402

dev = torch.device('cuda:0')
N = int(os.environ.get('BATCH', 16))
ITERS = int(os.environ.get('ITERS', 10))


def timeit(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(ITERS):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / ITERS


def layer(name, cin, cout, h, k, stride, pad, mode, scale=True, only=None):
    if only and only not in name:
        return
    x = torch.randn(N, cin, h, h, device=dev)
    w = torch.randn(cout, cin, k, k, device=dev) / (cin * k * k) ** 0.5
    wm = cm.PackedWeight(w, 1, cout, cin, k, 0, 0)
    s_in = torch.rand(N, cin, device=dev) + 0.5 if scale else None
    s_out = torch.rand(N, cout, device=dev) + 0.5 if scale else None
    y = cm.conv_forward(x, wm, N, 1, cin, cout, k, stride, pad, mode, in_scale=s_in, out_scale=s_out)
    oh = y.shape[-1]
    flops = 2.0 * N * cin * cout * k * k * (oh * oh if mode == 0 else h * h)
    t = timeit(lambda: cm.conv_forward(x, wm, N, 1, cin, cout, k, stride, pad, mode, in_scale=s_in, out_scale=s_out))
    if cm.PRECISION != 'fp32':
        cm.set_precision('fp32'); yref = cm.conv_forward(x, wm, N, 1, cin, cout, k, stride, pad, mode, in_scale=s_in, out_scale=s_out)
        cm.set_precision(os.environ['GANGEALING_CONV_PRECISION'])
        err = float((y - yref).abs().max() / yref.abs().max())
        name = name + f' [relerr {err:.1e}]'
    line = f'{name:34s} {cin:4d}->{cout:4d} {h:4d}->{oh:4d} k{k} s{stride} m{mode}  fwd {t:8.3f} ms {flops / t / 1e9:7.1f} TF'
    if mode == 0 and not scale:
        dy = torch.randn_like(y)
        tw = timeit(lambda: cm.conv_wgrad(x, dy, N, 1, cin, cout, k, stride, pad))
        line += f'  wgrad {tw:8.3f} ms {flops / tw / 1e9:7.1f} TF'
    print(line, flush=True)


only = sys.argv[1] if len(sys.argv) > 1 else None
print(f'batch {N}')
# generator (modulated: in/out scales), one pass
for (res, cin, cout) in [(4, 512, 512), (8, 512, 512), (16, 512, 512), (32, 512, 512), (64, 512, 512),
                         (128, 256, 256), (256, 128, 128)]:
    layer(f'G conv {res}', cin, cout, res, 3, 1, 1, 0, only=only)
for (res, cin, cout) in [(4, 512, 512), (8, 512, 512), (16, 512, 512), (32, 512, 512), (64, 512, 256), (128, 256, 128)]:
    layer(f'G upconv {res}->{2 * res}', cin, cout, res, 3, 2, 0, 1, only=only)
    layer(f'G upconv dgrad {2 * res + 1}->{res}', cout, cin, 2 * res + 1, 3, 2, 0, 0, only=only)
layer('G torgb 256', 128, 3, 256, 1, 1, 0, 0, only=only)
# STN (plain convs + wgrad)
for (name, cin, cout, h, k, s, p) in [('STN conv 128', 64, 64, 128, 3, 1, 1), ('STN down 129', 64, 128, 129, 3, 2, 0),
                                      ('STN skip 127', 64, 128, 127, 1, 2, 0), ('STN conv 64', 128, 128, 64, 3, 1, 1),
                                      ('STN down 65', 128, 512, 65, 3, 2, 0), ('STN conv 32', 512, 512, 32, 3, 1, 1),
                                      ('STN down 33', 512, 512, 33, 3, 2, 0), ('STN conv 16', 512, 512, 16, 3, 1, 1),
                                      ('STN mask 16', 512, 576, 16, 3, 1, 1), ('STN first 1x1', 3, 64, 128, 1, 1, 0)]:
    layer(name, cin, cout, h, k, s, p, 0, scale=False, only=only)
layer('STN down dgrad 64->129 (mode1)', 128, 64, 64, 3, 2, 0, 1, scale=False, only=only)
# VGG (batch 2N)
N = 2 * N
for (name, cin, cout, h) in [('VGG 128 3->64', 3, 64, 128), ('VGG 128 64->64', 64, 64, 128), ('VGG 64 128', 128, 128, 64),
                             ('VGG 32 256', 256, 256, 32), ('VGG 16 512', 512, 512, 16), ('VGG 8 512', 512, 512, 8)]:
    layer(name, cin, cout, h, 3, 1, 1, 0, scale=False, only=only)
if only == 'probe':
    N = 16
    layer('probe 1x1 K=1152 @256', 1152, 128, 256, 1, 1, 0, 0, scale=False, only='probe')
    layer('probe 1x1 K=1152 @256 scaled', 1152, 128, 256, 1, 1, 0, 0, scale=True, only='probe')
    layer('probe 3x3 128 @256 noscale', 128, 128, 256, 3, 1, 1, 0, scale=False, only='probe')
    layer('probe 3x3 128 @256 scaled', 128, 128, 256, 3, 1, 1, 0, scale=True, only='probe')
    layer('probe 3x3 512->512 @64 noscale', 512, 512, 64, 3, 1, 1, 0, scale=False, only='probe')
