#!/usr/bin/env python
"""What HBM rates does this box give to plain streaming kernels?  (Context for the HBM-bound kernels' TB/s figures:
the guide's 8 TB/s is the interface peak.)  1 GiB tensors, fp32."""
import torch

dev = torch.device('cuda:0')
n = 256 * 1024 * 1024
a = torch.randn(n, device=dev)
b = torch.empty_like(a)


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


gib = n * 4 / 1e9
t = timed(lambda: b.copy_(a));            print(f'copy  (read 1 + write 1): {t:7.3f} ms  {2 * gib / t:6.2f} TB/s')
t = timed(lambda: b.fill_(1.0));          print(f'fill  (write 1)         : {t:7.3f} ms  {gib / t:6.2f} TB/s')
t = timed(lambda: a.sum());               print(f'sum   (read 1)          : {t:7.3f} ms  {gib / t:6.2f} TB/s')
t = timed(lambda: torch.add(a, b, out=b)); print(f'add   (read 2 + write 1): {t:7.3f} ms  {3 * gib / t:6.2f} TB/s')
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gangealing_amd.op import fused_leaky_relu
x = a.view(16, 256, 256, 256)
bias = torch.zeros(256, device=dev)
t = timed(lambda: fused_leaky_relu(x, bias));  print(f'fused_bias_act (read 1 + write 1): {t:7.3f} ms  {2 * gib / t:6.2f} TB/s')
