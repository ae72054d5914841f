"""Known-traffic kernels for calibrating rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950:
fused_bias_act on a 512 MiB tensor reads 512 MiB + writes 512 MiB (16 B per lane)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gangealing_amd.op import fused_leaky_relu
x = torch.randn(16, 128, 256, 256, device='cuda')
b = torch.randn(128, device='cuda')
for _ in range(3):
    y = fused_leaky_relu(x, b)
torch.cuda.synchronize()
print(x.numel() * 4 / 2**20, 'MiB in, same out')
