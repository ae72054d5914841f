import os, sys
sys.path.insert(0, '/root/repo')
import torch
from gangealing_amd import _lib
from gangealing_amd.op import conv_mfma as cm
from gangealing_amd.op.upfirdn2d import blur_noise_act, blur_bits_words, blur_bits_unpack
from gangealing_amd.stylegan2.networks import make_kernel
dev = torch.device('cuda', 0)
torch.manual_seed(0)
n, c, h, w = 2, 8, 129, 129
k = (make_kernel([1, 3, 3, 1]) * 4).to(dev)
x = torch.randn(n, c, h, w, device=dev)
oh, ow = h - 1, w - 1
noise = torch.randn(n, 1, oh, ow, device=dev)
nw = torch.tensor([0.3], device=dev)
b = torch.randn(c, device=dev) * 0.2
g = torch.randn(n, c, oh, ow, device=dev)
out = torch.empty(n, c, oh, ow, device=dev)
bits = torch.full((n * c, blur_bits_words(oh, ow)), -1, dtype=torch.int32, device=dev)
_lib.call('gg_blur4_fused_bits_f32', out, x, k, n, c, h, w, 1, 1, 1, 1, noise, nw, b, bits, 0.2, 2 ** 0.5)
u = blur_bits_unpack(bits, oh, ow)
ref = (out > 0).reshape(n * c, oh, ow)
print('plane mismatches', int((u != ref).sum()), 'of', ref.numel())
bad = (u != ref).nonzero()
print(bad[:10])
from gangealing_amd.op.upfirdn2d import _flipped
dx1 = torch.empty(n, c, 129, 129, device=dev); dx2 = torch.empty_like(dx1)
_lib.call('gg_blur4_fused_f32', dx1, g, _flipped(k), n, c, oh, ow, 2, 2, 2, 2, None, None, None, out, 0.2, 2 ** 0.5)
_lib.call('gg_blur4_fused_bits_f32', dx2, g, _flipped(k), n, c, oh, ow, 2, 2, 2, 2, None, None, None, bits, 0.2, 2 ** 0.5)
d = (dx1 != dx2)
print('grad mismatches', int(d.sum()), 'of', d.numel())
print(d.nonzero()[:12])
print(d.any(dim=0).any(dim=0).nonzero()[:, 0].unique()[:40], d.any(dim=0).any(dim=0).nonzero()[:, 1].unique()[:40])

print('max abs diff', float((dx1 - dx2).abs().max()), 'max |dx|', float(dx1.abs().max()))
