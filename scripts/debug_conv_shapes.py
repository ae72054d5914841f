"""conv2d / conv3x3_bias_act forward + input gradient at VGG-like small shapes vs float64 torch."""
import sys; sys.path.insert(0, '/root/repo')
import torch, torch.nn.functional as F
from gangealing_amd.op import conv_mfma
dev = torch.device('cuda:0')
shapes = [(6, 64, 64, 64), (6, 64, 128, 32), (6, 128, 128, 32), (6, 128, 256, 16), (6, 256, 256, 16), (6, 256, 512, 8),
          (6, 512, 512, 8), (6, 512, 512, 4), (32, 128, 256, 32), (32, 512, 512, 8), (2, 128, 256, 16), (6, 128, 256, 8)]
for mode in ('fp32', 'bf16x3'):
    conv_mfma.set_precision(mode)
    for (n, cin, cout, hw) in shapes:
        g = torch.Generator().manual_seed(n * 1000 + cin + hw)
        x = torch.randn(n, cin, hw, hw, generator=g)
        w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
        b = torch.randn(cout, generator=g) * 0.05
        dy = torch.randn(n, cout, hw, hw, generator=g)
        xr = x.double().requires_grad_(True)
        yr = F.relu(F.conv2d(xr, w.double(), b.double(), padding=1))
        (gxr,) = torch.autograd.grad(yr, xr, dy.double())
        res = []
        for name in ('bias_act', 'conv2d+relu'):
            xd = x.to(dev).requires_grad_(True)
            wd = w.to(dev).requires_grad_(False)
            if name == 'bias_act':
                y = conv_mfma.conv3x3_bias_act(xd, wd, b.to(dev), 0.0, 1.0)
            else:
                y = F.relu(conv_mfma.conv2d(xd, wd, b.to(dev), stride=1, padding=1))
            (gx,) = torch.autograd.grad(y, xd, dy.to(dev))
            e_f = float((y.detach().cpu().double() - yr.detach()).abs().max() / yr.abs().max())
            e_b = float((gx.cpu().double() - gxr).abs().max() / gxr.abs().max())
            res.append(f'{name}: fwd {e_f:.1e} dgrad {e_b:.1e}')
        print(mode, (n, cin, cout, hw), ' | '.join(res))
