"""Where does the VGG/LPIPS input gradient lose accuracy?  Per-layer gradient comparison HIP vs a float64 torch run."""
import sys; sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch, torch.nn as nn, torch.nn.functional as F
from conftest import load_golden
from oracle import config_cases as cc
from gangealing_amd import losses
from gangealing_amd.op import conv_mfma

def run(device, dtype, emulate, mode='fp32'):
    conv_mfma.set_precision(mode)
    case = load_golden('lpips')[0]
    net = losses.LPIPS(net='vgg', lpips=False, pnet_rand=True, pretrained=False)
    torch.nn.Module.load_state_dict(net, cc.det_lpips_state_dict(net), strict=False)
    net = net.to(device).to(dtype).eval()
    grads = {}
    in0 = torch.from_numpy(case['in0']).to(device).to(dtype).requires_grad_(True)
    in1 = torch.from_numpy(case['in1']).to(device).to(dtype)
    x = net.scaling_layer(torch.cat([in0, in1], 0))
    feats = []
    li = 0
    for si in range(5):
        for mod in getattr(net.net, f'slice{si + 1}'):
            if isinstance(mod, nn.Conv2d):
                x.register_hook(lambda g, k=f'in_conv{li}': grads.__setitem__(k, g.detach().double().cpu()))
                if emulate:
                    x = F.relu(F.conv2d(x, mod.weight, mod.bias, padding=1))
                elif x.shape[1] % 32 == 0:
                    x = conv_mfma.conv3x3_bias_act(x, mod.weight, mod.bias, 0.0, 1.0)
                else:
                    x = F.relu(conv_mfma.conv2d(x, mod.weight, mod.bias, stride=1, padding=1))
                li += 1
            elif isinstance(mod, nn.MaxPool2d):
                x = F.max_pool2d(x, 2, 2)
        feats.append(x)
    val = 0
    for kk, f in enumerate(feats):
        f.register_hook(lambda g, k=f'tap{kk}': grads.__setitem__(k, g.detach().double().cpu()))
        if emulate:
            fn = f / (torch.sqrt(torch.sum(f ** 2, dim=1, keepdim=True)) + 1e-10)
            d = (fn[:3] - fn[3:]) ** 2
            val = val + d.sum(dim=1, keepdim=True).mean(dim=(2, 3), keepdim=True)
        else:
            val = val + losses.lpips_tail(f).view(3, 1, 1, 1)
    val.backward(torch.from_numpy(case['g']).to(device).to(dtype))
    grads['in0'] = in0.grad.detach().double().cpu()
    return grads

ref = run('cpu', torch.float64, True)
for mode in ('fp32', 'bf16x3'):
    got = run('cuda', torch.float32, False, mode)
    emu = run('cuda', torch.float32, True, mode)      # torch ops on the GPU in float32 (MIOpen convs)
    print('mode', mode)
    for k in sorted(ref, key=lambda s: (s[:2] != 'ta', -int(''.join(c for c in s if c.isdigit()) or 0))):
        s = float(ref[k].abs().max())
        print(f'  {k:10s} hip err {float((got[k] - ref[k]).abs().max()) / s:.2e}   torch-fp32-gpu err {float((emu[k] - ref[k]).abs().max()) / s:.2e}')
