import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
from conftest import load_golden
from gangealing_amd.spatial_transformers.antialiased_sampling import MipmapWarp
cuda = torch.device('cuda:0')
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
for case in load_golden('mipmap_warp'):
    m = case['meta']
    x, grid = T(case['x']).requires_grad_(True), T(case['grid']).requires_grad_(True)
    w = MipmapWarp(3.5).to(cuda)
    out = w(x, grid, padding_mode=m['padding_mode'])
    out.backward(T(case['g']))
    rel = lambda a, b: float(np.linalg.norm(a.detach().cpu().numpy().astype(np.float64) - b) / max(np.linalg.norm(b), 1e-30))
    print(f"{case['x'].shape[-1]:3d} {m['padding_mode']:10s} {m['grid']:15s} out {rel(out, case['out']):.1e} ggrid {rel(grid.grad, case['ggrid']):.1e} gx {rel(x.grad, case['gx']):.1e} lv[{case['levels_map'].min()*2.5:.2f},{case['levels_map'].max()*2.5:.2f}]")
