import os, sys, time
sys.path.insert(0, '/root/repo')
import torch
from gangealing_amd.op import upfirdn2d
import numpy as np
k = torch.tensor(np.outer([1, 3, 3, 1], [1, 3, 3, 1]) / 64.0 * 4, dtype=torch.float32, device='cuda')
knon = k.clone(); knon[1, 2] += 0.01
for (shape, pad) in [((16, 128, 257, 257), (1, 1)), ((16, 256, 129, 129), (1, 1)), ((16, 128, 256, 256), (2, 2))]:
    x = torch.randn(*shape, device='cuda')
    for name, kk in (('separable', k), ('general', knon)):
        for _ in range(3):
            y = upfirdn2d(x, kk, pad=pad)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            y = upfirdn2d(x, kk, pad=pad)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        gb = (x.numel() + y.numel()) * 4 / 1e9
        print(shape, pad, name, f'{dt * 1e6:.0f} us  {gb / dt / 1e3:.2f} TB/s')
