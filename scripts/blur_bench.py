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

# the generic kernel's hot shapes in C2: the ResBlock skip's blur + stride-2 read-out (forward: down 2; its adjoint: up 2)
# and the ToRGB skip up-sampling (up 2; adjoint: down 2)
def timed(fn, iters=20):
    for _ in range(3):
        y = fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        y = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters, y

k1 = torch.tensor(np.outer([1, 3, 3, 1], [1, 3, 3, 1]) / 64.0, dtype=torch.float32, device='cuda')
for (shape, up, down, pad, what) in [((16, 64, 128, 128), 1, 2, (1, 1), 'STN skip 128->64'),
                                     ((16, 128, 64, 64), 1, 2, (1, 1), 'STN skip 64->32'),
                                     ((16, 512, 32, 32), 1, 2, (1, 1), 'STN skip 32->16'),
                                     ((16, 64, 64, 64), 2, 1, (2, 1), 'adjoint of STN skip 128->64'),
                                     ((16, 128, 32, 32), 2, 1, (2, 1), 'adjoint of STN skip 64->32'),
                                     ((16, 3, 128, 128), 2, 1, (2, 1), 'ToRGB skip 128->256'),
                                     ((16, 512, 17, 17), 1, 1, (1, 1), 'G blur 17->16 (small plane)')]:
    x = torch.randn(*shape, device='cuda')
    kk = k1 * (up * up)
    dt, y = timed(lambda: upfirdn2d(x, kk, up=up, down=down, pad=pad))
    gb = (x.numel() + y.numel()) * 4 / 1e9
    print(f'direct: {what:30s} {tuple(x.shape)} -> {tuple(y.shape)} {dt * 1e6:7.1f} us  {gb / dt / 1e3:.2f} TB/s')
