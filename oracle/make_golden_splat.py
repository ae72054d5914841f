"""Golden vectors for splat2d from the REFERENCE KERNEL itself.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Runs on a GPU box (the authoring container has no GPU):

    gpurun -- 'python oracle/make_golden_splat.py gpurun_out/splat2d.npz'     # then: cp gpurun_out/splat2d.npz tests/golden/

oracle/_ref/libsplat_ref.so is utils/splat2d_cuda/src/splat_gpu_impl.cu compiled unmodified (oracle/Makefile); this
script calls its C entry point SplatForwardGpu (splat_gpu_impl.cuh:11-22) through ctypes on torch device buffers and
applies the binding's few lines around it (splat_gpu.c:19-40: output = clone(input), alpha = zeros, launch,
alpha.clamp(1) when soft_normalize, output / (alpha + 1e-8)) - that file itself needs <THC/THC.h> and cannot be built
against this torch.  Cases follow SURVEY.md section 8c (v): P in {1, 17, 4096}, sigma in {0.3, 1.3, 3}, points on the
border, out of bounds, duplicated; zero and non-zero `input`; both normalisations.
"""
import ctypes
import json
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(REPO, 'oracle', '_ref', 'libsplat_ref.so')

_lib = None


def reference_available():
    return os.path.exists(REF_SO)


def reference_splat2d(input, coordinates, values, sigma, soft_normalize=False):
    """splat_forward_cuda (splat_gpu.c:12-42) on top of the reference's compiled SplatForwardGpu."""
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(REF_SO)
        _lib.SplatForwardGpu.restype = None
        _lib.SplatForwardGpu.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int] * 5
    n, c, h, w = input.shape
    p = coordinates.shape[1]
    coordinates, values, sigma = coordinates.contiguous(), values.contiguous(), sigma.contiguous()
    alpha = torch.zeros((n, h, w), dtype=torch.float32, device=input.device)
    output = input.clone().contiguous()
    if output.numel() == 0:
        return output
    stream = torch.cuda.current_stream().cuda_stream
    _lib.SplatForwardGpu(ctypes.c_void_p(stream), ctypes.c_void_p(coordinates.data_ptr()),
                         ctypes.c_void_p(values.data_ptr()), ctypes.c_void_p(sigma.data_ptr()),
                         ctypes.c_void_p(alpha.data_ptr()), ctypes.c_void_p(output.data_ptr()), p, c, h, w, n * p)
    alpha = alpha.view(n, 1, h, w)
    if soft_normalize:
        alpha = alpha.clamp(1.0)
    return output / (alpha + 1e-8)


def reference_splat_points(images, points, sigma, opacity, colors, alpha_channel=None):
    """splat_points (utils/vis_tools/helpers.py:134-194) with explicit colours and alpha compositing, on the reference
    kernel: the lines around its two splat2d calls (:181-186) restated."""
    n = images.size(0)
    if points.dim() == 4:
        points = points.reshape(points.size(0), points.size(1) * points.size(2), 2)
    if alpha_channel is None:
        alpha_channel = torch.ones(n, points.size(1), 1, device=images.device)
    if isinstance(sigma, (float, int)):
        sigma = torch.tensor(sigma, device=images.device, dtype=torch.float).view(1).repeat(n)
    blank_img = torch.zeros_like(images)
    blank_mask = torch.zeros(n, 1, images.size(2), images.size(3), device=images.device)
    obj = reference_splat2d(blank_img, points, colors, sigma, False)
    mask = reference_splat2d(blank_mask, points, alpha_channel, sigma, True) * opacity
    return mask * obj + (1 - mask) * images


def make_cases():
    specs = []
    ci = 0
    for p in (1, 17, 4096):
        for sig in (0.3, 1.3, 3.0):
            for soft in (False, True):
                n, c, h, w = (2, 3, 24, 40) if p < 4096 else (1, 2, 64, 64)
                specs.append((ci, n, c, h, w, p, sig, soft, ci % 4 == 3))
                ci += 1
    cases = []
    for ci, n, c, h, w, p, sig, soft, nonzero_input in specs:
        rs = np.random.RandomState(1000 + ci)
        coords = (rs.rand(n, p, 2) * [w + 6, h + 6] - 3).astype(np.float32)        # a share falls out of bounds
        coords[:, 0] = [0.0, 0.0]                                                    # exactly on the border
        if p > 4:
            coords[:, 1] = [w - 1.0, h - 1.0]                                        # last in-bounds integer position
            coords[:, 2] = [np.nextafter(np.float32(w), np.float32(0)), 0.5]         # just inside the right edge
            coords[:, 3] = [float(w), 0.5]                                           # x == width: skipped (:76)
            coords[:, 4] = coords[:, 5]                                              # duplicate
        values = rs.randn(n, p, c).astype(np.float32)
        sigma = np.full((n,), sig, dtype=np.float32)
        if n > 1:
            sigma[1] = sig * 1.5                                                     # per-sample sigma
        inp = rs.randn(n, c, h, w).astype(np.float32) if nonzero_input else np.zeros((n, c, h, w), np.float32)
        cases.append(dict(input=inp, coords=coords, values=values, sigma=sigma, soft=soft))
    return cases


def main(out_path):
    assert torch.cuda.is_available() and reference_available(), 'needs a GPU and oracle/_ref/libsplat_ref.so'
    dev = torch.device('cuda:0')
    flat = {}
    for ci, c in enumerate(make_cases()):
        t = lambda a: torch.from_numpy(a).to(dev)
        out = reference_splat2d(t(c['input']), t(c['coords']), t(c['values']), t(c['sigma']), c['soft'])
        torch.cuda.synchronize()
        for k in ('input', 'coords', 'values', 'sigma'):
            flat[f'case{ci:02d}/{k}'] = c[k]
        flat[f'case{ci:02d}/out'] = out.cpu().numpy()
        meta = dict(soft_normalize=c['soft'], source='reference-kernel',
                    recipe='oracle/Makefile + oracle/make_golden_splat.py on MI355X (gfx950)')
        flat[f'case{ci:02d}/meta'] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    np.savez_compressed(out_path, **flat)
    print(f'{out_path}: {len(flat) // 6} cases, {os.path.getsize(out_path) / 1024:.1f} KiB')


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, 'gpurun_out', 'splat2d.npz'))
