"""splat2d stress benchmark (BASELINE.md section 2, config C4 note): P = 1e6 points onto 512^2 and 1024^2 canvases,
sigma in {0.3, 1.3}, C = 3 value channels, the HIP kernel next to the reference's own kernel compiled unmodified
(oracle/_ref/libsplat_ref.so, when it travelled).  Reports per case: time per call (HIP events, median of `reps`),
points/s, and the achieved rate against the ALGORITHMIC traffic 4*P*(3+C) bytes of point data read +
4*(C+1)*H*W bytes of canvas read-modify-write (each atomic touches a 4-byte word that lives in L2: this is an
L2-atomic-bound operator, so the HBM figure is a lower bound on what the atomic units do: atomics issued =
P * footprint * (C+1)).  Round 4: the HIP operator bins the points per 32x32 tile and gathers (no float atomics);
`atomics_issued_upper_bound` now describes the reference formulation only, and a bitwise-repeat check is added.

    python scripts/splat_bench.py [out.json]
"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch                                                  # noqa: E402

from gangealing_amd.splat2d_cuda import splat2d               # noqa: E402


def timed(fn, reps=7):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


def main():
    dev = torch.device('cuda', 0)
    try:
        from oracle.make_golden_splat import reference_available, reference_splat2d      # checker only
        have_ref = reference_available()
    except Exception:                                          # noqa: BLE001
        have_ref = False
    rows = []
    P, C = 1_000_000, 3
    g = torch.Generator(device='cpu').manual_seed(0)
    for size in (512, 1024):
        for sigma in (0.3, 1.3):
            coords = (torch.rand(1, P, 2, generator=g) * size).to(dev)
            values = torch.randn(1, P, C, generator=g).to(dev)
            canvas = torch.zeros(1, C, size, size, device=dev)
            sig = torch.full((1,), sigma, device=dev)
            ms = timed(lambda: splat2d(canvas, coords, values, sig, False))
            foot = (2 * int(-(-2 * sigma // 1)) + 1) ** 2     # (2*ceil(2 sigma)+1)^2 pixels at most
            alg_bytes = 4 * P * (3 + C) + 2 * 4 * (C + 1) * size * size
            row = dict(canvas=size, sigma=sigma, points=P, channels=C, ms=round(ms, 4),
                       mpoints_per_s=round(P / ms / 1e3, 1), algorithmic_MB=round(alg_bytes / 1e6, 2),
                       algorithmic_GBps=round(alg_bytes / ms / 1e6, 1),
                       atomics_issued_upper_bound=P * foot * (C + 1),
                       gatomics_per_s=round(P * foot * (C + 1) / ms / 1e6, 2))
            if have_ref:
                ref_ms = timed(lambda: reference_splat2d(canvas, coords, values, sig, False))
                out = splat2d(canvas, coords, values, sig, False)
                ref = reference_splat2d(canvas, coords, values, sig, False)
                row.update(reference_kernel_ms=round(ref_ms, 4), speedup_vs_reference_kernel=round(ref_ms / ms, 2),
                           max_abs_diff_vs_reference_kernel=float((out - ref).abs().max()),
                           ref_scale=float(ref.abs().max()))
            row['bitwise_repeatable'] = bool(torch.equal(splat2d(canvas, coords, values, sig, False),
                                                         splat2d(canvas, coords, values, sig, False)))
            rows.append(row)
            print(json.dumps(row), flush=True)
    if len(sys.argv) > 1:
        with open(sys.argv[1], 'w') as f:
            json.dump(dict(device=torch.cuda.get_device_name(0), rows=rows), f, indent=1)


if __name__ == '__main__':
    main()
