#!/usr/bin/env python
"""Copy the summaries of one scripts/measure.sh session (gpurun_out/<round>/) into profiles/ under their round names,
stamp each with the commit / kernel-source hash the session ran at, and rebuild profiles/<round>_pmc_traffic.json from the
FETCH_SIZE / WRITE_SIZE passes and the calibration kernel (bench.py only trusts that file while the recorded
kernel-source hash equals the tree's)."""
import json
import os
import re
import shutil
import subprocess

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = os.environ.get('ROUND', 'r06')
O = os.path.join(R, 'gpurun_out', ROUND)
P = os.path.join(R, 'profiles')
DOM = 'conv3x3_patch_kernel<2, true, 256, 2, 0,'        # (MASK = 0: the plain tile; round 4's bool printed as false)


def rows(path):
    out = {}
    for line in open(os.path.join(O, path)):
        m = re.match(r'(.*?)\s+(\w+)\s+n=\s*(\d+)\s+mean=\s*([\d.]+)', line)
        if m:
            out[(m.group(1).strip(), m.group(2))] = (int(m.group(3)), float(m.group(4)))
    return out


def find(table, needle, counter):
    hits = [(k, v) for k, v in table.items() if needle in k[0] and k[1] == counter]
    assert len(hits) == 1, (needle, counter, hits)
    return hits[0][1]


sha = open(os.path.join(O, 'kernel_source_sha16.txt')).read().strip()
commit = open(os.path.join(O, 'commit.txt')).read().strip()
if 'no-git' in commit:        # the GPU box receives a snapshot without .git: the session ran at the tree it was sent from
    commit = subprocess.run(['git', 'rev-parse', 'HEAD'], cwd=R, capture_output=True, text=True).stdout.strip() + \
        ' (HEAD of the authoring tree when the snapshot was sent)'
STAMP = f'# measured by scripts/measure.sh at commit {commit}; kernel-source sha16 {sha}\n'


def stamped(src, dst, header=''):
    with open(os.path.join(P, dst), 'w') as f:
        f.write(STAMP + header + open(os.path.join(O, src)).read())


for src, dst in ((f'parity_{ROUND}.json', f'parity_{ROUND}.json'), ('bench_fp16x3.json', f'bench_{ROUND}_fp16x3.json'), ('bench_bf16x3.json', f'bench_{ROUND}_bf16x3.json'),
                 ('bench_fp32.json', f'bench_{ROUND}_fp32.json'), ('bench_bf16.json', f'bench_{ROUND}_bf16.json'),
                 ('bench_fp16x3_batch5.json', f'bench_{ROUND}_fp16x3_batch5.json'),
                 ('bench_fp16x3_batch5_hipgraph.json', f'bench_{ROUND}_fp16x3_batch5_hipgraph.json'),
                 ('bench_fp16x3_batch32.json', f'bench_{ROUND}_fp16x3_batch32.json'),
                 ('bench_c4_batch4.json', f'bench_{ROUND}_c4_batch4.json'), ('bench_c4_batch16.json', f'bench_{ROUND}_c4_batch16.json'),
                 ('bench_c5_batch4.json', f'bench_{ROUND}_c5_batch4.json'), ('bench_c5_batch16.json', f'bench_{ROUND}_c5_batch16.json'),
                 ('bench_c4_batch4_hipgraph.json', f'bench_{ROUND}_c4_batch4_hipgraph.json'),
                 ('bench_c5_batch4_hipgraph.json', f'bench_{ROUND}_c5_batch4_hipgraph.json'),
                 ('bench_c2_under_rocprofv3.json', f'bench_{ROUND}_fp16x3_under_rocprofv3.json'),
                 ('bench_c4_under_rocprofv3.json', f'bench_{ROUND}_c4_batch16_under_rocprofv3.json'),
                 ('bench_c5_under_rocprofv3.json', f'bench_{ROUND}_c5_batch16_under_rocprofv3.json'),
                 ('smoke.txt', f'{ROUND}_smoke.txt'),
                 ('splat_bench.json', f'{ROUND}_splat_bench.json')):
    if os.path.exists(os.path.join(O, src)):
        shutil.copy(os.path.join(O, src), os.path.join(P, dst))

under = json.loads(open(os.path.join(O, 'bench_c2_under_rocprofv3.json')).read().strip().splitlines()[-1])
stats = open(os.path.join(O, 'kernel_stats_c2.txt')).read()
dom_line = next(l for l in stats.splitlines() if DOM in l)
dom_avg = float(dom_line.split()[2])
stamped('kernel_stats_c2.txt', f'{ROUND}_a_kernel_stats.txt',
        '# rocprofv3 --kernel-trace --output-format rocpd -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline '
        '--no-extras   (15 steps traced: 3 warm-up, the 2 steps of the kernel survey, 10 timed; per-step = total/15)\n'
        f'# bench line printed by the same run: profiles/bench_{ROUND}_fp16x3_under_rocprofv3.json '
        f'(roofline.avg_launch_ms {under["roofline"]["avg_launch_ms"]} vs {dom_avg} us below)\n')
for w in ('c4', 'c5'):
    stamped(f'kernel_stats_{w}.txt', f'{ROUND}_a_kernel_stats_{w}.txt',
            f'# rocprofv3 --kernel-trace -- python bench.py --workload {w} --batch 16 --steps 10 --warmup 3 --no-cpu-baseline '
            f'--batch 16 --no-extras (per-GPU batch 16; 15 steps traced)\n')
stamped('blur_bench.txt', f'{ROUND}_d_blur_bench.txt')
if os.path.exists(os.path.join(O, 'splat_kernel_stats.txt')):
    stamped('splat_kernel_stats.txt', f'{ROUND}_g_splat_kernel_stats.txt', '# rocprofv3 --kernel-trace -- python scripts/splat_bench.py (the reference kernel SplatForward runs in the same process as the checker)\n')
if os.path.exists(os.path.join(O, 'pytest_reference.txt')):
    stamped('pytest_reference.txt', f'{ROUND}_pytest_reference_dropin.txt', '# python -m pytest tests/test_gpu_reference_dropin.py tests/test_gpu_rccl_single_rank.py -q -m gpu (tail)\n')
stamped('conv_layers.txt', f'{ROUND}_f_conv_layers.txt', '# scripts/conv_bench.py, fp16x3 (forward launches: binary16 limbs; the wgrad column: bf16 limbs), batch 16, ITERS=20\n')
if os.path.exists(os.path.join(O, 'conv_layers_bf16x3.txt')):
    stamped('conv_layers_bf16x3.txt', f'{ROUND}_f_conv_layers_bf16x3.txt', '# scripts/conv_bench.py "G ", bf16x3, batch 16, ITERS=20\n')
if os.path.exists(os.path.join(O, 'timeline_c2.txt')):
    stamped('timeline_c2.txt', f'{ROUND}_j_step_timeline_under_rocprofv3.txt',
            '# one training step of the trace behind the kernel_stats file of the same session in launch order (scripts/rocpd_timeline.py): start, '
            'duration, idle gap before the launch.  Under rocprofv3 the host is slower than the GPU (gaps of 5 - 11 us in front of '
            'most library launches); untraced, the step is GPU-bound (bench: kernel time ~ step time)\n')
stamped('determinism.txt', f'{ROUND}_determinism.txt',
        '# scripts/check_determinism.py: two runs of two training iterations from the same seeds, compared bit for bit\n')
stamped('pytest_gpu.txt', f'{ROUND}_pytest_gpu.txt', '# python -m pytest tests -m gpu -q (tail)\n')

OLD_SPLAT = ''
_prev = os.path.join(P, f'{ROUND}_b_pmc_hbm_traffic.txt')
if os.path.exists(_prev):
    _t = open(_prev).read()
    _i = _t.find('# splat2d stress')
    OLD_SPLAT = _t[_i:] if _i >= 0 else ''
cal = {**rows('cal_fetch.txt'), **rows('cal_write.txt')}
cal_f = find(cal, 'fused_bias_act_kernel', 'FETCH_SIZE')[1]
cal_w = find(cal, 'fused_bias_act_kernel', 'WRITE_SIZE')[1]
fetch, write = rows('pmc_fetch.txt'), rows('pmc_write.txt')
with open(os.path.join(P, f'{ROUND}_b_pmc_hbm_traffic.txt'), 'w') as f:
    f.write(STAMP + '# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --steps 2 '
            '--warmup 1 --no-cpu-baseline --no-extras\n# mean counter value per kernel (KB); calibration kernel (512 MiB in, '
            '512 MiB out):\n')
    for k, (n, v) in sorted(cal.items()):
        if 'fused_bias_act_kernel' in k[0]:
            f.write(f'{k[0]:70s} {k[1]:28s} n={n:4d} mean={v:16.1f}\n')
    f.write('# kernels of the step:\n')
    for k, (n, v) in sorted({**fetch, **write}.items()):
        f.write(f'{k[0]:70s} {k[1]:28s} n={n:4d} mean={v:16.1f}\n')
    fresh = [open(os.path.join(O, name)).read() for name in ('splat_fetch.txt', 'splat_write.txt')
             if os.path.exists(os.path.join(O, name)) and os.path.getsize(os.path.join(O, name)) > 0]
    if fresh:
        f.write('# splat2d stress (scripts/splat_bench.py under the same two passes):\n' + ''.join(fresh))
    else:          # a QUICK session skipped the (unchanged) splat2d passes: keep the last full session's lines
        f.write(OLD_SPLAT)
stamped('pmc_sq.txt', f'{ROUND}_c_pmc_sq.txt', '# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -- python bench.py '
        '--steps 2 --warmup 1 --no-cpu-baseline --no-extras\n')

fn, fv = find(fetch, DOM, 'FETCH_SIZE')
wn, wv = find(write, DOM, 'WRITE_SIZE')
scale_f = 512 * 1024 / cal_f
fetch_b, write_b = int(fv * 1024 * round(scale_f)), int(wv * 1024)
json.dump({
    'kernel': 'conv3x3_patch_kernel<2, true, 256, 2, 0, 1, true>', 'precision': 'fp16x3', 'workload': 'c2', 'batch': 16,
    'kernel_source_sha16': sha, 'commit': commit,
    'command': 'rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) --output-format csv -- python bench.py --steps 2 '
               '--warmup 1 --no-cpu-baseline --no-extras',
    'launches_averaged': fn, 'FETCH_SIZE_KB_per_launch': fv, 'WRITE_SIZE_KB_per_launch': wv,
    'calibration': f'fused_bias_act on a 512 MiB tensor (scripts/pmc_calibrate.py) in the same session: FETCH_SIZE {cal_f} KB '
                   f'= 1/{round(scale_f)} of the 512 MiB read (gfx950 under-count of 16 B/lane streaming reads, '
                   f'MI355X_MICROARCH.md HBM section) -> x{round(scale_f)}; WRITE_SIZE {cal_w} KB = exact',
    'fetch_bytes_per_launch': fetch_b, 'write_bytes_per_launch': write_b, 'hbm_bytes_per_launch': fetch_b + write_b,
    'algorithmic_bytes_per_launch': {
        'activations_in': 313174698, 'activations_out': 313174698,
        'note': 'mean over the 6 forward launches per step: 512->512@64^2, 256->256@128^2, 128->128@256^2 (x2 generator '
                'passes), batch 16, fp32'},
    'reading': f'reads = {fetch_b / 313174698:.2f}x the input tensor (tile halos), writes = {write_b / 313174698:.2f}x the output',
}, open(os.path.join(P, f'{ROUND}_pmc_traffic.json'), 'w'), indent=1)
print(open(os.path.join(P, f'{ROUND}_pmc_traffic.json')).read())

# C4 / C5 (round 6): HBM traffic of the kernel that tops the workload's own kernel-stats table
for w in ('c4', 'c5'):
    if not os.path.exists(os.path.join(O, f'pmc_fetch_{w}.txt')):
        continue
    try:
        _collect_w = True
        top = next(l for l in open(os.path.join(O, f'kernel_stats_{w}.txt')).read().splitlines()
                   if re.match(r'\s*\d+\s+[\d.]+\s+[\d.]+', l))
        name = top.split(None, 6)[6].strip()
        needle = name[:70].strip()
        fw, ww = rows(f'pmc_fetch_{w}.txt'), rows(f'pmc_write_{w}.txt')
        fn2, fv2 = find(fw, needle, 'FETCH_SIZE')
        wn2, wv2 = find(ww, needle, 'WRITE_SIZE')
        fb, wb = int(fv2 * 1024 * round(scale_f)), int(wv2 * 1024)
        json.dump({'kernel': name[:140], 'precision': 'fp16x3', 'workload': w, 'batch': 16, 'kernel_source_sha16': sha,
                   'commit': commit,
                   'command': f'rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) --output-format csv -- python bench.py '
                              f'--workload {w} --batch 16 --steps 2 --warmup 1 --no-cpu-baseline --no-extras',
                   'launches_averaged': fn2, 'FETCH_SIZE_KB_per_launch': fv2, 'WRITE_SIZE_KB_per_launch': wv2,
                   'calibration': f'as profiles/{ROUND}_pmc_traffic.json (same session): FETCH_SIZE x{round(scale_f)}, WRITE_SIZE exact',
                   'fetch_bytes_per_launch': fb, 'write_bytes_per_launch': wb, 'hbm_bytes_per_launch': fb + wb,
                   'note': 'mean over every launch of the kernel instantiation with the largest share of the workload\'s GPU '
                           f'time (profiles/{ROUND}_a_kernel_stats_{w}.txt, first row)'},
                  open(os.path.join(P, f'{ROUND}_pmc_traffic_{w}.json'), 'w'), indent=1)
        print(w, name[:90], fb + wb)
    except Exception as e:            # noqa: BLE001 - the C2 record must not depend on these
        print(f'{w}: traffic record not written ({e})')
