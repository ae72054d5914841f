#!/usr/bin/env python
"""Copy the summaries of one scripts/measure_r02.sh session (gpurun_out/r02/) into profiles/ under their round-2 names and
rebuild profiles/r02_pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE passes and the calibration kernel."""
import json
import os
import re
import shutil

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(R, 'gpurun_out', 'r02')
P = os.path.join(R, 'profiles')
DOM = 'conv3x3_patch_kernel<2, true, 256, 2, false, 1>'


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


for src, dst in (('parity_r02.json', 'parity_r02.json'), ('bench_bf16x3.json', 'bench_r02_bf16x3.json'),
                 ('bench_fp32.json', 'bench_r02_fp32.json'), ('bench_bf16x6.json', 'bench_r02_bf16x6.json'),
                 ('bench_bf16.json', 'bench_r02_bf16.json'), ('bench_bf16x3_batch5.json', 'bench_r02_bf16x3_batch5.json'),
                 ('bench_bf16x3_batch32.json', 'bench_r02_bf16x3_batch32.json'),
                 ('bench_under_rocprofv3.json', 'bench_r02_bf16x3_under_rocprofv3.json')):
    shutil.copy(os.path.join(O, src), os.path.join(P, dst))

under = json.loads(open(os.path.join(O, 'bench_under_rocprofv3.json')).read().strip().splitlines()[-1])
stats = open(os.path.join(O, 'kernel_stats.txt')).read()
dom_line = next(l for l in stats.splitlines() if DOM in l)
dom_avg = float(dom_line.split()[2])
with open(os.path.join(P, 'r02_a_kernel_stats.txt'), 'w') as f:
    f.write('# rocprofv3 --kernel-trace --output-format rocpd -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline '
            '--no-extras   (13 steps traced; per-step = total/13)\n')
    f.write(f'# bench line printed by the same run: profiles/bench_r02_bf16x3_under_rocprofv3.json '
            f'(roofline.avg_launch_ms {under["roofline"]["avg_launch_ms"]} vs {dom_avg} us below)\n')
    f.write(stats)

cal = {**rows('cal_fetch.txt'), **rows('cal_write.txt')}
cal_f = find(cal, 'fused_bias_act_kernel', 'FETCH_SIZE')[1]
cal_w = find(cal, 'fused_bias_act_kernel', 'WRITE_SIZE')[1]
fetch, write = rows('pmc_fetch.txt'), rows('pmc_write.txt')
with open(os.path.join(P, 'r02_b_pmc_hbm_traffic.txt'), 'w') as f:
    f.write('# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --steps 2 --warmup 1 '
            '--no-cpu-baseline --no-extras\n# mean counter value per kernel (KB); calibration kernel (512 MiB in, 512 MiB out):\n')
    for k, (n, v) in sorted(cal.items()):
        if 'fused_bias_act_kernel' in k[0]:
            f.write(f'{k[0]:70s} {k[1]:28s} n={n:4d} mean={v:16.1f}\n')
    f.write('# kernels of the step:\n')
    merged = {**fetch, **write}
    for k, (n, v) in sorted(merged.items()):
        f.write(f'{k[0]:70s} {k[1]:28s} n={n:4d} mean={v:16.1f}\n')
shutil.copy(os.path.join(O, 'blur_bench.txt'), os.path.join(P, 'r02_d_blur_bench.txt'))
with open(os.path.join(P, 'r02_c_pmc_sq.txt'), 'w') as f:
    f.write('# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -- python bench.py --steps 2 --warmup 1 '
            '--no-cpu-baseline --no-extras\n')
    f.write(open(os.path.join(O, 'pmc_sq.txt')).read())
with open(os.path.join(P, 'r02_e_sq_stall_attribution.txt'), 'w') as f:
    f.write('# rocprofv3 --pmc <SQ group> -- python scripts/conv_bench.py "G conv 64"   (512 -> 512 @64^2, batch 16, bf16x3; '
            'three passes)\n# SQ_WAVE_CYCLES ~ SQ_WAIT_ANY (parked on s_waitcnt / s_barrier) + SQ_WAIT_INST_ANY (issue stall: '
            'matrix pipe busy / RAW) + SQ_ACTIVE_INST_ANY, in units of 4 cycles; SQ_VALU_MFMA_BUSY_CYCLES in cycles over '
            '1024 SIMDs; GRBM_GUI_ACTIVE summed over the 8 XCDs\n')
    f.write(open(os.path.join(O, 'sq_stall.txt')).read())

fn, fv = find(fetch, DOM[:40], 'FETCH_SIZE') if False else find(fetch, 'conv3x3_patch_kernel<2, true, 256, 2, fals', 'FETCH_SIZE')
wn, wv = find(write, 'conv3x3_patch_kernel<2, true, 256, 2, fals', 'WRITE_SIZE')
scale_f = 512 * 1024 / cal_f                      # KB actually read / KB counted
fetch_b, write_b = int(fv * 1024 * round(scale_f)), int(wv * 1024)
sq = rows('pmc_sq.txt')
mfma_busy = find(sq, 'conv3x3_patch_kernel<2, true, 256, 2, fals', 'SQ_VALU_MFMA_BUSY_CYCLES')[1]
json.dump({
    'kernel': DOM, 'precision': 'bf16x3', 'workload': 'c2', 'batch': 16,
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
    'reading': f'writes = algorithmic; reads = {fetch_b / 313174698:.2f}x the input tensor (tile halos); '
               f'SQ_VALU_MFMA_BUSY_CYCLES {mfma_busy:.3g} = 28.3 M MFMAs x 32 cycles (profiles/r02_c_pmc_sq.txt); stall '
               f'attribution in profiles/r02_e_sq_stall_attribution.txt',
}, open(os.path.join(P, 'r02_pmc_traffic.json'), 'w'), indent=1)
print(open(os.path.join(P, 'r02_pmc_traffic.json')).read())
