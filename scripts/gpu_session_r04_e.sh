#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_split_precision.py tests/test_gpu_ops.py -q -m gpu --maxfail=8 2>&1 | tail -30 > $O/r04e_pytest_kernels.txt
OLD=$PWD/ab_lib/r03conv/libgangealing_hip.so
for rep in 1 2; do
  GANGEALING_F16_GRADS=0 GANGEALING_HIP_LIB=$OLD python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $O/r04e_bench_r03conv_$rep.json 2>$O/r04e_bench_r03conv_$rep.err
  GANGEALING_F16_GRADS=0 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $O/r04e_bench_new_bf16grads_$rep.json 2>/dev/null
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $O/r04e_bench_new_$rep.json 2>$O/r04e_bench_new_$rep.err
  GANGEALING_F16_GRADS=all python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $O/r04e_bench_new_allf16_$rep.json 2>/dev/null
done
GANGEALING_CONV_PRECISION=fp16x3 python scripts/conv_bench.py > $O/r04e_layers_new.txt 2>&1
GANGEALING_CONV_PRECISION=fp16x3 GANGEALING_HIP_LIB=$OLD python scripts/conv_bench.py > $O/r04e_layers_r03conv.txt 2>&1
python scripts/splat_bench.py $O/r04e_splat_bench.json > $O/r04e_splat_bench.txt 2>&1
python bench.py --workload c4 --batch 4 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/r04e_bench_c4_b4.json 2>$O/r04e_bench_c4_b4.err
for f in $O/r04e_bench_*.json; do python -c "
import json,sys
try:
    d=json.loads([l for l in open('$f').read().strip().splitlines() if l.startswith('{')][-1]); print('$f'.split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['kernel'][:50], d['roofline']['frac'])
except Exception as e: print('$f', 'ERR', e)"; done
tail -6 $O/r04e_pytest_kernels.txt; cut -c1-220 $O/r04e_splat_bench.txt | tail -5
