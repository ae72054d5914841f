#!/bin/bash
# round 4 session D: overlapped block-exponent work (patch / split kernels), binned splat2d, bench kernel survey
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_split_precision.py tests/test_gpu_ops.py tests/test_gpu_c2_layer_ops.py -q -m gpu --maxfail=8 2>&1 | tail -30 > $O/r04d_pytest_kernels.txt
OLD=$PWD/ab_lib/r03conv/libgangealing_hip.so
for rep in 1 2; do
  GANGEALING_F16_GRADS=0 GANGEALING_HIP_LIB=$OLD python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $O/r04d_bench_r03conv_$rep.json 2>/dev/null
  GANGEALING_F16_GRADS=0 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $O/r04d_bench_new_bf16grads_$rep.json 2>/dev/null
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $O/r04d_bench_new_$rep.json 2>$O/r04d_bench_new_$rep.err
done
GANGEALING_CONV_PRECISION=fp16x3 python scripts/conv_bench.py > $O/r04d_layers_new.txt 2>&1
python scripts/splat_bench.py $O/r04d_splat_bench.json > $O/r04d_splat_bench.txt 2>&1
for w in c4 c5; do for b in 4 16; do
  timeout 600 python bench.py --workload $w --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/r04d_bench_${w}_b$b.json 2>$O/r04d_bench_${w}_b$b.err
done; done
for f in $O/r04d_bench_*.json; do python -c "
import json,sys
try:
    d=json.loads([l for l in open('$f').read().strip().splitlines() if l.startswith('{')][-1]); print('$f'.split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['kernel'][:60], d['roofline']['frac'])
except Exception as e: print('$f', 'ERR', e)"; done
tail -6 $O/r04d_pytest_kernels.txt; cat $O/r04d_splat_bench.txt | tail -5
