#!/bin/bash
# Round-3 session G: the fp16x3 mode (binary16 limbs on the forward convolutions): parity tests in that mode, bench A/B
# against bf16x3.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03g
mkdir -p $O
export GANGEALING_SYNTHETIC=1
cd $R
timeout 1200 python -m pytest tests -m gpu -q -k "fp16x3" > $O/pytest_fp16x3.log 2>&1
echo "rc $?" >> $O/pytest_fp16x3.log
cp gpurun_out/parity_report.json $O/parity_fp16x3.json 2>/dev/null
for rep in 1 2; do
  python bench.py --precision bf16x3 --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_bf16x3_$rep.json 2>$O/err_a_$rep.txt
  python bench.py --precision fp16x3 --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_fp16x3_$rep.json 2>$O/err_b_$rep.txt
done
GANGEALING_CONV_PRECISION=fp16x3 ITERS=20 python scripts/conv_bench.py "G " > $O/conv_layers_fp16x3.txt 2>&1
tail -12 $O/pytest_fp16x3.log
for f in $O/bench_*_?.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['config']['loss'])"; done
grep "G conv\|upconv [0-9]" $O/conv_layers_fp16x3.txt | cut -c1-120
