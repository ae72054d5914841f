#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03e
mkdir -p $O
export GANGEALING_SYNTHETIC=1
cd $R
timeout 300 python scripts/op_trace.py $O/op_trace.txt > $O/op_trace.log 2>&1
timeout 600 python -m pytest tests/test_gpu_c2_layer_ops.py tests/test_gpu_applications.py tests/test_gpu_models.py -m gpu -q -x > $O/pytest_subset.log 2>&1
for rep in 1 2; do
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_new_$rep.json 2>$O/err_new_$rep.txt
done
tail -3 $O/pytest_subset.log
for f in $O/bench_*_?.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"; done
head -60 $O/op_trace.txt | cut -c1-200
