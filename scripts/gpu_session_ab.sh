#!/bin/bash
# within-box A/B of two library builds on the bench, then the GPU suite.  usage: gpu_session_ab.sh <old.so> [precision]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/ab
mkdir -p $O
cd $R
OLD=$R/$1
P=${2:-bf16x3}
for rep in 1 2; do
  GANGEALING_HIP_LIB=$OLD python bench.py --steps 30 --warmup 5 --precision $P --no-cpu-baseline --no-extras > $O/bench_old_$rep.json 2>$O/err_old_$rep.txt
  python bench.py --steps 30 --warmup 5 --precision $P --no-cpu-baseline --no-extras > $O/bench_new_$rep.json 2>$O/err_new_$rep.txt
done
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > $O/pytest_gpu.txt
for f in $O/bench_*.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"; done
cat $O/pytest_gpu.txt
