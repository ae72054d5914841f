#!/bin/bash
# A/B of the single-limb patch kernel's taps-per-barrier-interval staging (ab_lib/ holds both builds), then the GPU suite.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/tpi
mkdir -p $O
cd $R
for rep in 1 2; do
for v in tpi1 tpi3; do
  GANGEALING_HIP_LIB=$R/ab_lib/libgg_$v.so python bench.py --steps 30 --warmup 5 --precision bf16 --no-cpu-baseline --no-extras > $O/bench_bf16_${v}_$rep.json 2>$O/err_${v}_$rep.txt
done
done
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > $O/pytest_gpu.txt
cp gpurun_out/parity_report.json $O/parity.json 2>/dev/null
for f in $O/bench_bf16_*.json; do echo $f; python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline'])"; done
cat $O/pytest_gpu.txt
