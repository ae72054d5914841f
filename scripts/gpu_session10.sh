#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_gpu_models.py tests/test_gpu_configs.py -m gpu -q -x > $O/s10_tests.log 2>&1
grep "^E  \|^FAILED\|passed\|failed" $O/s10_tests.log | cut -c1-300 | tail -8
for env in "" "GG_DISABLE=two_streams"; do
  for args in "" "--batch 5"; do
    env $env timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline $args > $O/s10_bench.json 2> $O/s10_bench.err || tail -3 $O/s10_bench.err
    python -c "
import json
d=json.load(open('$O/s10_bench.json')); print('[$env $args]', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d.get('extras'))"
  done
done
