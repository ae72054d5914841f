#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 2000 python -m pytest tests -m gpu -q -x > $O/s7_tests.log 2>&1
echo "pytest rc=$?" >> $O/s7_tests.log
grep "^E  \|^FAILED\|passed\|failed" $O/s7_tests.log | cut -c1-300 | tail -20
for args in "" "--graph" "--batch 5" "--batch 5 --graph" "--batch 8 --graph" "--precision bf16 --graph"; do
  tag=$(echo "$args" | tr -d ' -')
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $args > $O/s7_bench_$tag.json 2> $O/s7_bench_$tag.err || tail -5 $O/s7_bench_$tag.err
  python -c "
import json
d=json.load(open('$O/s7_bench_$tag.json')); print('[$args]', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['avg_launch_ms'])" 2>/dev/null || echo "[$args] failed"
done
