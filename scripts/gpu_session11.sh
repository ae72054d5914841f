#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests/test_gpu_c2_layer_ops.py tests/test_gpu_ddp.py -m gpu -q > $O/s11_tests.log 2>&1
echo "pytest rc=$?" >> $O/s11_tests.log
grep "^E  \|^FAILED\|passed\|failed" $O/s11_tests.log | cut -c1-300 | tail -12
timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $O/s11_bench.json 2> $O/s11_bench.err || tail -3 $O/s11_bench.err
python -c "
import json
d=json.load(open('$O/s11_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d.get('extras'))"
