#!/bin/bash
# Round-3 GPU session A: determinism diagnostic, full GPU test suite, bench lines (c2 + extras, c4, c5), splat stress.
set -u
OUT=gpurun_out/r03a
mkdir -p $OUT
export GANGEALING_SYNTHETIC=1
cd "$(dirname "$0")/.."
( for cfg in small cluster c2; do for prec in bf16x3 fp32; do
    timeout 300 python scripts/check_determinism.py $cfg $prec 2>&1 | grep -v Warning | tail -12
  done; done ) > $OUT/determinism.log 2>&1
timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_ddp.py > $OUT/pytest.log 2>&1
echo "pytest rc $?" >> $OUT/pytest.log
timeout 600 python -m pytest tests/test_gpu_ddp.py -m gpu -q > $OUT/pytest_ddp.log 2>&1
echo "pytest ddp rc $?" >> $OUT/pytest_ddp.log
cp gpurun_out/parity_report.json $OUT/parity_report.json 2>/dev/null
timeout 600 python bench.py --steps 20 --warmup 3 > $OUT/bench_c2.json 2> $OUT/bench_c2.err
timeout 300 python bench.py --workload c4 --steps 10 --warmup 3 --no-extras --no-cpu-baseline > $OUT/bench_c4.json 2> $OUT/bench_c4.err
timeout 300 python bench.py --workload c5 --steps 10 --warmup 3 --no-extras --no-cpu-baseline > $OUT/bench_c5.json 2> $OUT/bench_c5.err
timeout 300 python scripts/splat_bench.py $OUT/splat_bench.json > $OUT/splat_bench.log 2>&1
tail -3 $OUT/determinism.log; tail -5 $OUT/pytest.log; tail -2 $OUT/pytest_ddp.log; cat $OUT/bench_c2.json | cut -c1-600
