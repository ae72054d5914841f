#!/bin/bash
# One measurement session on the GPU box: tests, bench in the three arithmetic modes, rocprofv3 kernel trace of the
# bench command, and FETCH_SIZE / WRITE_SIZE PMC passes (with a known-traffic calibration kernel).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q 2>&1 | tail -3 > $O/pytest_gpu.txt
python bench.py --steps 20 --warmup 5 > $O/bench_bf16x3.json 2> $O/bench_bf16x3.err
python bench.py --steps 10 --warmup 3 --precision fp32 --no-cpu-baseline > $O/bench_fp32.json 2>/dev/null
python bench.py --steps 10 --warmup 3 --precision bf16x6 --no-cpu-baseline > $O/bench_bf16x6.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_under_rocprofv3.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/cal_fetch -- python $R/scripts/pmc_calibrate.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/cal_write -- python $R/scripts/pmc_calibrate.py > /dev/null 2>&1
cd $R
for d in pmc_fetch pmc_write cal_fetch cal_write; do
  python scripts/pmc_kernel.py $O/$d "" > $O/$d.txt 2>&1
  find $O/$d -name "*.csv" -delete
done
ls -la $O
