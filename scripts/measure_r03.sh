#!/bin/bash
# Round-3 measurement session on the GPU box (one gpurun call).  Everything under profiles/r03_* and
# profiles/bench_r03_* comes from one run of this script (scripts/collect_profiles_r03.py copies the summaries):
#   GPU test suite + parity report; the bench in every arithmetic mode / batch / workload; determinism check;
#   rocprofv3 kernel trace of the default bench command (c2) and of c4 / c5; FETCH_SIZE / WRITE_SIZE PMC passes with the
#   known-traffic calibration kernel; matrix-pipe counters of the dominant kernel; per-layer convolution table; blur
#   and splat2d stand-alone benchmarks (splat2d also under the PMC passes).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03
mkdir -p $O
export GANGEALING_SYNTHETIC=1
cd $R
git rev-parse HEAD > $O/commit.txt 2>/dev/null || echo "no-git-on-box" > $O/commit.txt
python - > $O/kernel_source_sha16.txt <<'PY'
import bench
print(bench.kernel_source_hash())
PY
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $O/pytest_gpu.txt
cp gpurun_out/parity_report.json $O/parity_r03.json 2>/dev/null
( for cfg in small cluster c2; do for prec in fp16x3 bf16x3 fp32; do timeout 300 python scripts/check_determinism.py $cfg $prec 2>&1 | tail -1; done; done ) > $O/determinism.txt 2>&1
python bench.py --steps 30 --warmup 5 > $O/bench_fp16x3.json 2> $O/bench_fp16x3.err
python bench.py --steps 30 --warmup 5 --precision bf16x3 --no-cpu-baseline --no-extras > $O/bench_bf16x3.json 2>/dev/null
python bench.py --steps 10 --warmup 3 --precision fp32 --no-cpu-baseline --no-extras > $O/bench_fp32.json 2>/dev/null
python bench.py --steps 30 --warmup 5 --precision bf16 --no-cpu-baseline --no-extras > $O/bench_bf16.json 2>/dev/null
python bench.py --steps 30 --warmup 5 --batch 5 --no-cpu-baseline --no-extras > $O/bench_fp16x3_batch5.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --batch 32 --no-cpu-baseline --no-extras > $O/bench_fp16x3_batch32.json 2>/dev/null
python bench.py --workload c4 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_c4.json 2>/dev/null
python bench.py --workload c5 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_c5.json 2>/dev/null
python scripts/blur_bench.py > $O/blur_bench.txt 2>&1
python scripts/splat_bench.py $O/splat_bench.json > $O/splat_bench.txt 2>&1
GANGEALING_CONV_PRECISION=fp16x3 ITERS=20 python scripts/conv_bench.py > $O/conv_layers.txt 2>&1
GANGEALING_CONV_PRECISION=bf16x3 ITERS=20 python scripts/conv_bench.py "G " > $O/conv_layers_bf16x3.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for w in c2 c4 c5; do
  CMD="python $R/bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-extras"
  timeout 600 rocprofv3 --kernel-trace -d $O/trace_$w -o trace --output-format rocpd -- $CMD > $O/bench_${w}_under_rocprofv3.json 2>/dev/null
  DB=$(find $O/trace_$w -name "*.db" | head -1)
  python $R/scripts/rocpd_stats.py $DB 120 > $O/kernel_stats_$w.txt 2>&1
  rm -rf $O/trace_$w
done
SHORT="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $SHORT > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $SHORT > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/cal_fetch -- python $R/scripts/pmc_calibrate.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/cal_write -- python $R/scripts/pmc_calibrate.py > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/pmc_sq -- $SHORT > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/splat_fetch -- python $R/scripts/splat_bench.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/splat_write -- python $R/scripts/splat_bench.py > /dev/null 2>&1
cd $R
for d in pmc_fetch pmc_write cal_fetch cal_write pmc_sq; do
  python scripts/pmc_kernel.py $O/$d "" > $O/$d.txt 2>&1
  rm -rf $O/$d
done
for d in splat_fetch splat_write; do
  python scripts/pmc_kernel.py $O/$d "splat" > $O/$d.txt 2>&1
  rm -rf $O/$d
done
ls -la $O; cat $O/pytest_gpu.txt; cat $O/determinism.txt; head -c 700 $O/bench_fp16x3.json; echo; head -8 $O/kernel_stats_c2.txt | cut -c1-170
