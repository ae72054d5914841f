#!/bin/bash
# Round-2 measurement session on the GPU box (one gpurun call): GPU test suite with the parity report, the bench in
# every arithmetic mode and at the reference recipe's batch, a rocprofv3 kernel trace of the default bench command,
# FETCH_SIZE / WRITE_SIZE PMC passes (plus the known-traffic calibration kernel) and SQ counters of the dominant kernels.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $O/pytest_gpu.txt
cp gpurun_out/parity_report.json $O/parity_r02.json 2>/dev/null
python bench.py --steps 30 --warmup 5 > $O/bench_bf16x3.json 2> $O/bench_bf16x3.err
python bench.py --steps 10 --warmup 3 --precision fp32 --no-cpu-baseline --no-extras > $O/bench_fp32.json 2>/dev/null
python bench.py --steps 10 --warmup 3 --precision bf16x6 --no-cpu-baseline --no-extras > $O/bench_bf16x6.json 2>/dev/null
python bench.py --steps 30 --warmup 5 --precision bf16 --no-cpu-baseline --no-extras > $O/bench_bf16.json 2>/dev/null
python bench.py --steps 30 --warmup 5 --batch 5 --no-cpu-baseline > $O/bench_bf16x3_batch5.json 2>/dev/null
python bench.py --steps 30 --warmup 5 --batch 32 --no-cpu-baseline --no-extras > $O/bench_bf16x3_batch32.json 2>/dev/null
python scripts/blur_bench.py > $O/blur_bench.txt 2>&1
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace -d $O/trace -o trace --output-format rocpd -- $CMD > $O/bench_under_rocprofv3.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/cal_fetch -- python $R/scripts/pmc_calibrate.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/cal_write -- python $R/scripts/pmc_calibrate.py > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/pmc_sq -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
# stall attribution of the dominant kernel on its own (per-layer benchmark, 512 -> 512 @64^2): SQ wave-cycle breakdown
( export GANGEALING_CONV_PRECISION=bf16x3 ITERS=5
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" \
             "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    rocprofv3 --pmc $grp --output-format csv -d $O/stall$i -- python $R/scripts/conv_bench.py "G conv 64" > /dev/null 2>&1
    python $R/scripts/pmc_kernel.py $O/stall$i "patch_kernel" >> $O/sq_stall.txt 2>&1
    rm -rf $O/stall$i
  done )
cd $R
DB=$(find $O/trace -name "*.db" | head -1)
python scripts/rocpd_stats.py $DB 90 > $O/kernel_stats.txt 2>&1
rm -rf $O/trace
for d in pmc_fetch pmc_write cal_fetch cal_write pmc_sq; do
  python scripts/pmc_kernel.py $O/$d "" > $O/$d.txt 2>&1
  rm -rf $O/$d
done
ls -la $O; cat $O/pytest_gpu.txt; head -c 600 $O/bench_bf16x3.json; echo; head -8 $O/kernel_stats.txt | cut -c1-170
