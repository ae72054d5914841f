#!/bin/bash
# The round's measurement session on the GPU box, in SECTIONS (one gpurun call runs the sections named on the command line; no
# arguments = the full session behind profiles/<round>_* and profiles/bench_<round>_*, copied by scripts/collect_profiles.py).
# ROUND=r06 (default) names the output directory gpurun_out/$ROUND and the files; rounds 2 - 4 used earlier forms of this script:
#
#   smoke        __graft_entry__.smoke()
#   suite        GPU test suite + parity report (-> parity_$ROUND.json)
#   determinism  scripts/check_determinism.py on three configurations x three arithmetic modes
#   bench        the driver's command, then every arithmetic mode / batch / workload (c4, c5 at batch 4 and 16)
#   layers       per-layer convolution table (scripts/conv_bench.py), blur and splat2d stand-alone benchmarks
#   trace        rocprofv3 --kernel-trace of the bench command for c2 / c4 / c5 (-> kernel_stats_*.txt)
#   pmc          FETCH_SIZE / WRITE_SIZE passes of the bench command + the known-traffic calibration kernel, matrix-pipe
#                busy counters, the same two passes over the splat2d benchmark
#   ab <so>      same-box A/B of another build of the library (scripts/build_ab_lib.sh) on the bench and the layer table
#   reference    the reference's own modules on the HIP operators (literal drop-in) + the 1-rank RCCL trainer test
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
ROUND=${ROUND:-r06}
O=$R/gpurun_out/$ROUND
mkdir -p $O
export GANGEALING_SYNTHETIC=1 TMPDIR=/tmp
cd $R
git rev-parse HEAD > $O/commit.txt 2>/dev/null || echo "no-git-on-box" > $O/commit.txt
python - > $O/kernel_source_sha16.txt <<'PY'
import bench
print(bench.kernel_source_hash())
PY
SECTIONS="$*"
[ -z "$SECTIONS" ] && SECTIONS="suite determinism bench layers trace pmc"
B="python bench.py --no-cpu-baseline --no-extras"

sec_smoke() {
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
  tail -n 2 $O/smoke.txt
}
sec_suite() {
  timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/pytest_gpu.txt
  cp gpurun_out/parity_report.json $O/parity_$ROUND.json 2>/dev/null
  cat $O/pytest_gpu.txt
}
sec_reference() {
  timeout 900 python -m pytest tests/test_gpu_reference_dropin.py tests/test_gpu_rccl_single_rank.py -q -m gpu 2>&1 | tail -6 > $O/pytest_reference.txt
  cat $O/pytest_reference.txt
}
sec_determinism() {
  ( for cfg in small cluster c2; do for prec in fp16x3 bf16x3 fp32; do timeout 300 python scripts/check_determinism.py $cfg $prec 2>&1 | tail -1; done; done ) > $O/determinism.txt 2>&1
  cat $O/determinism.txt
}
sec_bench() {
  # the driver's command (its extras carry bf16x3 / bf16 / hipGraph / batch 5 eager + replay / C4 and C5 at batch 16 /
  # both drop-in routes), then the exact-fp32 mode
  python bench.py --steps 20 --warmup 5 > $O/bench_fp16x3.json 2> $O/bench_fp16x3.err
  $B --steps 10 --warmup 3 --precision fp32 > $O/bench_fp32.json 2>/dev/null
  if [ -z "${QUICK:-}" ]; then
    $B --steps 30 --warmup 5 --precision bf16x3 > $O/bench_bf16x3.json 2>/dev/null
    $B --steps 30 --warmup 5 --precision bf16 > $O/bench_bf16.json 2>/dev/null
    $B --steps 30 --warmup 5 --batch 5 > $O/bench_fp16x3_batch5.json 2>/dev/null
    $B --steps 30 --warmup 5 --batch 5 --graph > $O/bench_fp16x3_batch5_hipgraph.json 2>/dev/null
    $B --steps 20 --warmup 5 --batch 32 > $O/bench_fp16x3_batch32.json 2>/dev/null
    for w in c4 c5; do for b in 4 16; do
      $B --workload $w --batch $b --steps 10 --warmup 3 > $O/bench_${w}_batch$b.json 2>/dev/null
    done; done
  fi
  head -c 600 $O/bench_fp16x3.json; echo
}
sec_layers() {
  python scripts/blur_bench.py > $O/blur_bench.txt 2>&1
  if [ -z "${QUICK:-}" ]; then        # (QUICK=1: the splat2d kernels did not change since the last full session)
    python scripts/splat_bench.py $O/splat_bench.json > $O/splat_bench.txt 2>&1
    ( cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $O/trace_splat -o trace --output-format rocpd -- python $R/scripts/splat_bench.py > /dev/null 2>&1 )
    python scripts/rocpd_stats.py $(find $O/trace_splat -name "*.db" | head -1) 20 > $O/splat_kernel_stats.txt 2>&1
    rm -rf $O/trace_splat
  fi
  GANGEALING_CONV_PRECISION=fp16x3 ITERS=20 python scripts/conv_bench.py > $O/conv_layers.txt 2>&1
  GANGEALING_CONV_PRECISION=bf16x3 ITERS=20 python scripts/conv_bench.py "G " > $O/conv_layers_bf16x3.txt 2>&1
}
sec_trace() {
  cd /tmp
  for w in c2 c4 c5; do
    BATCH=""; [ $w != c2 ] && BATCH="--batch 16"
    CMD="python $R/bench.py --workload $w $BATCH --steps 10 --warmup 3 --no-cpu-baseline --no-extras"
    timeout 900 rocprofv3 --kernel-trace -d $O/trace_$w -o trace --output-format rocpd -- $CMD > $O/bench_${w}_under_rocprofv3.json 2>/dev/null
    DB=$(find $O/trace_$w -name "*.db" | head -1)
    python $R/scripts/rocpd_stats.py $DB 130 > $O/kernel_stats_$w.txt 2>&1
    [ $w = c2 ] && python $R/scripts/rocpd_timeline.py $DB > $O/timeline_c2.txt 2>&1
    rm -rf $O/trace_$w
  done
  cd $R
  head -8 $O/kernel_stats_c2.txt | cut -c1-170
}
sec_pmc() {
  cd /tmp
  SHORT="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $SHORT > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $SHORT > /dev/null 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/cal_fetch -- python $R/scripts/pmc_calibrate.py > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/cal_write -- python $R/scripts/pmc_calibrate.py > /dev/null 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/pmc_sq -- $SHORT > /dev/null 2>&1
  if [ -z "${QUICK:-}" ]; then
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/splat_fetch -- python $R/scripts/splat_bench.py > /dev/null 2>&1
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/splat_write -- python $R/scripts/splat_bench.py > /dev/null 2>&1
  fi
  cd $R
  # round 6: the same two passes for C4 / C5 at the recipes' per-GPU batch (roofline.traffic of extras.c4_batch16 / c5_batch16)
  for w in c4 c5; do
    WCMD="python $R/bench.py --workload $w --batch 16 --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$w -- $WCMD > /dev/null 2>&1
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$w -- $WCMD > /dev/null 2>&1
  done
  cd $R
  for d in pmc_fetch_c4 pmc_write_c4 pmc_fetch_c5 pmc_write_c5; do
    python scripts/pmc_kernel.py $O/$d "" > $O/$d.txt 2>&1
    rm -rf $O/$d
  done
  for d in pmc_fetch pmc_write cal_fetch cal_write pmc_sq; do
    python scripts/pmc_kernel.py $O/$d "" > $O/$d.txt 2>&1
    rm -rf $O/$d
  done
  for d in splat_fetch splat_write; do
    python scripts/pmc_kernel.py $O/$d "splat" > $O/$d.txt 2>&1
    rm -rf $O/$d
  done
}
sec_ab() {      # $AB_LIB: path of the other build, relative to the repo root
  OLD=$R/${AB_LIB:?set AB_LIB=ab_lib/<name>/libgangealing_hip.so}
  for rep in 1 2; do
    GANGEALING_HIP_LIB=$OLD $B --steps 30 --warmup 5 > $O/ab_bench_old_$rep.json 2>/dev/null
    $B --steps 30 --warmup 5 > $O/ab_bench_new_$rep.json 2>/dev/null
  done
  GANGEALING_CONV_PRECISION=fp16x3 GANGEALING_HIP_LIB=$OLD python scripts/conv_bench.py > $O/ab_layers_old.txt 2>&1
  GANGEALING_CONV_PRECISION=fp16x3 python scripts/conv_bench.py > $O/ab_layers_new.txt 2>&1
  for f in $O/ab_bench_*.json; do python -c "
import json
d=json.loads([l for l in open('$f').read().strip().splitlines() if l.startswith('{')][-1]); print('$f'.split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"; done
}
for s in $SECTIONS; do echo "== $s"; sec_$s; done
ls $O | head -80
