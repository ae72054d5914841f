#!/bin/bash
# Round 5, session 7: with the two-stream fork opt-in (the new default: no fork) - repeated determinism runs of the
# configuration that differed in 5 % of the runs, the cost at per-GPU batch 5, the 9-configuration determinism record.
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
OUT=$R/gpurun_out/r05
mkdir -p $OUT
export GANGEALING_SYNTHETIC=1
bad=0
for i in $(seq 1 60); do
  timeout 120 python scripts/check_determinism.py cluster fp16x3 > /tmp/det_$i.txt 2>&1
  grep -q "BITWISE IDENTICAL" /tmp/det_$i.txt || { bad=$((bad+1)); grep "cluster fp16x3" /tmp/det_$i.txt | head -n 5; }
done
echo "default (no fork): $bad of 60 runs differ" | tee $OUT/s7_two_stream_check.txt
for rep in 1 2; do
  for en in none two_streams; do
    if [ $en = none ]; then unset GG_ENABLE; else export GG_ENABLE=$en; fi
    timeout 200 python bench.py --batch 5 --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch 5 eager, GG_ENABLE=$en:', d['value'], 'img/s', d['ms_per_step'], 'ms/step')" | tee -a $OUT/s7_two_stream_check.txt
  done
done
unset GG_ENABLE
( for cfg in small cluster c2; do for prec in fp16x3 bf16x3 fp32; do timeout 300 python scripts/check_determinism.py $cfg $prec 2>&1 | tail -1; done; done ) > $OUT/determinism.txt 2>&1
cat $OUT/determinism.txt
timeout 600 python -m pytest tests/test_gpu_determinism.py tests/test_gpu_ddp.py tests/test_gpu_bench_smoke.py -q 2>&1 | tail -n 3 | tee -a $OUT/s7_two_stream_check.txt
