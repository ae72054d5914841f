#!/bin/bash
# Round 5, session 6: how often does `check_determinism.py cluster fp16x3` differ, and with which of the round's features
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
OUT=$R/gpurun_out/r05
mkdir -p $OUT
export GANGEALING_SYNTHETIC=1
run() {   # name, env...
  name=$1; shift
  bad=0
  for i in $(seq 1 ${RUNS:-14}); do
    env "$@" timeout 120 python scripts/check_determinism.py cluster fp16x3 > /tmp/det_$name_$i.txt 2>&1
    if ! grep -q "BITWISE IDENTICAL" /tmp/det_$name_$i.txt; then
      bad=$((bad+1))
      echo "---- $name run $i" >> $OUT/s6_determinism_failures.txt
      grep "cluster fp16x3" /tmp/det_$name_$i.txt | head -n 70 | cut -c1-200 >> $OUT/s6_determinism_failures.txt
    fi
  done
  echo "$name: $bad of ${RUNS:-14} runs differ"
}
( run default A=1
  run no_sign_bits GG_DISABLE=sign_bits
  run round4_transposed_tile GG_CONVT16=0
  run round4_equivalent GG_DISABLE=sign_bits GG_CONVT16=0
  [ -z "${SKIP_TS:-}" ] && run no_two_streams GG_DISABLE=two_streams ) >> $OUT/s6_determinism_stats.txt 2>&1
cat $OUT/s6_determinism_stats.txt
head -n 80 $OUT/s6_determinism_failures.txt 2>/dev/null
