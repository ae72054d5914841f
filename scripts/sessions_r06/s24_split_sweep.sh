#!/bin/bash
# Round 6, session 24: split-K thresholds of the patch / transposed tiles on the train step (environment overrides the
# dispatcher reads once per process); interleaved repetitions, one box.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_s24
mkdir -p $O
export GANGEALING_SYNTHETIC=1 TMPDIR=/tmp
cd $R
run() { env "$@" python bench.py --steps 60 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'])" >> $O/sweep.txt; }
for rep in 1 2 3; do
  run GG_SPLIT_PATCH=512
  run GG_SPLIT_PATCH=256
  run GG_SPLIT_PATCH=384
  run GG_SPLIT_PATCH=768
  run GG_SPLIT_PATCH=1024
  run GG_SPLIT_CONVT=128
  run GG_SPLIT_CONVT=256
  run GG_SPLIT_CONVT=384
done
sort $O/sweep.txt
