#!/bin/bash
# Round 5, session 5: which of the round's changes makes `check_determinism.py cluster fp16x3` differ between two runs
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
OUT=$R/gpurun_out/r05
mkdir -p $OUT
export GANGEALING_SYNTHETIC=1
( echo "== default"; timeout 300 python scripts/check_determinism.py cluster fp16x3 2>&1 | tail -n 12
  echo "== GG_DISABLE=sign_bits"; GG_DISABLE=sign_bits timeout 300 python scripts/check_determinism.py cluster fp16x3 2>&1 | tail -n 6
  echo "== GG_CONVT16=0"; GG_CONVT16=0 timeout 300 python scripts/check_determinism.py cluster fp16x3 2>&1 | tail -n 6
  echo "== GG_DISABLE=two_streams"; GG_DISABLE=two_streams timeout 300 python scripts/check_determinism.py cluster fp16x3 2>&1 | tail -n 6
  echo "== GG_DISABLE=sign_bits GG_CONVT16=0"; GG_DISABLE=sign_bits GG_CONVT16=0 timeout 300 python scripts/check_determinism.py cluster fp16x3 2>&1 | tail -n 6
) > $OUT/s5_determinism_debug.txt 2>&1
cut -c1-180 $OUT/s5_determinism_debug.txt
