#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_s20
mkdir -p $O
export GANGEALING_SYNTHETIC=1 TMPDIR=/tmp
cd $R
for dis in conv_residual conv_s2_act head_relu blur_bits resblock_fold vgg_stem; do
  echo "== GG_DISABLE=$dis" >> $O/bisect.txt
  GG_DISABLE=$dis timeout 300 python scripts/poison_check.py small 2>/dev/null | grep -E "IDENTICAL|differ|PASSED|FAILED" | head -6 >> $O/bisect.txt
done
cat $O/bisect.txt
