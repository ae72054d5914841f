#!/bin/bash
# Round-3 session U: per-GPU batch 5 (the reference recipe; two-stream overlap of the generator passes) - is the 7 % spread
# between sessions run-to-run noise or a regression of the late-round kernel changes?  shipped | session-M library
# (before patch slicing / epilogue read grouping / blur load hoisting) | shipped convolutions with the previous blur
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03u
mkdir -p $O
export GANGEALING_SYNTHETIC=1
cd $R
for i in 1 2 3 4; do
  python bench.py --steps 40 --warmup 8 --batch 5 --no-cpu-baseline --no-extras > $O/b5_new_$i.json 2>/dev/null
  GANGEALING_HIP_LIB=$R/ab_lib/libgg_convM.so python bench.py --steps 40 --warmup 8 --batch 5 --no-cpu-baseline --no-extras > $O/b5_convM_$i.json 2>/dev/null
  GANGEALING_HIP_LIB=$R/ab_lib/libgg_sliced.so python bench.py --steps 40 --warmup 8 --batch 5 --no-cpu-baseline --no-extras > $O/b5_sliced_$i.json 2>/dev/null
done
for f in $O/b5_*.json; do echo -n "$f "; head -c 175 $f | tail -c 62; echo; done
