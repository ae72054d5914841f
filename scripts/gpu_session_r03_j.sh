#!/bin/bash
# Round-3 session J: two hypotheses for the epilogue cost of the transposed / stride-1 tiles.
#  PITCH   - the (2W+1)-wide output rows start at odd dword offsets: time the transposed kernel writing 16-byte aligned rows
#  STAGGER - all CUs reach their epilogue at the same time (HBM write burst, idle matrix pipes): de-phase the first blocks
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03j
mkdir -p $O
cd $R
GANGEALING_CONV_PRECISION=fp16x3 python scripts/convt_probe.py > $O/probe_shipped.txt 2>&1
for v in PITCH STAGGER; do
  GANGEALING_HIP_LIB=$R/ab_lib/libgg_$v.so GANGEALING_CONV_PRECISION=fp16x3 python scripts/convt_probe.py > $O/probe_$v.txt 2>&1
done
GANGEALING_CONV_PRECISION=fp16x3 ITERS=50 python scripts/conv_bench.py "G conv" > $O/gconv_shipped.txt 2>&1
GANGEALING_HIP_LIB=$R/ab_lib/libgg_STAGGER.so GANGEALING_CONV_PRECISION=fp16x3 ITERS=50 python scripts/conv_bench.py "G conv" > $O/gconv_STAGGER.txt 2>&1
for f in $O/*.txt; do echo "== $f"; grep -v amdgpu.ids $f; done
