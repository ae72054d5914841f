#!/bin/bash
# Round-3 session L: what in the transposed kernel's epilogue costs time - the LDS transposition or the global stores?
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03l
mkdir -p $O
cd $R
GANGEALING_CONV_PRECISION=fp16x3 python scripts/convt_probe.py > $O/probe_shipped.txt 2>&1
for v in prevconv NO_EPI EPI_NOSTORE EPI_NOLDS; do
  GANGEALING_HIP_LIB=$R/ab_lib/libgg_$v.so GANGEALING_CONV_PRECISION=fp16x3 python scripts/convt_probe.py > $O/probe_$v.txt 2>&1
done
for f in $O/probe_*.txt; do echo "== $f"; grep -v amdgpu.ids $f | grep upconv; done
GANGEALING_CONV_PRECISION=fp16x3 ITERS=30 python scripts/conv_bench.py > $O/layers_new.txt 2>&1
GANGEALING_HIP_LIB=$R/ab_lib/libgg_prevconv.so GANGEALING_CONV_PRECISION=fp16x3 ITERS=30 python scripts/conv_bench.py > $O/layers_prev.txt 2>&1
paste <(grep -v "amdgpu\|^batch" $O/layers_new.txt | cut -c1-100) <(grep -v "amdgpu\|^batch" $O/layers_prev.txt | cut -c66-100)
