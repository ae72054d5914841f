#!/bin/bash
# Round-3 session O: ablation of the dominant stride-1 tile (power_probe.py on measurement builds: no activation loads /
# no weight loads / epilogue without its stores / no epilogue), random and zero operands.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03o
mkdir -p $O
cd $R
ITERS=100 python scripts/power_probe.py > $O/probe_shipped.txt 2>&1
for v in P_NO_PLOAD P_NO_WLOAD P_EPI_NOSTORE NO_EPI; do
  GANGEALING_HIP_LIB=$R/ab_lib/libgg_$v.so ITERS=100 python scripts/power_probe.py > $O/probe_$v.txt 2>&1
done
for f in $O/probe_*.txt; do echo "== $f"; grep -v amdgpu.ids $f | grep -v "^bf16 " | grep "random\|all zero"; done
