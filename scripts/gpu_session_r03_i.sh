#!/bin/bash
# Round-3 session I: where does the stride-2 family lose its time?  The probe on the shipped library and on three
# measurement builds of the transposed kernel (no weight loads / no activation loads / no epilogue).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03i
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_split_precision.py -m gpu -q -x 2>&1 | tail -3 > $O/pytest.txt
for p in fp16x3 bf16x3; do
  GANGEALING_CONV_PRECISION=$p python scripts/convt_probe.py > $O/probe_shipped_$p.txt 2>&1
done
for v in NO_WLOAD NO_PLOAD NO_EPI; do
  GANGEALING_HIP_LIB=$R/ab_lib/libgg_$v.so GANGEALING_CONV_PRECISION=fp16x3 python scripts/convt_probe.py > $O/probe_$v.txt 2>&1
done
cat $O/pytest.txt
for f in $O/probe_*.txt; do echo "== $f"; grep -v amdgpu.ids $f; done
