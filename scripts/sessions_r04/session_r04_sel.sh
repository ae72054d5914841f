#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/sel
mkdir -p $O
export GANGEALING_SYNTHETIC=1 TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_gpu_split_precision.py -q -m gpu -k "split_conv or s2_patch or block_exponent" 2>&1 | tail -1
for lib in ab_lib/s2_head/libgangealing_hip.so gangealing_amd/lib/libgangealing_hip.so; do
  echo "== $lib"
  for rep in 1 2; do for f in dgrad down; do GANGEALING_HIP_LIB=$lib GANGEALING_CONV_PRECISION=fp16x3 ITERS=30 timeout 300 python scripts/conv_bench.py "$f" 2>&1 | grep "dgrad 65\|dgrad 129\|dgrad 257\|STN down 129\|STN down 65"; done; done
done
B="python bench.py --no-cpu-baseline --no-extras --steps 30 --warmup 5"
for rep in 1 2; do for lib in ab_lib/s2_head/libgangealing_hip.so gangealing_amd/lib/libgangealing_hip.so; do
  GANGEALING_HIP_LIB=$lib $B 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1]); print('$lib', d['value'], d['ms_per_step'])"
done; done
