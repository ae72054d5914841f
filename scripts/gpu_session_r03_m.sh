#!/bin/bash
# Round-3 session M: full GPU suite on the new library; generic weight-gradient kernels without the per-slab 64-bit
# division (STN lines of conv_bench: wgrad column) and the whole step against the previous builds.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03m
mkdir -p $O
export GANGEALING_SYNTHETIC=1
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 > $O/pytest.txt
GANGEALING_CONV_PRECISION=fp16x3 ITERS=30 python scripts/conv_bench.py STN > $O/stn_new.txt 2>&1
GANGEALING_HIP_LIB=$R/ab_lib/libgg_prevconv.so GANGEALING_CONV_PRECISION=fp16x3 ITERS=30 python scripts/conv_bench.py STN > $O/stn_prev.txt 2>&1
for i in 1 2; do
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_new_$i.json 2>/dev/null
  GANGEALING_HIP_LIB=$R/ab_lib/libgg_prevconv.so python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_prevconv_$i.json 2>/dev/null
  GANGEALING_HIP_LIB=$R/ab_lib/libgangealing_hip_prev.so python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_prevall_$i.json 2>/dev/null
done
python bench.py --steps 30 --warmup 5 --batch 5 --no-cpu-baseline --no-extras > $O/bench_new_b5.json 2>/dev/null
GANGEALING_HIP_LIB=$R/ab_lib/libgangealing_hip_prev.so python bench.py --steps 30 --warmup 5 --batch 5 --no-cpu-baseline --no-extras > $O/bench_prevall_b5.json 2>/dev/null
cat $O/pytest.txt
paste <(grep "STN" $O/stn_new.txt | cut -c1-140) <(grep "STN" $O/stn_prev.txt | cut -c100-140)
for f in $O/bench_*.json; do echo $f; head -c 175 $f | tail -c 60; echo; done
