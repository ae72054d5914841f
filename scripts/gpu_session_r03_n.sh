#!/bin/bash
# Round-3 session N: tile shape of the stride-1 patch kernel (4 x 64 vs 8 x 32 vs 16 x 16 pixels: halo 1.55x / 1.33x / 1.27x)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03n
mkdir -p $O
export GANGEALING_SYNTHETIC=1
cd $R
for tw in 64 32 16; do
  GG_PATCH_TW=$tw GANGEALING_CONV_PRECISION=fp16x3 ITERS=30 python scripts/conv_bench.py "conv" 2>&1 | grep -v amdgpu > $O/layers_tw$tw.txt
  GG_PATCH_TW=$tw python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_tw$tw.json 2>/dev/null
done
paste <(cut -c1-100 $O/layers_tw64.txt) <(cut -c66-100 $O/layers_tw32.txt) <(cut -c66-100 $O/layers_tw16.txt)
for f in $O/bench_*.json; do echo $f; head -c 175 $f | tail -c 60; echo; done
