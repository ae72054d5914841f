#!/bin/bash
# Round-3 session W: generic split kernel (stride-2 correlations, 1x1, small images) gathering through one buffer resource
# (lane offset + scalar channel offset) instead of 64-bit pointer arithmetic; against ab_lib/libgg_prevconv.so (HEAD).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03w
mkdir -p $O
export GANGEALING_SYNTHETIC=1
cd $R
timeout 900 python -m pytest tests/test_gpu_c2_layer_ops.py tests/test_gpu_split_precision.py tests/test_gpu_ops.py -m gpu -q -x 2>&1 | tail -3 > $O/pytest.txt
GANGEALING_CONV_PRECISION=fp16x3 ITERS=40 python scripts/conv_bench.py > $O/layers_new.txt 2>&1
GANGEALING_HIP_LIB=$R/ab_lib/libgg_prevconv.so GANGEALING_CONV_PRECISION=fp16x3 ITERS=40 python scripts/conv_bench.py > $O/layers_prev.txt 2>&1
for i in 1 2; do
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_new_$i.json 2>/dev/null
  GANGEALING_HIP_LIB=$R/ab_lib/libgg_prevconv.so python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_prev_$i.json 2>/dev/null
done
cat $O/pytest.txt
paste <(grep "dgrad\|down\|skip\|conv 4 \|conv 8 \|VGG 8" $O/layers_new.txt | cut -c1-100) <(grep "dgrad\|down\|skip\|conv 4 \|conv 8 \|VGG 8" $O/layers_prev.txt | cut -c66-100)
for f in $O/bench_*.json; do echo -n "$f "; head -c 175 $f | tail -c 60; echo; done
