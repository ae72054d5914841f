#!/bin/bash
# Round-3 session V: sliced patch prefetch in the 4-wave / single-buffer stride-1 tiles (STN, VGG, small G layers)
# against the single burst (ab_lib/libgg_burst.so = same source with -DGG_EXP_PATCH_BURST)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03v
mkdir -p $O
export GANGEALING_SYNTHETIC=1
cd $R
timeout 600 python -m pytest tests/test_gpu_c2_layer_ops.py tests/test_gpu_split_precision.py -m gpu -q -x 2>&1 | tail -3 > $O/pytest.txt
GANGEALING_CONV_PRECISION=fp16x3 ITERS=40 python scripts/conv_bench.py "conv" > $O/layers_new.txt 2>&1
GANGEALING_HIP_LIB=$R/ab_lib/libgg_burst.so GANGEALING_CONV_PRECISION=fp16x3 ITERS=40 python scripts/conv_bench.py "conv" > $O/layers_burst.txt 2>&1
GANGEALING_CONV_PRECISION=fp16x3 ITERS=40 python scripts/conv_bench.py "VGG" > $O/vgg_new.txt 2>&1
GANGEALING_HIP_LIB=$R/ab_lib/libgg_burst.so GANGEALING_CONV_PRECISION=fp16x3 ITERS=40 python scripts/conv_bench.py "VGG" > $O/vgg_burst.txt 2>&1
for i in 1 2; do
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_new_$i.json 2>/dev/null
  GANGEALING_HIP_LIB=$R/ab_lib/libgg_burst.so python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_burst_$i.json 2>/dev/null
done
cat $O/pytest.txt
paste <(grep "conv" $O/layers_new.txt | grep -v upconv | cut -c1-100) <(grep "conv" $O/layers_burst.txt | grep -v upconv | cut -c66-100)
paste <(grep "VGG" $O/vgg_new.txt | cut -c1-100) <(grep "VGG" $O/vgg_burst.txt | cut -c66-100)
for f in $O/bench_*.json; do echo -n "$f "; head -c 175 $f | tail -c 60; echo; done
