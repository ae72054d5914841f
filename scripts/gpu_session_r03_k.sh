#!/bin/bash
# Round-3 session K: epilogues that load their per-channel / per-pixel parameters before the first store
# (LDS-resident) against the previous convolution library (ab_lib/libgg_prevconv.so = HEAD's conv_mfma.hip).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03k
mkdir -p $O
export GANGEALING_SYNTHETIC=1
cd $R
timeout 900 python -m pytest tests/test_gpu_c2_layer_ops.py tests/test_gpu_split_precision.py tests/test_gpu_ops.py tests/test_gpu_models.py tests/test_gpu_determinism.py -m gpu -q -x 2>&1 | tail -5 > $O/pytest.txt
GANGEALING_CONV_PRECISION=fp16x3 python scripts/convt_probe.py > $O/probe_new.txt 2>&1
GANGEALING_HIP_LIB=$R/ab_lib/libgg_prevconv.so GANGEALING_CONV_PRECISION=fp16x3 python scripts/convt_probe.py > $O/probe_prev.txt 2>&1
GANGEALING_CONV_PRECISION=fp16x3 ITERS=30 python scripts/conv_bench.py > $O/layers_new.txt 2>&1
GANGEALING_HIP_LIB=$R/ab_lib/libgg_prevconv.so GANGEALING_CONV_PRECISION=fp16x3 ITERS=30 python scripts/conv_bench.py > $O/layers_prev.txt 2>&1
for i in 1 2; do
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_new_$i.json 2>/dev/null
  GANGEALING_HIP_LIB=$R/ab_lib/libgg_prevconv.so python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_prev_$i.json 2>/dev/null
done
cat $O/pytest.txt
paste <(grep -v amdgpu $O/probe_new.txt) <(grep -v amdgpu $O/probe_prev.txt | awk '{print $(NF-3), $(NF-2), $(NF-1), $NF}')
paste <(grep -v "amdgpu\|^batch" $O/layers_new.txt | cut -c1-100) <(grep -v "amdgpu\|^batch" $O/layers_prev.txt | cut -c66-100)
for f in $O/bench_*.json; do echo $f; head -c 175 $f | tail -c 60; echo; done
