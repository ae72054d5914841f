#!/bin/bash
# Round-3 session R: streaming blur with every load of a strip (rows and, in the fused StyledConv tail, noise) issued
# before the strip's first store; against the previous library (ab_lib/libgangealing_hip_prev.so = HEAD's upfirdn2d.hip).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03r
mkdir -p $O
export GANGEALING_SYNTHETIC=1
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py tests/test_gpu_determinism.py -m gpu -q -x 2>&1 | tail -3 > $O/pytest.txt
python scripts/blur_bench.py > $O/blur_new.txt 2>&1
GANGEALING_HIP_LIB=$R/ab_lib/libgangealing_hip_prev.so python scripts/blur_bench.py > $O/blur_prev.txt 2>&1
for i in 1 2; do
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_new_$i.json 2>/dev/null
  GANGEALING_HIP_LIB=$R/ab_lib/libgangealing_hip_prev.so python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_prev_$i.json 2>/dev/null
done
cat $O/pytest.txt
paste <(grep -v "amdgpu\|direct" $O/blur_new.txt) <(grep -v "amdgpu\|direct" $O/blur_prev.txt | awk '{print $(NF-3), $(NF-2), $(NF-1), $NF}')
for f in $O/bench_*.json; do echo $f; head -c 175 $f | tail -c 60; echo; done
