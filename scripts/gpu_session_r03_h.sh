#!/bin/bash
# Round-3 session H: (1) tests touched since the last full run; (2) power / schedule probe of the dominant kernel;
# (3) the re-indexed generic upfirdn2d kernel against the previous library (ab_lib/, built from HEAD~ sources);
# (4) 128-pixel tiles for the 128 -> 128 @256^2 layer; (5) whole-step A/B.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03h
mkdir -p $O
export GANGEALING_SYNTHETIC=1
cd $R
timeout 600 python -m pytest tests/test_gpu_split_precision.py tests/test_gpu_ops.py -m gpu -q -x 2>&1 | tail -5 > $O/pytest.txt
python scripts/power_probe.py $O/power_probe.json > $O/power_probe.txt 2>&1
python scripts/blur_bench.py > $O/blur_new.txt 2>&1
GANGEALING_HIP_LIB=$R/ab_lib/libgangealing_hip_prev.so python scripts/blur_bench.py > $O/blur_prev.txt 2>&1
GANGEALING_CONV_PRECISION=fp16x3 ITERS=30 python scripts/conv_bench.py "G conv 256" > $O/tile256.txt 2>&1
GG_PATCH256_MIN_CIN=128 GANGEALING_CONV_PRECISION=fp16x3 ITERS=30 python scripts/conv_bench.py "G conv 256" > $O/tile128.txt 2>&1
for i in 1 2; do
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_new_$i.json 2>/dev/null
  GANGEALING_HIP_LIB=$R/ab_lib/libgangealing_hip_prev.so python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_prev_$i.json 2>/dev/null
done
GG_PATCH256_MIN_CIN=128 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_tile128.json 2>/dev/null
cat $O/pytest.txt; cat $O/power_probe.txt; grep direct $O/blur_new.txt; grep direct $O/blur_prev.txt; cat $O/tile256.txt $O/tile128.txt | grep "G conv"
for f in $O/bench_*.json; do echo $f; head -c 160 $f; echo; done
