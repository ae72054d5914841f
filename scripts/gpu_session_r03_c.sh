#!/bin/bash
# Round-3 session C: A/B of the transposed-convolution kernel with tap-grouped B-fragment reuse (new) against the
# session-A build (ab_lib/libgg_r03a.so), per layer and on the bench; skip-path blur-at-stride A/B; targeted tests.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03c
mkdir -p $O
export GANGEALING_SYNTHETIC=1 GANGEALING_CONV_PRECISION=bf16x3 ITERS=20
cd $R
timeout 600 python -m pytest tests/test_gpu_c2_layer_ops.py tests/test_gpu_split_precision.py tests/test_gpu_models.py -m gpu -q -x > $O/pytest_subset.log 2>&1
echo "rc $?" >> $O/pytest_subset.log
GANGEALING_HIP_LIB=$R/ab_lib/libgg_r03a.so python scripts/conv_bench.py "G upconv" > $O/convbench_old.txt 2>&1
python scripts/conv_bench.py "G upconv" > $O/convbench_new.txt 2>&1
for rep in 1 2; do
  GANGEALING_HIP_LIB=$R/ab_lib/libgg_r03a.so GG_DISABLE=skip_down python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_old_$rep.json 2>$O/err_old_$rep.txt
  GG_DISABLE=skip_down python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_newkernel_$rep.json 2>$O/err_nk_$rep.txt
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_new_$rep.json 2>$O/err_new_$rep.txt
done
tail -3 $O/pytest_subset.log
paste -d'\n' $O/convbench_old.txt $O/convbench_new.txt | grep upconv | cut -c1-140
for f in $O/bench_*.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"; done
