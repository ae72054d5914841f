#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_split_precision.py tests/test_gpu_c2_layer_ops.py tests/test_gpu_dropin.py tests/test_gpu_ops.py -q -m gpu --maxfail=8 2>&1 | tail -8 > $O/r04h_pytest.txt
B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras"
for rep in 1 2; do
  GANGEALING_F16_GRADS=0 GANGEALING_HIP_LIB=$PWD/ab_lib/r03conv/libgangealing_hip.so $B > $O/r04h_bench_r03conv_$rep.json 2>/dev/null
  GANGEALING_HIP_LIB=$PWD/ab_lib/t8/libgangealing_hip.so $B > $O/r04h_bench_prev_$rep.json 2>/dev/null
  $B > $O/r04h_bench_new_$rep.json 2>/dev/null
done
GANGEALING_CONV_PRECISION=fp16x3 python scripts/conv_bench.py upconv > $O/r04h_layers_new.txt 2>&1
GANGEALING_CONV_PRECISION=fp16x3 GANGEALING_HIP_LIB=$PWD/ab_lib/t8/libgangealing_hip.so python scripts/conv_bench.py upconv > $O/r04h_layers_prev.txt 2>&1
GANGEALING_CONV_PRECISION=bf16x3 python scripts/conv_bench.py "upconv dgrad" > $O/r04h_layers_new_bf16x3.txt 2>&1
GANGEALING_CONV_PRECISION=bf16x3 GANGEALING_HIP_LIB=$PWD/ab_lib/r03conv/libgangealing_hip.so python scripts/conv_bench.py "upconv dgrad" > $O/r04h_layers_r03_bf16x3.txt 2>&1
for f in $O/r04h_bench_*.json; do python -c "
import json,sys
try:
    d=json.loads([l for l in open('$f').read().strip().splitlines() if l.startswith('{')][-1]); print('$f'.split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])
except Exception as e: print('$f', 'ERR', e)"; done
tail -3 $O/r04h_pytest.txt; paste -d'\n' $O/r04h_layers_r03_bf16x3.txt $O/r04h_layers_new_bf16x3.txt | grep dgrad | cut -c1-140
