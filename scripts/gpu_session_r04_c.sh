#!/bin/bash
# round 4 session C: whole GPU suite at the new library (allocator hook, block exponents), then a same-box A/B of the
# convolution kernels: round-3 conv_mfma.hip (ab_lib/r03conv) vs today's, forward-only change (F16 grads off) and full
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $O/r04c_pytest_gpu.txt
cp $O/parity_report.json $O/r04c_parity_report.json 2>/dev/null
OLD=$PWD/ab_lib/r03conv/libgangealing_hip.so
for rep in 1 2; do
  GANGEALING_F16_GRADS=0 GANGEALING_HIP_LIB=$OLD python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $O/r04c_bench_r03conv_$rep.json 2>/dev/null
  GANGEALING_F16_GRADS=0 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $O/r04c_bench_new_bf16grads_$rep.json 2>/dev/null
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $O/r04c_bench_new_$rep.json 2>/dev/null
done
GANGEALING_CONV_PRECISION=fp16x3 GANGEALING_HIP_LIB=$OLD python scripts/conv_bench.py > $O/r04c_layers_r03conv.txt 2>&1
GANGEALING_CONV_PRECISION=fp16x3 python scripts/conv_bench.py > $O/r04c_layers_new.txt 2>&1
for f in $O/r04c_bench_*.json; do python -c "
import json,sys
d=json.loads([l for l in open('$f').read().strip().splitlines() if l.startswith('{')][-1]); print('$f'.split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"; done
tail -4 $O/r04c_pytest_gpu.txt
