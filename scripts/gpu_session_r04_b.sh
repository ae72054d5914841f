#!/bin/bash
# round 4 session B: block-exponent binary16 limbs (forward + data gradients): parity suites, then a bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_split_precision.py tests/test_gpu_c2_layer_ops.py tests/test_gpu_dropin.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r04b_pytest_kernels.txt
timeout 1500 python -m pytest tests/test_gpu_configs.py tests/test_gpu_determinism.py tests/test_gpu_lpips_masks.py -q -m gpu 2>&1 | tail -40 > gpurun_out/r04b_pytest_configs.txt
cp gpurun_out/parity_report.json gpurun_out/r04b_parity_report.json 2>/dev/null
timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r04b_bench.json 2> gpurun_out/r04b_bench.err
GANGEALING_F16_GRADS=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r04b_bench_bf16grads.json 2>> gpurun_out/r04b_bench.err
tail -5 gpurun_out/r04b_pytest_kernels.txt; tail -8 gpurun_out/r04b_pytest_configs.txt
