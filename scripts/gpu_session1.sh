#!/bin/bash
# round-2 GPU session 1: reference-kernel splat fixture, full GPU test suite (with measured parity), a bench line
mkdir -p gpurun_out
python oracle/make_golden_splat.py gpurun_out/splat2d.npz > gpurun_out/s1_splat.log 2>&1 && cp gpurun_out/splat2d.npz tests/golden/splat2d.npz
timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_ddp.py > gpurun_out/s1_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/s1_tests.log
timeout 600 python -m pytest tests/test_gpu_ddp.py -m gpu -q > gpurun_out/s1_ddp.log 2>&1
echo "pytest rc=$?" >> gpurun_out/s1_ddp.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/s1_bench.json 2> gpurun_out/s1_bench.err
tail -5 gpurun_out/s1_tests.log; tail -3 gpurun_out/s1_ddp.log; cat gpurun_out/s1_bench.json | cut -c1-400
