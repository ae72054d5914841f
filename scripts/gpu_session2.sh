#!/bin/bash
# round-2 GPU session 2: full GPU suite (no -x) with the fp64-referenced gradient checks, blur A/B, bench
mkdir -p gpurun_out
cp gpurun_out/splat2d.npz tests/golden/splat2d.npz 2>/dev/null
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/s2_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/s2_tests.log
echo "== stream kernel" > gpurun_out/s2_blur.log; python scripts/blur_bench.py >> gpurun_out/s2_blur.log 2>&1
echo "== tile kernel (round 1)" >> gpurun_out/s2_blur.log; GG_BLUR_TILE=1 python scripts/blur_bench.py >> gpurun_out/s2_blur.log 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/s2_bench.json 2> gpurun_out/s2_bench.err
GG_BLUR_TILE=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/s2_bench_tile.json 2>> gpurun_out/s2_bench.err
grep -v "^$" gpurun_out/s2_tests.log | tail -25; cat gpurun_out/s2_blur.log; cut -c1-200 gpurun_out/s2_bench.json gpurun_out/s2_bench_tile.json
