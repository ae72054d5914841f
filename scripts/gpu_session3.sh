#!/bin/bash
mkdir -p gpurun_out
python scripts/debug_ema.py > gpurun_out/s3_ema.log 2>&1
timeout 1500 python -m pytest tests/test_gpu_configs.py tests/test_gpu_ddp.py tests/test_gpu_models.py -m gpu -q > gpurun_out/s3_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/s3_tests.log
grep -v "Warn\|warn" gpurun_out/s3_ema.log | tail; grep "^E  \|^FAILED\|passed\|failed" gpurun_out/s3_tests.log | cut -c1-300 | tail -30
