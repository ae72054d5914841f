#!/bin/bash
mkdir -p gpurun_out
python scripts/debug_uninit.py > gpurun_out/s4_uninit.log 2>&1
grep -v "Warn\|warn\|amdgpu.ids" gpurun_out/s4_uninit.log | tail -40
timeout 1500 python -m pytest tests/test_gpu_configs.py tests/test_gpu_ddp.py tests/test_gpu_models.py -m gpu -q > gpurun_out/s4_tests.log 2>&1
grep "^E  \|^FAILED\|passed\|failed" gpurun_out/s4_tests.log | cut -c1-300 | tail -30
