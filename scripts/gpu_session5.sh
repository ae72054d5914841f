#!/bin/bash
mkdir -p gpurun_out
timeout 2000 python -m pytest tests -m gpu -q > gpurun_out/s5_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/s5_tests.log
grep "^E  \|^FAILED\|passed\|failed" gpurun_out/s5_tests.log | cut -c1-400 | tail -40
