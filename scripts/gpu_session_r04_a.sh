#!/bin/bash
# round 4 session A: the reference's own Python on the HIP operators, 1-rank RCCL, and a bench line on this box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
ls oracle/_ref oracle/_ref/pyref | head -20 > gpurun_out/r04a_ref_listing.txt
timeout 900 python -m pytest tests/test_gpu_reference_dropin.py tests/test_gpu_rccl_single_rank.py -x -q -m gpu 2>&1 | tail -40 > gpurun_out/r04a_pytest.txt
cp gpurun_out/parity_report.json gpurun_out/r04a_parity_report.json 2>/dev/null
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r04a_bench.json 2> gpurun_out/r04a_bench.err
tail -c 600 gpurun_out/r04a_pytest.txt
