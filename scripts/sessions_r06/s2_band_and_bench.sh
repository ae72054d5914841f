#!/bin/bash
# Round 6, session 2: the block-exponent band test, the new bench fields, the bf16 record.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_s2
mkdir -p $O
export GANGEALING_SYNTHETIC=1 TMPDIR=/tmp
cd $R
timeout 900 python -m pytest -q -m gpu tests/test_gpu_block_exponent_band.py tests/test_gpu_split_precision.py tests/test_gpu_convt16.py \
   tests/test_gpu_rccl_single_rank.py tests/test_gpu_bench_smoke.py "tests/test_gpu_configs.py::test_config_c2_one_limb_bf16_error_is_recorded" 2>&1 | tail -40 > $O/pytest.txt
cat $O/pytest.txt
cp gpurun_out/parity_report.json $O/parity_partial.json 2>/dev/null
python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r06_s2/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
print(json.dumps(d['roofline'].get('step'), indent=1))
print(d['roofline'].get('traffic_source'))
PY
python bench.py --allreduce-only --steps 2 | head -c 600
