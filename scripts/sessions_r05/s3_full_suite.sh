#!/bin/bash
# Round 5, session 3: the whole GPU suite at the new defaults (16-channel-chunk transposed tile per launch, 1-bit sign
# plane for the masked data gradients), then the step with / without the sign plane on the same box.
#   gpurun --timeout 1800 -- 'bash scripts/sessions_r05/s3_full_suite.sh'
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
OUT=$R/gpurun_out/r05
mkdir -p $OUT
export GANGEALING_SYNTHETIC=1 TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_sign_bits.py -q -x > $OUT/s3_pytest_sign_bits.txt 2>&1
tail -n 12 $OUT/s3_pytest_sign_bits.txt
timeout 1200 python -m pytest tests -q -m gpu > $OUT/s3_pytest_gpu.txt 2>&1
tail -n 15 $OUT/s3_pytest_gpu.txt
cp gpurun_out/parity_report.json $OUT/s3_parity_report.json 2>/dev/null
for rep in 1 2; do
  for dis in none sign_bits; do
    if [ $dis = none ]; then unset GG_DISABLE; else export GG_DISABLE=$dis; fi
    timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $OUT/s3_bench_${dis}_$rep.json 2>> $OUT/s3_bench_err.txt
  done
done
unset GG_DISABLE
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r05/s3_bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        ks = {k['kernel']: (k['launches_per_step'], k['ms_per_step'], k['rate']) for k in d['roofline'].get('kernels', [])}
        print(f, d['ms_per_step'], d['value'])
        for k, v in ks.items():
            print('    ', k, v)
    except Exception as e:
        print(f, 'unreadable', e)
PY
