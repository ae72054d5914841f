#!/bin/bash
# bf16 single-limb mode: speed + layer-op error; fresh kernel stats of the default bench
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
for p in bf16x3 bf16 fp32; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --precision $p > $O/s6_bench_$p.json 2> $O/s6_bench_$p.err
done
for b in 5 8; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --batch $b > $O/s6_bench_b$b.json 2>> $O/s6_bench.err
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/s6_trace -o trace --output-format rocpd -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/s6_bench_traced.json 2>/dev/null
cd $R
DB=$(find $O/s6_trace -name "*.db" | head -1)
python scripts/rocpd_stats.py $DB 80 > $O/s6_kernel_stats.txt 2>&1
rm -rf $O/s6_trace
for f in $O/s6_bench_bf16x3.json $O/s6_bench_bf16.json $O/s6_bench_fp32.json $O/s6_bench_b5.json $O/s6_bench_b8.json; do python -c "
import json,sys
d=json.load(open('$f')); print('$f'.split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"; done
head -50 $O/s6_kernel_stats.txt | cut -c1-180
