#!/bin/bash
# Round 6, session 1: the boundary nits + the RCCL-captured graph on hardware, a bench line of the starting build, and the
# eager-vs-hipGraph timeline.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_s1
mkdir -p $O
export GANGEALING_SYNTHETIC=1 TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest -q -m gpu -x tests/test_gpu_ops.py tests/test_gpu_indices.py tests/test_gpu_rccl_single_rank.py tests/test_gpu_ddp.py \
  tests/test_gpu_bench_smoke.py tests/test_gpu_determinism.py "tests/test_gpu_models.py::test_graph_replay_trainer" 2>&1 | tail -15 > $O/pytest_targeted.txt
cat $O/pytest_targeted.txt
python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_start.json 2> $O/bench_start.err
head -c 400 $O/bench_start.json; echo
cd /tmp
for mode in eager graph; do
  G=""; [ $mode = graph ] && G="--graph"
  for b in 16 5; do
    timeout 600 rocprofv3 --kernel-trace -d $O/trace_${mode}_b$b -o trace --output-format rocpd -- \
      python $R/bench.py --batch $b $G --steps 8 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_${mode}_b${b}_traced.json 2>/dev/null
    DB=$(find $O/trace_${mode}_b$b -name "*.db" | head -1)
    python $R/scripts/rocpd_gaps.py $DB > $O/gaps_${mode}_b$b.txt 2>&1
    rm -rf $O/trace_${mode}_b$b
    # untraced, same box
    python $R/bench.py --batch $b $G --steps 30 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$mode b$b untraced', d['value'], d['ms_per_step'])" >> $O/graph_vs_eager.txt
  done
done
cat $O/graph_vs_eager.txt $O/gaps_*.txt
