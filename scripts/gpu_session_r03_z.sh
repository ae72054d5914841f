#!/bin/bash
# Round-3 session Z: clock and matrix-pipe occupancy of the transposed tile (largest up-convolution), random vs zero operands
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03z
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export CONVT_ONLY="128->257" ITERS=40 GANGEALING_CONV_PRECISION=fp16x3
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc -- python $R/scripts/convt_probe.py > $O/probe.txt 2>&1
cd $R
python - <<'PY' > $O/summary.txt 2>&1
import csv, glob, os
O = os.environ.get('GRAFT_REPO_ROOT', '/root/repo') + '/gpurun_out/r03z'
cc = glob.glob(O + '/pmc/**/*counter_collection.csv', recursive=True)
kt = glob.glob(O + '/pmc/**/*kernel_trace.csv', recursive=True)
dur, vals = {}, {}
for path in kt:
    for row in csv.DictReader(open(path)):
        if 'convT3x3s2' in row['Kernel_Name']:
            dur[row['Dispatch_Id']] = (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3
for path in cc:
    for row in csv.DictReader(open(path)):
        if 'convT3x3s2' in row['Kernel_Name']:
            vals.setdefault(row['Dispatch_Id'], {})[row['Counter_Name']] = float(row['Counter_Value'])
ids = sorted(vals, key=int)
print('dispatches', len(ids))
# program order: 1 shape probe + 10 warm-up + 40 timed on random data, then 1 + 10 + 40 on zeros
for name, grp in (('random', ids[11:51]), ('zeros', ids[62:102])):
    n = len(grp)
    gui = sum(vals[i].get('GRBM_GUI_ACTIVE', 0) for i in grp) / n
    mf = sum(vals[i].get('SQ_VALU_MFMA_BUSY_CYCLES', 0) for i in grp) / n
    d = [dur[i] for i in grp if i in dur]
    du = sum(d) / len(d)
    print(f'{name:7s} launches {n}  duration {du:8.1f} us  GRBM_GUI_ACTIVE/8 {gui / 8:12.0f} -> clock {gui / 8 / du / 1e3:5.2f} GHz'
          f'  MFMA busy cycles/SIMD {mf / 1024:12.0f} -> occupancy {mf / 1024 / (gui / 8):5.3f}')
PY
cat $O/summary.txt; rm -rf $O/pmc
