#!/bin/bash
# Round-3 session Q: clock and matrix-pipe occupancy of the dominant tile on random vs all-zero operands
# (GRBM_GUI_ACTIVE / duration = effective clock; SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GUI_ACTIVE/8) = occupancy).
# Dispatch order in the csv: per layer (512@64^2, 128@256^2): 20 warm-up + 40 timed launches on random data, then on zeros.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03q
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PROBE_MODES=fp16x3 PROBE_CASES=0,4 ITERS=40
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc -- python $R/scripts/power_probe.py > $O/probe_under_pmc.txt 2>&1
cd $R
python - <<'PY' > $O/summary.txt 2>&1
import csv, glob, os
O = os.environ.get('GRAFT_REPO_ROOT', '/root/repo') + '/gpurun_out/r03q'
cc = glob.glob(O + '/pmc/**/*counter_collection.csv', recursive=True)
kt = glob.glob(O + '/pmc/**/*kernel_trace.csv', recursive=True)
print('files', cc, kt)
dur = {}
for path in kt:
    for row in csv.DictReader(open(path)):
        if 'conv3x3_patch_kernel' in row['Kernel_Name']:
            dur[row['Dispatch_Id']] = (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3
vals = {}
for path in cc:
    for row in csv.DictReader(open(path)):
        if 'conv3x3_patch_kernel' in row['Kernel_Name']:
            vals.setdefault(row['Dispatch_Id'], {})[row['Counter_Name']] = float(row['Counter_Value'])
ids = sorted(vals, key=int)
print('dispatches', len(ids), 'with duration', sum(1 for i in ids if i in dur))
# groups of 60 launches (20 warm-up + 40 timed) in program order
for gi in range(0, len(ids), 60):
    grp = [i for i in ids[gi + 20:gi + 60]]
    if not grp:
        continue
    n = len(grp)
    gui = sum(vals[i].get('GRBM_GUI_ACTIVE', 0) for i in grp) / n
    mf = sum(vals[i].get('SQ_VALU_MFMA_BUSY_CYCLES', 0) for i in grp) / n
    d = [dur[i] for i in grp if i in dur]
    du = sum(d) / len(d) if d else float('nan')
    print(f'group {gi // 60}: launches {n}  duration {du:8.1f} us  GRBM_GUI_ACTIVE/8 {gui / 8:12.0f}  -> clock {gui / 8 / du / 1e3:5.2f} GHz'
          f'  MFMA busy cycles/SIMD {mf / 1024:12.0f} -> occupancy {mf / 1024 / (gui / 8):5.3f}')
PY
cat $O/summary.txt; grep -v amdgpu $O/probe_under_pmc.txt | tail -6
rm -rf $O/pmc
