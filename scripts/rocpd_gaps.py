#!/usr/bin/env python
"""Idle-gap statistics of a rocprofv3 rocpd database (--kernel-trace): for the launches between the first and the last
occurrence of an anchor kernel that runs once per step - kernel time, wall span, idle time between consecutive kernels
(overlap-aware: the gap in front of a kernel is its start minus the latest end seen so far), how many gaps exceed 1 / 5 /
20 us and which kernels follow the largest ones.  Used to compare the eager step with its hipGraph replay.

usage: rocpd_gaps.py trace.db [anchor substring]"""
import collections
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'\(.*$', '', name)
    return name.replace('at::native::', '')[:90]


def main(db, anchor='pack_weight_many_kernel'):
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = 'name' if 'name' in cols else 'kernel_name'
    rows = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
    idx = [i for i, r in enumerate(rows) if anchor in r[0]]
    starts = [i for k, i in enumerate(idx) if k == 0 or i - idx[k - 1] > 50]
    if len(starts) < 3:
        print(f'# only {len(starts)} anchors ({anchor}) in {len(rows)} launches')
        return
    lo, hi = starts[1], starts[-1]                  # skip the first (warm-up / capture) step
    nsteps = len(starts) - 2
    seg = rows[lo:hi]
    ktime = sum(e - s for _, s, e in seg)
    span = rows[hi][1] - seg[0][1]
    gaps, after = [], collections.Counter()
    latest = seg[0][2]
    for name, s, e in seg[1:]:
        g = s - latest
        if g > 0:
            gaps.append(g)
            if g > 5000:
                after[short(name)] += g
        latest = max(latest, e)
    idle = sum(gaps)
    print(f'# {nsteps} steps, {len(seg) / nsteps:.0f} launches/step: span {span / 1e6 / nsteps:.3f} ms/step, kernel time '
          f'{ktime / 1e6 / nsteps:.3f} ms/step, idle between kernels {idle / 1e6 / nsteps:.3f} ms/step '
          f'({len(gaps) / nsteps:.0f} gaps/step: mean {idle / max(len(gaps), 1) / 1e3:.2f} us; > 1 us: '
          f'{sum(g > 1000 for g in gaps) / nsteps:.0f}, > 5 us: {sum(g > 5000 for g in gaps) / nsteps:.0f}, > 20 us: '
          f'{sum(g > 20000 for g in gaps) / nsteps:.0f} per step)')
    print('# idle time in gaps > 5 us, by the kernel that FOLLOWS the gap (ms/step):')
    for name, t in after.most_common(12):
        print(f'  {t / 1e6 / nsteps:8.3f}  {name}')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else 'pack_weight_many_kernel')
