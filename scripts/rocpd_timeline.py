#!/usr/bin/env python
"""One training step of a rocprofv3 rocpd database (--kernel-trace) in launch order: start offset, duration, gap to the
previous kernel's end, short kernel name.  The step is found as the window between two consecutive launches of an anchor
kernel that runs once per step (default: adam_ema_kernel's first launch of a step = the one after a long idle gap is not
assumed - we simply take the launches between the k-th and (k+1)-th occurrence of the anchor's FIRST call in a step).

usage: rocpd_timeline.py trace.db [anchor substring] [which step]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'\(.*$', '', name)
    name = name.replace('at::native::', '')
    return name[:110]


def main(db, anchor='pack_weight_many_kernel', which=3):
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = 'name' if 'name' in cols else 'kernel_name'
    rows = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
    idx = [i for i, r in enumerate(rows) if anchor in r[0]]
    # the anchor may run several times per step: steps are separated by its largest recurring gap pattern; take
    # occurrences that are preceded by >= 50 other launches since the previous occurrence as step starts
    starts = [i for k, i in enumerate(idx) if k == 0 or i - idx[k - 1] > 50]
    if len(starts) < which + 2:
        print(f'# only {len(starts)} anchors found ({anchor})')
        which = max(0, len(starts) - 2)
    lo, hi = starts[which], starts[which + 1]
    t0 = rows[lo][1]
    print(f'# step {which}: launches {lo}..{hi - 1} ({hi - lo} kernels), span {(rows[hi][1] - t0) / 1e6:.3f} ms, '
          f'kernel time {sum(r[2] - r[1] for r in rows[lo:hi]) / 1e6:.3f} ms')
    print(f'{"t_ms":>8} {"dur_us":>8} {"gap_us":>7}  kernel')
    prev_end = t0
    for name, s, e in rows[lo:hi]:
        print(f'{(s - t0) / 1e6:8.3f} {(e - s) / 1e3:8.1f} {(s - prev_end) / 1e3:7.1f}  {short(name)}')
        prev_end = max(prev_end, e)


if __name__ == '__main__':
    a = sys.argv
    main(a[1], a[2] if len(a) > 2 else 'pack_weight_many_kernel', int(a[3]) if len(a) > 3 else 3)
