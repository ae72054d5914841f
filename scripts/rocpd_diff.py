#!/usr/bin/env python
"""Per-kernel difference of two rocprofv3 rocpd databases (--kernel-trace) of the same command: total ms and calls per
kernel name in A and B, sorted by |difference|.  usage: rocpd_diff.py a.db b.db [steps] [rows]"""
import sqlite3
import sys


def load(db):
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = 'name' if 'name' in cols else 'kernel_name'
    return {n: (c, t) for n, c, t in cur.execute(f"select {name_col}, count(*), sum(end-start) from kernels group by {name_col}")}


def main(a, b, steps=1, rows=40):
    A, B = load(a), load(b)
    ta, tb = sum(v[1] for v in A.values()), sum(v[1] for v in B.values())
    print(f'# A {ta / 1e6 / steps:.3f} ms/step ({sum(v[0] for v in A.values()) / steps:.0f} launches/step)   '
          f'B {tb / 1e6 / steps:.3f} ms/step ({sum(v[0] for v in B.values()) / steps:.0f} launches/step)   '
          f'B - A {(tb - ta) / 1e6 / steps:+.3f} ms/step')
    names = sorted(set(A) | set(B), key=lambda n: -abs(B.get(n, (0, 0))[1] - A.get(n, (0, 0))[1]))
    print(f'{"A ms/step":>10} {"B ms/step":>10} {"B - A":>8} {"A calls":>8} {"B calls":>8}  name')
    for n in names[:rows]:
        ca, xa = A.get(n, (0, 0))
        cb, xb = B.get(n, (0, 0))
        print(f'{xa / 1e6 / steps:10.3f} {xb / 1e6 / steps:10.3f} {(xb - xa) / 1e6 / steps:+8.3f} {ca / steps:8.1f} {cb / steps:8.1f}  {n[:120]}')


if __name__ == '__main__':
    a = sys.argv
    main(a[1], a[2], int(a[3]) if len(a) > 3 else 1, int(a[4]) if len(a) > 4 else 40)
