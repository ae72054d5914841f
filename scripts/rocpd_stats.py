#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (--kernel-trace) into a per-kernel stats table
(name, calls, total ms, avg us, min/max us, % of GPU kernel time)."""
import sqlite3
import sys


def main(db, top=40):
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = 'name' if 'name' in cols else 'kernel_name'
    rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       f"from kernels group by {name_col} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    span = cur.execute("select min(start), max(end) from kernels").fetchone()
    print(f'# kernels: {sum(r[1] for r in rows)} dispatches, {len(rows)} distinct, GPU kernel time {total / 1e6:.2f} ms, '
          f'trace span {(span[1] - span[0]) / 1e6:.2f} ms')
    print(f'{"calls":>7} {"total_ms":>10} {"avg_us":>10} {"min_us":>9} {"max_us":>9} {"pct":>6}  name')
    for name, n, tot, avg, mn, mx in rows[:top]:
        short = name if len(name) < 150 else name[:147] + '...'
        print(f'{n:7d} {tot / 1e6:10.3f} {avg / 1e3:10.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f} {100 * tot / total:6.2f}  {short}')


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
