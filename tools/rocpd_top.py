#!/usr/bin/env python
"""Per-kernel summary of a rocprofv3 rocpd database: python tools/rocpd_top.py <results.db> [name-filter]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ''
rows = c.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3 from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
for n, cnt, t, a, mn in rows:
    if flt in n:
        print('%-70s n=%5d total %10.1f us avg %8.1f min %8.1f  %5.1f%%' % (n[:70], cnt, t, a, mn, 100 * t / tot))
