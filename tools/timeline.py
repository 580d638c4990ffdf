#!/usr/bin/env python3
"""Kernel timeline of a rocprofv3 --kernel-trace database: start, duration, stream of every dispatch in a window,
and the device-idle gaps between them.  usage: tools/timeline.py <results.db> [first_row [n_rows]]"""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end, stream_id from kernels order by start").fetchall()
a = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n = int(sys.argv[3]) if len(sys.argv) > 3 else 60
t0 = rows[0][1]
busy_end = None
for name, s, e, st in rows[a:a + n]:
    short = re.sub(r"^void ", "", name).split("(")[0].replace("sr::", "")[:44]
    gap = "" if busy_end is None or s <= busy_end else "   <- idle %.3f ms" % ((s - busy_end) / 1e6)
    print("%-46s start %10.3f dur %8.3f ms stream %s%s" % (short, (s - t0) / 1e6, (e - s) / 1e6, st, gap))
    busy_end = e if busy_end is None else max(busy_end, e)
