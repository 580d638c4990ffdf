#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd databases (*.db) as text: per-kernel time stats
(--kernel-trace --stats runs) and per-kernel counter sums/averages (--pmc runs).

  python tools/rocprof_summary.py gpurun_out/prof/kt/kt_results.db [more.db ...] > profiles/xxx.txt
"""
import sqlite3
import sys


def short(n):
    n = n.replace("void ", "")
    return n.split("(")[0][:60]


for f in sys.argv[1:]:
    c = sqlite3.connect(f)
    print("== %s" % f)
    rows = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    if rows:
        print("%-62s %6s %14s %14s %7s" % ("kernel", "calls", "total_us", "avg_us", "%"))
        for n, k, t, a, p in rows:
            print("%-62s %6d %14.1f %14.1f %7.2f" % (short(n), k, t, a, p))
    try:
        rows = c.execute("select kernel_name, counter_name, count(distinct dispatch_id), sum(value), "
                         "avg(duration), max(vgpr_count), max(sgpr_count), max(lds_block_size) "
                         "from counters_collection group by kernel_name, counter_name").fetchall()
    except sqlite3.Error:
        rows = []
    if rows:
        print("%-62s %-24s %6s %18s %16s %12s" % ("kernel", "counter", "disp", "sum_all_dispatches",
                                                  "per_dispatch", "avg_dur_us"))
        for n, cn, k, v, d, vg, sg, lds in rows:
            print("%-62s %-24s %6d %18.6g %16.6g %12.1f  (vgpr %s sgpr %s lds %s)" % (
                short(n), cn, k, v, v / max(k, 1), (d or 0) / 1e3, vg, sg, lds))
    print()
