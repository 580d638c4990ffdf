#!/bin/bash
# kernel timeline of the LAST level-table build of tools/level_tables_probe.py (ROUTE=1): tools/kt_timeline.sh tag
tag=$1; mkdir -p gpurun_out/r06; d=gpurun_out/r06/tl_$tag
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
ROUTE=${ROUTE:-1} REPS=3 rocprofv3 --kernel-trace -d $d -o kt -- python3 tools/level_tables_probe.py > $d.log 2>&1
python3 - <<PY > gpurun_out/r06/tl_$tag.txt
import sqlite3, re
c = sqlite3.connect("$d/kt_results.db")
rows = c.execute("select name, start, end, stream_id from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "sr_zones_mc_kernel" in r[0]]
a = idx[-1] - 1          # the MC prep in front of the last zones launch
t0 = rows[a][1]; busy_end = None
for name, s, e, st in rows[a:]:
    short = re.sub(r"^void ", "", name).split("(")[0].replace("sr::", "")[:40]
    gap = "" if busy_end is None or s <= busy_end else "   <- idle %.1f us" % ((s - busy_end) / 1e3)
    print("%-42s start %9.1f end %9.1f dur %8.1f us stream %s%s" % (short, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, st, gap))
    busy_end = e if busy_end is None else max(busy_end, e)
PY
rm -rf $d; cat gpurun_out/r06/tl_$tag.txt
