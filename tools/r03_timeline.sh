#!/bin/bash
# kernel timeline of one shard's steps (launch gaps between dependent kernels)
set -u
s=${1:-3/8}
out=gpurun_out/tl
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
sh="--shard $s"; [ "$s" = "full" ] && sh=""   # full: the whole grid
timeout -k 10 200 rocprofv3 --kernel-trace -d $out/kt -o kt -- python3 bench.py $sh --cpu-seconds 0 --steps 30 --warmup 5 > $out/kt.log 2>&1
echo "rc=$?"
python3 - <<PY
import sqlite3, re
c = sqlite3.connect("$out/kt/kt_results.db")
rows = c.execute("select name, start, end, stream_id from kernels order by start").fetchall()
# find the 20th occurrence of sr_prep_kernel and print two steps from there
idx = [i for i, r in enumerate(rows) if "sr_prep_kernel" in r[0]]
a = idx[20]
t0 = rows[a][1]; busy_end = None
for name, s, e, st in rows[a:a + 44]:
    short = re.sub(r"^void ", "", name).split("(")[0].replace("sr::", "")[:40]
    gap = "" if busy_end is None or s <= busy_end else "   <- idle %.1f us" % ((s - busy_end) / 1e3)
    print("%-42s start %9.1f dur %8.1f us stream %s%s" % (short, (s - t0) / 1e3, (e - s) / 1e3, st, gap))
    busy_end = e if busy_end is None else max(busy_end, e)
PY
