#!/bin/bash
# HIP API calls (host) against kernel dispatches (device) over two steps of a shard run
set -u
s=${1:-3/8}
out=/tmp/ht_$$
mkdir -p $out gpurun_out/ht
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 200 rocprofv3 --hip-trace --kernel-trace -d $out/ht -o ht -- python3 bench.py --shard $s --cpu-seconds 0 --steps 30 --warmup 5 > gpurun_out/ht/ht.log 2>&1
echo "rc=$?"
python3 - <<PY > gpurun_out/ht/timeline.txt
import sqlite3, re
c = sqlite3.connect("$out/ht/ht_results.db")
cols = [r[1] for r in c.execute("pragma table_info(regions)").fetchall()]
print("regions cols:", cols)
k = c.execute("select name, start, end, stream_id from kernels order by start").fetchall()
idx = [i for i, r in enumerate(k) if "sr_prep_kernel" in r[0]]
a = idx[20]; t0 = k[a][1]; t1 = k[idx[22]][1]
ev = [("K", re.sub(r"^void ", "", n).split("(")[0].replace("sr::", "")[:40], s, e, st) for n, s, e, st in k if t0 - 2.5e6 <= s <= t1]
r = c.execute("select name, start, end, tid from regions where start >= ? and start <= ? order by start", (t0 - 2.5e6, t1)).fetchall()
ev += [("A", n, s, e, tid) for n, s, e, tid in r]
ev.sort(key=lambda x: x[2])
for kind, n, s, e, x in ev:
    print("%s %-44s t %9.1f dur %8.1f us  [%s]" % (kind, n, (s - t0) / 1e3, (e - s) / 1e3, x))
PY
wc -l gpurun_out/ht/timeline.txt
