#!/bin/bash
# kernel + copy timeline of the last retrieval iterations of `bench.py --config 4`: tools/kt_config4.sh tag
tag=$1; mkdir -p gpurun_out/r06; d=gpurun_out/r06/c4_$tag
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace ${COPIES:+--memory-copy-trace} -d $d -o kt -- python3 bench.py --config 4 > $d.log 2>&1
python3 - <<PY > gpurun_out/r06/c4_$tag.txt
import sqlite3, re
c = sqlite3.connect("$d/kt_results.db")
rows = [(n, s, e) for n, s, e in c.execute("select name, start, end from kernels")]
try:
    rows += [("copy " + str(n), s, e) for n, s, e in c.execute("select name, start, end from memory_copies")]
except Exception as ex:
    print("# no copies:", ex)
rows.sort(key=lambda r: r[1])
idx = [i for i, r in enumerate(rows) if "vmr_from_params" in r[0]]
a = idx[len(idx) // 2]          # an iteration in the middle of the timed retrievals
b = idx[len(idx) // 2 + 3]
t0 = rows[a][1]; last_end = None
for name, s, e in rows[a:b]:
    short = re.sub(r"^void ", "", name).split("(")[0].replace("sr::", "")[:44]
    gap = "" if last_end is None else "   gap %6.1f us" % ((s - last_end) / 1e3)
    print("%-46s start %9.1f dur %8.1f us%s" % (short, (s - t0) / 1e3, (e - s) / 1e3, gap))
    last_end = e
PY
rm -rf $d; cat gpurun_out/r06/c4_$tag.txt; tail -3 $d.log | cut -c1-400
