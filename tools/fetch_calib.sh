#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of tools/microbench/fetch_calib (1 GiB touched once per kernel): the counters' scale per access pattern
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 120 rocprofv3 --pmc $c -d /tmp/fc_$c -o p -- ./tools/microbench/fetch_calib > /dev/null 2>&1
  python3 - <<PY
import sqlite3
c = sqlite3.connect("/tmp/fc_$c/p_results.db")
for n, k, v in c.execute("select kernel_name, count(distinct dispatch_id), sum(value) from counters_collection where counter_name='$c' group by kernel_name"):
    print("%-12s %-40s %d launches: %.4f GB per launch reported (x1024: KiB) for 1.0737 GB touched -> factor %.3f" % ("$c", n.split("(")[0], k, v * 1024 / k / 1e9, v * 1024 / k / (1 << 30)))
PY
done
