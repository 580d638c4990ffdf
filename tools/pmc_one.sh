#!/bin/bash
# one --pmc pass per counter name (unknown names just fail): tools/pmc_one.sh "kernel-substring" COUNTER ...
pat=$1; shift
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - > /dev/null
for c in "$@"; do
  rm -rf /tmp/p1_$c
  timeout -k 10 200 rocprofv3 --pmc $c -d /tmp/p1_$c -o p -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 > /tmp/p1_$c.log 2>&1
  rc=$?
  if [ -f /tmp/p1_$c/p_results.db ]; then python3 tools/rocprof_summary.py /tmp/p1_$c/p_results.db 2>/dev/null | grep "$pat" | grep "$c" | cut -c1-60,63-130; else echo "$c: no database (rc $rc)"; fi
done
