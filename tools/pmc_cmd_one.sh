#!/bin/bash
# one --pmc pass per counter name for any python script: tools/pmc_cmd_one.sh "kernel-substring" "script.py args" COUNTER ...
pat=$1; cmd=$2; shift 2
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - > /dev/null
for c in "$@"; do
  rm -rf /tmp/pc1_$c
  timeout -k 10 200 rocprofv3 --pmc $c -d /tmp/pc1_$c -o p -- python3 $cmd > /tmp/pc1_$c.log 2>&1
  rc=$?
  if [ -f /tmp/pc1_$c/p_results.db ]; then python3 tools/rocprof_summary.py /tmp/pc1_$c/p_results.db 2>/dev/null | grep "$pat" | grep "$c" | cut -c1-60,63-130; else echo "$c: no database (rc $rc)"; fi
done
