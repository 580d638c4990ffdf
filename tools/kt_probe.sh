#!/bin/bash
# kernel-trace stats of tools/level_tables_probe.py (ROUTE / WHAT / N from the environment): tools/kt_probe.sh tag [lib.so]
tag=$1; [ -n "${2:-}" ] && export SPECTROBOT_HIP_LIB=$PWD/$2
mkdir -p gpurun_out/r06; d=gpurun_out/r06/kt_$tag
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
REPS=${REPS:-3} rocprofv3 --kernel-trace --stats -d $d -o kt -- python3 tools/level_tables_probe.py > $d.log 2>&1
python3 tools/rocprof_summary.py $d/kt_results.db > gpurun_out/r06/kt_${tag}_stats.txt; rm -rf $d
grep "route\|folded" $d.log; grep "_mc_kernel\|rows_kernel" gpurun_out/r06/kt_${tag}_stats.txt
