#!/bin/bash
# stand-alone duration of every kernel of the coefficient op (rocprofv3 --kernel-trace --stats, kernels one after
# the other): usage tools/kt.sh [lib.so]
set -u
[ -n "${1:-}" ] && export SPECTROBOT_HIP_LIB=$PWD/$1
out=/tmp/kt_$$
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
SR_SERIAL_ONLY=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $out -o kt -- python3 tools/bench_modes.py > $out.log 2>&1
python3 tools/rocprof_summary.py $out/kt_results.db | grep -v "^==" | awk '{printf "%-64s %6s %12s\n", $1" "$2, $(NF-3), $(NF-1)}' | head -16
