#!/bin/bash
# rocprofv3 passes behind the numbers in DESIGN.md / bench.py (run on the GPU box):
#   gpurun -- 'bash tools/profile.sh r01_v5'
# Writes gpurun_out/prof_<tag>/...; copy the summaries into profiles/ afterwards:
#   <tag>_kernel_trace_stats.txt  per-kernel times   (--kernel-trace --stats)
#   <tag>_pmc_summary.txt         SQ instruction / cycle counters
#   <tag>_pmc_hbm.txt             FETCH_SIZE / WRITE_SIZE (separate passes, as the guide prescribes)
# Counter passes never combine --pmc with API traces (the pool refuses that).
set -u
tag=${1:-prof}
out=gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - > /dev/null
B="python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0"
run() { # name, rocprof args...
  local name=$1; shift
  timeout -k 10 300 rocprofv3 "$@" -d "$out/$name" -o "$name" -- $B > "$out/$name.log" 2>&1
  echo "$name exit=$?"
}
run kt --kernel-trace --stats &&
run pmc1 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU &&
run pmc2 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA &&
run fetch --pmc FETCH_SIZE &&
run write --pmc WRITE_SIZE &&
run tcc --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum
python3 tools/rocprof_summary.py "$out"/kt/kt_results.db > "$out/${tag}_kernel_trace_stats.txt"
python3 tools/rocprof_summary.py "$out"/pmc1/pmc1_results.db "$out"/pmc2/pmc2_results.db > "$out/${tag}_pmc_summary.txt"
python3 tools/rocprof_summary.py "$out"/fetch/fetch_results.db "$out"/write/write_results.db "$out"/tcc/tcc_results.db > "$out/${tag}_pmc_hbm.txt"
python3 tools/pmc_hbm_json.py "$tag" "$out"/fetch/fetch_results.db "$out"/write/write_results.db > "$out/${tag}_pmc_hbm.json"
timeout -k 10 300 python3 bench.py > "$out/${tag}_bench.json" 2> "$out/bench.err"
echo "bench exit=$?"
tail -c 600 "$out/${tag}_bench.json"
