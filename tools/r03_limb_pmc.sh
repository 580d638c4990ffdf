#!/bin/bash
# HBM-side bytes and VALU counters of the configs[2] / [3] radiance kernels (separate --pmc passes)
set -u
out=gpurun_out/r03_limb_pmc
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for cfg in 2 3; do
  for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_THREAD_CYCLES_VALU"; do
    tag=c${cfg}_$(echo $pass | cut -d' ' -f1)
    timeout -k 10 300 rocprofv3 --pmc $pass -d /tmp/lp_$tag -o p -- python3 bench.py --config $cfg --steps 4 --warmup 1 --cpu-seconds 0 > $out/$tag.log 2>&1
    echo "$tag exit=$?"
    python3 tools/rocprof_summary.py /tmp/lp_$tag/p_results.db 2>/dev/null | grep -i "adjoint\|sr_limb\|los_col\|adj_pack\|^kernel " >> $out/config${cfg}_pmc_limb_kernels.txt
  done
done
cat $out/config2_pmc_limb_kernels.txt $out/config3_pmc_limb_kernels.txt
