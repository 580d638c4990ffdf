#!/bin/bash
# SQ counters of chosen kernels for several library variants on ONE box (run on the GPU box):
#   tools/pmc_ab.sh tag "kernel-substring" "" build/variants/x.so ...   ("" = the in-tree library)
# Two --pmc passes per variant (never combined with API traces), three steps of the headline each.
set -u
tag=$1; pat=$2; shift 2
out=gpurun_out/$tag; mkdir -p $out
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - > /dev/null
i=0
for lib in "$@"; do
  i=$((i+1))
  for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU" \
              "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
    p=$(echo $pass | cut -d' ' -f1)
    rm -rf /tmp/pab_${i}_$p
    SPECTROBOT_HIP_LIB=$lib timeout -k 10 300 rocprofv3 --pmc $pass -d /tmp/pab_${i}_$p -o p -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 > $out/pmc_${i}_$p.log 2>&1
    echo "variant $i [$lib] pass $p exit=$?"
    echo "== variant $i [$lib]" >> $out/pmc_ab.txt
    python3 tools/rocprof_summary.py /tmp/pab_${i}_$p/p_results.db 2>/dev/null | grep -i "$pat\|^kernel " >> $out/pmc_ab.txt
  done
done
cat $out/pmc_ab.txt
