#!/bin/bash
# SQ counters of chosen kernels of ANY python script for several library variants on one box:
#   tools/pmc_cmd.sh tag "kernel-substring" "script.py args" "" build/variants/x.so ...   ("" = the in-tree library)
set -u
tag=$1; pat=$2; cmd=$3; shift 3
out=gpurun_out/$tag; mkdir -p $out
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - > /dev/null
i=0
for lib in "$@"; do
  i=$((i+1))
  for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU" \
              "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
    p=$(echo $pass | cut -d' ' -f1)
    rm -rf /tmp/pcm_${i}_$p
    SPECTROBOT_HIP_LIB=$lib timeout -k 10 300 rocprofv3 --pmc $pass -d /tmp/pcm_${i}_$p -o p -- python3 $cmd > $out/pmc_${i}_$p.log 2>&1
    echo "variant $i [$lib] pass $p exit=$?"
    echo "== variant $i [$lib]" >> $out/pmc_cmd.txt
    python3 tools/rocprof_summary.py /tmp/pcm_${i}_$p/p_results.db 2>/dev/null | grep -i "$pat\|^kernel " >> $out/pmc_cmd.txt
  done
done
cat $out/pmc_cmd.txt
