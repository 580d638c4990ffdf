#!/bin/bash
# Sweep the far-field admissibility distance (kTheta) and expansion degree (kFD): build a VARIANT
# library per pair (-DSR_KTHETA / -DSR_KFD, separate output file: the tree and the default library are
# never touched), check the two evaluation modes against each other (tests hold the tolerance) and
# time the default bench.  Run on the GPU box:
#   gpurun -- 'bash tools/sweep_farfield.sh "8 14" "6 17" "5 19" "4 22"'
set -u
mkdir -p gpurun_out
out=gpurun_out/sweep_farfield.txt
: > "$out"
for pair in "$@"; do
  set -- $pair
  th=$1; fd=$2
  lib=$PWD/gpurun_out/libspectrobot_hip_t${th}_d${fd}.so
  echo "=== theta=$th degree=$fd" | tee -a "$out"
  python spectrobot_amd/build.py --out "$lib" -DSR_KTHETA=$th -DSR_KFD=$fd >> "$out" 2>&1 || { echo "build failed" | tee -a "$out"; continue; }
  export SPECTROBOT_HIP_LIB=$lib
  timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -x -m gpu \
      -k "far_field_vs_exact or randomized or e2e_ch4" 2>&1 | tail -4 | tee -a "$out"
  timeout -k 10 200 python tools/farfield_error.py 2>&1 | tee -a "$out"
  timeout -k 10 200 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 2>&1 | tail -1 | tee -a "$out"
  unset SPECTROBOT_HIP_LIB
  rm -f "$lib"
done
