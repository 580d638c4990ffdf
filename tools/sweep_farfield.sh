#!/bin/bash
# Sweep the far-field admissibility distance (kTheta) and expansion degree (kFD):
# rebuild the library for each pair, check the two evaluation modes against each
# other (tests hold the tolerance) and time the default bench.  Run on the GPU box:
#   gpurun -- 'bash tools/sweep_farfield.sh "8 14" "6 17" "5 19" "4 22"'
# Edits spectrobot_amd/csrc/sr_kernels.hpp in place (restored at the end).
set -u
hpp=spectrobot_amd/csrc/sr_kernels.hpp
cp "$hpp" /tmp/sr_kernels.hpp.orig
mkdir -p gpurun_out
out=gpurun_out/sweep_farfield.txt
: > "$out"
for pair in "$@"; do
  set -- $pair
  th=$1; fd=$2
  sed -e "s/^constexpr int kTheta = [0-9]*;/constexpr int kTheta = $th;/" \
      -e "s/^constexpr int kFD = [0-9]*;/constexpr int kFD = $fd;/" /tmp/sr_kernels.hpp.orig > "$hpp"
  echo "=== theta=$th degree=$fd" | tee -a "$out"
  python spectrobot_amd/build.py --force >> "$out" 2>&1 || { echo "build failed" | tee -a "$out"; continue; }
  timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -x -m gpu \
      -k "far_field_vs_exact or randomized or e2e_ch4" 2>&1 | tail -4 | tee -a "$out"
  timeout -k 10 200 python tools/farfield_error.py 2>&1 | tee -a "$out"
  timeout -k 10 200 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 2>&1 | tail -1 | tee -a "$out"
done
cp /tmp/sr_kernels.hpp.orig "$hpp"
