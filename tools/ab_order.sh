#!/bin/bash
# far-field order / admissibility variants on one box: error against the exact mode, headline, level tables
# tools/ab_order.sh "" build/variants/kfd19.so ...
mkdir -p gpurun_out/r06
for lib in "$@"; do echo "=== ${lib:-in-tree}"; SPECTROBOT_HIP_LIB=$lib timeout -k 10 200 python3 tools/farfield_error.py 2>/dev/null | head -2 | cut -c1-140
  for i in 1 2; do SPECTROBOT_HIP_LIB=$lib python3 bench.py --cpu-seconds 0 --steps 40 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline %.1f spectra/s %.3f ms' % (d['value'], d['ms_per_step']))"; done
  for N in 100000 200000; do N=$N ROUTE=1 SPECTROBOT_HIP_LIB=$lib python3 tools/level_tables_probe.py 2>/dev/null | tail -1; done; done
