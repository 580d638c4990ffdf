#!/bin/bash
# A/B of the level-table build on one box: tools/ab_tables.sh variant.so [reps]  (in-tree library against the variant, N = 1e5 and 2e5)
v=$1; n=${2:-2}
for i in $(seq $n); do for N in 100000 200000; do for lib in "" $v; do
  echo -n "[N=$N ${lib:-in-tree}] "; N=$N ROUTE=1 SPECTROBOT_HIP_LIB=$lib python3 tools/level_tables_probe.py 2>/dev/null | tr '\n' ' '; echo
done; done; done
