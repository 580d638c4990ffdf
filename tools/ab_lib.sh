#!/bin/bash
# A/B of the headline on one box: tools/ab_lib.sh variant.so [reps]   (in-tree library against the variant; executed-work counters too)
v=$1; n=${2:-2}
for i in $(seq $n); do
for lib in "" $v; do
  SPECTROBOT_HIP_LIB=$lib python3 bench.py --cpu-seconds 0 --steps 40 2>/dev/null | grep "^{" | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; c=r['executed_counts']
print('[%s] %.3f ms/step  far field / wings / zones %s  region1 %d window_ends %d' % ('$lib' or 'in-tree', d['ms_per_step'], [round(v['ms'],3) for v in r['kernels'].values()], c['region1_evals'], c['window_end_expansions']))"
done; done
