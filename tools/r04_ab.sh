#!/bin/bash
# A/B of the round-4 step variants on ONE box: bench.py headline (20 steps after 3) per environment setting, twice each.
# usage: tools/r04_ab.sh tag "ENV1=a ENV2=b" "ENV1=c" ...   (an empty string = defaults)
tag=$1; shift
out=gpurun_out/${tag}_ab.txt
: > $out
for rep in 1 2; do
  for envs in "$@"; do
    line=$(env $envs python bench.py --steps 40 --warmup 5 --cpu-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']; print('%.2f spectra/s  %.3f ms/step   serial: %s  prep %.3f' % (d['value'], d['ms_per_step'], '  '.join('%.3f' % v['ms'] for v in k.values()), d['roofline']['sr_prep_kernel_ms']))")
    echo "rep $rep [$envs] $line" | tee -a $out
  done
done
