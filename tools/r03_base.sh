#!/bin/bash
# round-3 measurement batch: GPU tests, the bench line, the 1/8 shards (tuning aid)
set -u
tag=${1:-r03a}
out=gpurun_out/$tag
mkdir -p $out
timeout -k 10 400 python -m pytest tests -m gpu -x -q > $out/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $out/gpu_tests.log
timeout -k 10 300 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print('bench %.1f spectra/s %.3f ms/step frac %.3f' % (d['value'], d['ms_per_step'], r['frac']))
print('serial: prep %.3f' % r['sr_prep_kernel_ms'], {k[:22]: round(v['ms'],3) for k,v in r['kernels'].items()})
PY
for s in ${SHARDS:-0/8 3/8 7/8}; do timeout -k 10 120 python bench.py --shard $s --cpu-seconds 0 --steps 40 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('shard %s: %.3f ms/step  op %.3f  serial: prep %.3f ff %.3f wings %.3f zones %.3f' % (d['config']['sharding'][11:14], d['ms_per_step'], r['coefficient_op_ms_in_timed_steps'], r['sr_prep_kernel_ms'], *[v['ms'] for v in r['kernels'].values()]))" | tee -a $out/shards.txt; done
