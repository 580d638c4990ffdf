#!/bin/bash
# On the GPU box: per-kernel times (tools/bench_modes.py) of the default library and of every build/variants/*.so;
# CHECK=1 also runs the far-field-vs-exact parity tests with each variant.
set -u
mkdir -p gpurun_out
out=gpurun_out/variants.txt
: > $out
python tools/bench_modes.py 2>/dev/null | tee -a $out
for lib in build/variants/*.so; do
  export SPECTROBOT_HIP_LIB=$PWD/$lib
  timeout -k 10 120 python tools/bench_modes.py 2>/dev/null | tee -a $out
  if [ "${CHECK:-0}" = "1" ]; then
    timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "far_field_vs_exact or randomized or e2e_ch4" 2>&1 | tail -2 | tee -a $out
  fi
  if [ -n "${SHARD:-}" ]; then
    timeout -k 10 120 python bench.py --shard $SHARD --cpu-seconds 0 --steps 100 --warmup 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('   shard $SHARD: %.3f ms/step  op %.3f  serial: prep %.3f ff %.3f wings %.3f zones %.3f' % (d['ms_per_step'], r['coefficient_op_ms_in_timed_steps'], r['sr_prep_kernel_ms'], *[v['ms'] for v in r['kernels'].values()]))" | tee -a $out
  fi
  unset SPECTROBOT_HIP_LIB
done
