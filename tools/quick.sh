#!/bin/bash
# quick GPU check of the default library: parity subset (or all GPU tests with ALL=1), per-kernel times, one shard
if [ "${ALL:-0}" = "1" ]; then python -m pytest tests -m gpu -x -q 2>&1 | tail -3; else
python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "far_field_vs_exact or randomized or e2e_ch4 or outer or counters" 2>&1 | tail -2; fi
python tools/bench_modes.py 2>/dev/null
python bench.py --shard ${SHARD:-3/8} --cpu-seconds 0 --steps 100 --warmup 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('shard: %.3f ms/step  op %.3f  serial: prep %.3f ff %.3f wings %.3f zones %.3f' % (d['ms_per_step'], r['coefficient_op_ms_in_timed_steps'], r['sr_prep_kernel_ms'], *[v['ms'] for v in r['kernels'].values()]))"
