#!/bin/bash
# Round-6 evidence batch (GPU box): the rocprofv3 passes + bench line of the headline (tools/profile.sh), a counter-free
# kernel trace of the SERIAL step (what bench.py's serial kernel_ms must be reproducible from: VERDICT round 5, 2c),
# Joules per kernel and per step (tools/energy_by_kernel.py), the level-table build on both routes with its kernel
# trace, every 1/8 shard timed alone, the --config lines (2, 3, 3 --3d, 4, lut; 3 and lut also on the per-level route).
set -u
tag=${1:-r06_v1}
out=gpurun_out/$tag
mkdir -p $out
bash tools/profile.sh $tag > $out/profile.log 2>&1; tail -3 $out/profile.log
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
SR_SERIAL_ONLY=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /tmp/kt_serial -o kt -- python3 tools/bench_modes.py > $out/serial_kt.log 2>&1
python3 tools/rocprof_summary.py /tmp/kt_serial/kt_results.db > $out/serial_step_kernel_trace_stats.txt; head -12 $out/serial_step_kernel_trace_stats.txt
timeout -k 10 300 python3 tools/energy_by_kernel.py > $out/energy_by_kernel.txt 2> $out/energy.err; tail -12 $out/energy_by_kernel.txt
for what in pairs ctypes; do
  WHAT=$what ROUTE=both REPS=5 timeout -k 10 300 python3 tools/level_tables_probe.py 2>/dev/null > $out/level_tables_$what.txt; cat $out/level_tables_$what.txt
done
ROUTE=1 REPS=3 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_mc -o kt -- python3 tools/level_tables_probe.py > /dev/null 2>&1
python3 tools/rocprof_summary.py /tmp/kt_mc/kt_results.db > $out/level_tables_kernel_trace_stats.txt
{
for s in 0/8 1/8 2/8 3/8 4/8 5/8 6/8 7/8 0/4 0/2; do timeout -k 10 120 python bench.py --shard $s --cpu-seconds 0 --steps 100 --warmup 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('shard %s: %.3f ms/step  op %.3f  serial: prep %.3f ff %.3f wings %.3f zones %.3f' % (d['config']['sharding'][11:14], d['ms_per_step'], r['coefficient_op_ms_in_timed_steps'], r['sr_prep_kernel_ms'], *[v['ms'] for v in r['kernels'].values()]))"; done
timeout -k 10 100 python tools/host_overhead.py
} > $out/shards.txt 2>&1; cat $out/shards.txt
for c in 2 3 4 lut; do timeout -k 10 400 python bench.py --config $c --cpu-seconds 8 > $out/config$c.json 2> $out/config$c.err; echo "config $c rc=$?"; done
timeout -k 10 400 python bench.py --config 3 --3d --cpu-seconds 8 > $out/config3_3d.json 2> $out/config3_3d.err; echo "config 3 3d rc=$?"
for c in 3 lut; do timeout -k 10 400 python bench.py --config $c --cpu-seconds 0 --level-route 0 > $out/config${c}_per_level_route.json 2>/dev/null; done
timeout -k 10 400 python bench.py --config 3 --3d --cpu-seconds 0 --level-route 0 > $out/config3_3d_per_level_route.json 2>/dev/null
timeout -k 10 200 python bench.py --rays 64 --cpu-seconds 0 > $out/rays64.json 2>/dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_c3 -o kt -- python3 bench.py --config 3 --steps 4 --warmup 1 --cpu-seconds 0 > /dev/null 2>&1
python3 tools/rocprof_summary.py /tmp/kt_c3/kt_results.db > $out/config3_kernel_trace_stats.txt
python3 - <<PY
import json
for f in ("config2","config3","config3_3d","config4","configlut","config3_per_level_route","configlut_per_level_route","config3_3d_per_level_route","rays64"):
    try:
        d=json.loads([l for l in open("$out/%s.json"%f) if l.startswith("{")][-1]); print(f, d["value"], d["unit"], d["ms_per_step"])
    except Exception as e: print(f, "ERR", e)
PY
