#!/bin/bash
# Round-5 evidence batch (same passes as round 4) (GPU box): the rocprofv3 passes + bench line of the headline (tools/profile.sh), every 1/8 shard
# and the 1/4, 1/2 splits timed alone, the --config lines (2, 3, 3 --3d, 4, lut), kernel traces of configs[2] / [3] and the
# counters of the configs[3] radiance / Jacobian / combine kernels -- FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes
# (they cannot share one: 3 + 2 TCC slots; round 3's last pass asked for both and produced no database).
set -u
tag=${1:-r05_v2}
out=gpurun_out/$tag
mkdir -p $out
bash tools/profile.sh $tag > $out/profile.log 2>&1; tail -3 $out/profile.log
{
for s in 0/8 1/8 2/8 3/8 4/8 5/8 6/8 7/8 0/4 1/4 2/4 3/4 0/2 1/2; do timeout -k 10 120 python bench.py --shard $s --cpu-seconds 0 --steps 100 --warmup 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('shard %s: %.3f ms/step  op %.3f  serial: prep %.3f ff %.3f wings %.3f zones %.3f' % (d['config']['sharding'][11:14], d['ms_per_step'], r['coefficient_op_ms_in_timed_steps'], r['sr_prep_kernel_ms'], *[v['ms'] for v in r['kernels'].values()]))"; done
timeout -k 10 200 python tools/balanced_shards.py
timeout -k 10 100 python tools/host_overhead.py
} > $out/shards.txt 2>&1; cat $out/shards.txt
for c in 2 3 4 lut; do timeout -k 10 400 python bench.py --config $c --cpu-seconds 8 > $out/config$c.json 2> $out/config$c.err; echo "config $c rc=$?"; done
timeout -k 10 400 python bench.py --config 3 --3d --cpu-seconds 8 > $out/config3_3d.json 2> $out/config3_3d.err; echo "config 3 3d rc=$?"
timeout -k 10 200 python bench.py --rays 64 --cpu-seconds 0 > $out/rays64.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for cfg in 2 3; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_c$cfg -o kt -- python3 bench.py --config $cfg --steps 4 --warmup 1 --cpu-seconds 0 > /dev/null 2>&1
  python3 tools/rocprof_summary.py /tmp/kt_c$cfg/kt_results.db > $out/config${cfg}_kernel_trace_stats.txt
done
: > $out/config3_pmc_limb_kernels.txt
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_THREAD_CYCLES_VALU"; do
  p=$(echo $pass | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --pmc $pass -d /tmp/lp_$p -o p -- python3 bench.py --config 3 --steps 4 --warmup 1 --cpu-seconds 0 > $out/pmc_c3_$p.log 2>&1
  echo "config 3 pmc $p exit=$?"
  python3 tools/rocprof_summary.py /tmp/lp_$p/p_results.db 2>/dev/null | grep -i "adjoint\|sr_limb\|los_col\|adj_pack\|fold_pack\|glevel\|^kernel " >> $out/config3_pmc_limb_kernels.txt
done
head -12 $out/config3_kernel_trace_stats.txt
