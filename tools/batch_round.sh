set -u
mkdir -p gpurun_out/b1
python tools/stress_modes.py --large 0 40 > gpurun_out/b1/stress_large.txt 2>&1; echo "stress large rc=$?"; tail -2 gpurun_out/b1/stress_large.txt
python tools/stress_modes.py --far1 0 100 > gpurun_out/b1/stress_far1.txt 2>&1; echo "stress far1 rc=$?"; tail -1 gpurun_out/b1/stress_far1.txt
bash tools/shard_times.sh > gpurun_out/b1/shards.txt 2>&1; cat gpurun_out/b1/shards.txt
for c in 2 3 4; do timeout -k 10 280 python bench.py --config $c > gpurun_out/b1/config$c.json 2> gpurun_out/b1/config$c.err; echo "config $c rc=$?"; python -c "
import json; d=json.loads(open('gpurun_out/b1/config$c.json').read().strip().splitlines()[-1]); print(d['metric'], d['value'], d['unit'], d['ms_per_step'])"; done
python bench.py --rays 64 --cpu-seconds 0 > gpurun_out/b1/rays64.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/b1/rays64.json').read().strip().splitlines()[-1]); print('rays64', d['value'], d['ms_per_step'])"
