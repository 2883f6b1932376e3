#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
t0=$(date +%s)
python -m pytest tests/ -x -q -m gpu --durations=15 > gpurun_out/r5_13_gputests.log 2>&1
echo "rc $? wall $(( $(date +%s) - t0 )) s" >> gpurun_out/r5_13_gputests.log
cp gpurun_out/parity_errors.json gpurun_out/r5_13_parity_errors.json 2>/dev/null
tail -24 gpurun_out/r5_13_gputests.log | cut -c1-180
for w in c2 c4 c3; do
python bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r5_13_bench_$w.json 2> gpurun_out/r5_13_bench_$w.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r5_13_bench_$w.json').read().strip().splitlines()[-1]); print('$w', d['value'], d['ms_per_step'], d['roofline'].get('phases_ms_per_step'))
PY
done
