#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
t0=$(date +%s)
python -m pytest tests/ -x -q -m gpu --durations=20 > gpurun_out/r5_10_gputests.log 2>&1
echo "rc $? wall $(( $(date +%s) - t0 )) s" >> gpurun_out/r5_10_gputests.log
cp gpurun_out/parity_errors.json gpurun_out/r5_10_parity_errors.json 2>/dev/null
tail -26 gpurun_out/r5_10_gputests.log | cut -c1-180
for w in c4is c2is512; do
python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r5_10_bench_$w.json 2> gpurun_out/r5_10_bench_$w.err
python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r5_10_bench_$w.json').read().strip().splitlines()[-1]); print('$w', d['value'], d['ms_per_step'], (d.get('roofline') or {}).get('phases_ms_per_iteration') or d.get('roofline'))
except Exception as ex: print('$w', 'failed', ex)
PY
tail -2 gpurun_out/r5_10_bench_$w.err
done
