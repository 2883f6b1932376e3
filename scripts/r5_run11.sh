#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for w in c2 c4 c3; do
python bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r5_11_bench_$w.json 2> gpurun_out/r5_11_bench_$w.err
NMFX_NO_SIDE_STREAM=1 python bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r5_11_bench_${w}_noside.json 2> /dev/null
python - <<PY
import json
for f in ('r5_11_bench_$w', 'r5_11_bench_${w}_noside'):
    try:
        d=json.loads(open('gpurun_out/%s.json' % f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d['roofline'].get('phases_ms_per_step'))
    except Exception as ex: print(f, 'failed', ex)
PY
tail -2 gpurun_out/r5_11_bench_$w.err
done
bash scripts/timeline.sh r5_11 c4 1
cd $GRAFT_REPO_ROOT
cat gpurun_out/r5_11_timeline_c4.txt | cut -c1-150
t0=$(date +%s)
python -m pytest tests/ -x -q -m gpu --durations=12 > gpurun_out/r5_11_gputests.log 2>&1
echo "rc $? wall $(( $(date +%s) - t0 )) s" >> gpurun_out/r5_11_gputests.log
cp gpurun_out/parity_errors.json gpurun_out/r5_11_parity_errors.json 2>/dev/null
tail -22 gpurun_out/r5_11_gputests.log | cut -c1-180
