#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for w in c2 c4; do
python bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r5_12_bench_$w.json 2> gpurun_out/r5_12_bench_$w.err
python - <<PY
import json
for f in ('r5_12_bench_$w',):
    try:
        d=json.loads(open('gpurun_out/%s.json' % f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d['roofline'].get('phases_ms_per_step'))
    except Exception as ex: print(f, 'failed', ex)
PY
done
bash scripts/timeline.sh r5_12 c4 1
cd $GRAFT_REPO_ROOT
cat gpurun_out/r5_12_timeline_c4.txt | cut -c1-150
