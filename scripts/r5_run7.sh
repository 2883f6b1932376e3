#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash scripts/timeline.sh r5_07 c2 2
bash scripts/timeline.sh r5_07 c4 1
bash scripts/prof_cmd.sh r5_07_c4sc python $GRAFT_REPO_ROOT/bench.py --workload c4sc --steps 10 --warmup 5 --no-cpu-baseline
cd $GRAFT_REPO_ROOT
python bench.py --workload c4sc --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5_07_bench_c4sc.json 2>/dev/null
python bench.py --workload c3 --api blocking --gpus 1 --backends rccl,peer --steps 10 > gpurun_out/r5_07_blocking_c3_backends.json 2>/dev/null
cat gpurun_out/r5_07_timeline_c2.txt gpurun_out/r5_07_timeline_c4.txt | cut -c1-150
head -14 gpurun_out/r5_07_c4sc_kernel_stats.md | cut -c1-150
python -c "
import json
d=json.loads(open('gpurun_out/r5_07_bench_c4sc.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['tag_frac_of_peak'], d['roofline']['phases_ms_per_iteration_whole_call'])
for l in open('gpurun_out/r5_07_blocking_c3_backends.json'):
    d=json.loads(l); print({k:d.get(k) for k in ('backend','backend_used','allreduce_ms_per_step','allreduces_timed','ms_per_iteration_inside','ingest_s','error')})
"
