#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5_44_bench_c3.json 2>/dev/null
tail -1 gpurun_out/r5_44_bench_c3.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r.get('traffic'), r['traffic_source'][-60:], d['cpu_baseline']['value'])"
