#!/bin/bash
# bench lines of the final tree (after the PMC file was re-stamped): the driver's default command first, then every workload with its CPU baseline
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python bench.py > gpurun_out/r5_21_bench_default.json 2> gpurun_out/r5_21_bench_default.err
python bench.py --steps 20 --warmup 5 > gpurun_out/r5_21_bench_c3_steps20.json 2>/dev/null
for w in c2 c4 c4kl c5 c2is c2is256 c4sc; do
  python bench.py --workload $w --steps 20 --warmup 5 > gpurun_out/r5_21_bench_$w.json 2> gpurun_out/r5_21_bench_$w.err
done
for f in default c3_steps20 c2 c4 c4kl c5 c2is c2is256 c4sc; do tail -1 gpurun_out/r5_21_bench_$f.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$f', d['value'], d['ms_per_step'], r['kernel'][:40], r['frac'], r.get('traffic'), (d.get('cpu_baseline') or {}).get('value'))"; done
