#!/bin/bash
# bench lines (with cpu_baseline) + rocprofv3 --kernel-trace --stats summaries of the same commands + the N = 2 self-launch line:  scripts/final_evidence.sh <tag>
R=$GRAFT_REPO_ROOT; TAG=$1
cd $R
for w in c3 c2 c4 c4kl c5 c2is c2is256 c4sc; do
  python bench.py --workload $w --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_$w.json 2> gpurun_out/${TAG}_bench_$w.err
done
NMFX_BENCH_BACKEND=gloo NMFX_BENCH_ONE_DEVICE=1 python bench.py --gpus 2 --steps 10 --warmup 3 --workload c3_shard2 --no-cpu-baseline > gpurun_out/${TAG}_bench_selflaunch_2ranks_gloo.json 2> gpurun_out/${TAG}_bench_selflaunch.err
cd /tmp && export TMPDIR=/tmp
for w in c3 c2 c4 c4kl c5 c4sc c2is256; do
  rm -rf /tmp/ks_$w
  rocprofv3 --kernel-trace --stats -d /tmp/ks_$w -o ks -- python $R/bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
  python $R/profiles/summarize_rocprof.py /tmp/ks_$w/ks_results.db "python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline   (rocprofv3 --kernel-trace --stats; 5 warm-up + 20 timed iterations, averages include the warm-up launches)" > $R/gpurun_out/${TAG}_${w}_kernel_stats.md
done
cd $R
for w in c3 c2 c4 c4kl c5 c2is c2is256 c4sc; do tail -1 gpurun_out/${TAG}_bench_$w.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['config']['name'], d['value'], d['ms_per_step'], r['kernel'][:40], r['frac'], r['avg_launch_ms'], (d.get('cpu_baseline') or {}).get('value'))"; done
tail -1 gpurun_out/${TAG}_bench_selflaunch_2ranks_gloo.json | cut -c1-400
