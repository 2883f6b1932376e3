#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
t0=$(date +%s)
python -m pytest tests/ -x -q -m gpu --durations=8 > gpurun_out/r5_25_gputests.log 2>&1
echo "rc $? wall $(( $(date +%s) - t0 )) s" >> gpurun_out/r5_25_gputests.log
cp gpurun_out/parity_errors.json gpurun_out/r5_25_parity_errors.json 2>/dev/null
tail -14 gpurun_out/r5_25_gputests.log | cut -c1-160
for w in c3 c2 c4; do
  python bench.py --workload $w --steps 20 --warmup 5 > gpurun_out/r5_25_bench_$w.json 2> /dev/null
  tail -1 gpurun_out/r5_25_bench_$w.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$w', d['value'], d['ms_per_step'], r['frac'], r.get('traffic'), r['phases_ms_per_step'])"
done
python -c "
import __graft_entry__ as g; g.smoke(); print('smoke ok')"
