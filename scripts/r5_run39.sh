#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_abi_and_host.py -q -k "above_256 or stamp" 2>&1 | tail -4 | cut -c1-200
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5_39_bench_c3.json 2>/dev/null
for w in c4is c2is512; do python bench.py --workload $w --steps 20 --warmup 5 > gpurun_out/r5_39_bench_$w.json 2>/dev/null; done
for w in c3 c4is c2is512; do tail -1 gpurun_out/r5_39_bench_$w.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$w', d['value'], d['ms_per_step'], r['frac'], r.get('traffic'), (d.get('cpu_baseline') or {}).get('value'))"; done
