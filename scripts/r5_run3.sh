#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_gpu_conditioning.py -q 2>&1 | tail -40 > gpurun_out/r5_03_conditioning.log
cp gpurun_out/parity_errors.json gpurun_out/r5_03_conditioning_errors.json 2>/dev/null
python -m pytest tests -m gpu -q --durations=30 2>&1 | tail -80 > gpurun_out/r5_03_gputests.log
cp gpurun_out/parity_errors.json gpurun_out/r5_03_parity_errors.json 2>/dev/null
for w in c3 c2 c4; do python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r5_03_bench_$w.json 2> gpurun_out/r5_03_bench_$w.err; done
tail -32 gpurun_out/r5_03_conditioning.log; tail -45 gpurun_out/r5_03_gputests.log
for w in c3 c2 c4; do python -c "import json,sys; d=json.loads(open('gpurun_out/r5_03_bench_$w.json').read().strip().splitlines()[-1]); print('$w', d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'))"; done
