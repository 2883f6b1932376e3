#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -q -x -k "above_256 or K_above or is_and_alpha or ab_" 2>&1 | tail -12 | cut -c1-220
python bench.py --workload c2is512 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5_37_bench_c2is512.json 2> gpurun_out/r5_37_bench_c2is512.err
tail -1 gpurun_out/r5_37_bench_c2is512.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('c2is512', d['value'], d['ms_per_step'], d['roofline']['phases_ms_per_step'], d['config']['path'][:60])"
tail -2 gpurun_out/r5_37_bench_c2is512.err
bash scripts/prof_cmd.sh r5_37_c2is512 python $GRAFT_REPO_ROOT/bench.py --workload c2is512 --steps 10 --warmup 3 --no-cpu-baseline
cd $GRAFT_REPO_ROOT; head -14 gpurun_out/r5_37_c2is512_kernel_stats.md | cut -c1-170
