#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_sharded.py -q -x -k "cnmf or is_and_alpha or ab_" 2>&1 | tail -12 | cut -c1-220
for w in c4is c4 c4kl; do
python bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r5_35_bench_$w.json 2>/dev/null
tail -1 gpurun_out/r5_35_bench_$w.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$w', d['value'], d['ms_per_step'], d['roofline']['phases_ms_per_step'])"
done
bash scripts/prof_cmd.sh r5_35_c4is python $GRAFT_REPO_ROOT/bench.py --workload c4is --steps 20 --warmup 5 --no-cpu-baseline
cd $GRAFT_REPO_ROOT; head -14 gpurun_out/r5_35_c4is_kernel_stats.md | cut -c1-170
