#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_gpu_golden.py tests/test_gpu_pins.py -x -q 2>&1 | tail -2
for w in c3 c2 c4; do
python bench.py --workload $w --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r5_30_bench_$w.json 2>/dev/null
tail -1 gpurun_out/r5_30_bench_$w.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$w', d['value'], d['ms_per_step'], d['roofline']['phases_ms_per_step']['small kernels + gaps (remainder)'], d['cost_first_last'])"
done
bash scripts/prof_cmd.sh r5_30_c3 python $GRAFT_REPO_ROOT/bench.py --workload c3 --steps 20 --warmup 5 --no-cpu-baseline
cd $GRAFT_REPO_ROOT; grep -E "w_update" gpurun_out/r5_30_c3_kernel_stats.md | cut -c1-160
bash scripts/prof_cmd.sh r5_30_c2 python $GRAFT_REPO_ROOT/bench.py --workload c2 --steps 200 --warmup 20 --no-cpu-baseline
cd $GRAFT_REPO_ROOT; grep -E "w_update|h_update|gemm64|w_normalize" gpurun_out/r5_30_c2_kernel_stats.md | cut -c1-160
