#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py -q -k gemm64 2>&1 | tail -2
python scripts/bench_gemm64.py 2>&1 | grep gemm64
python -m pytest tests/test_gpu_golden.py tests/test_gpu_pins.py tests/test_gpu_conditioning.py -x -q 2>&1 | tail -2
for w in c2 c4; do
python bench.py --workload $w --steps 2000 --warmup 100 --no-cpu-baseline > gpurun_out/r5_23_bench_${w}_steady.json 2>/dev/null
tail -1 gpurun_out/r5_23_bench_${w}_steady.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$w', d['value'], d['ms_per_step'], d['roofline']['phases_ms_per_step'], d['cost_first_last'])"
done
bash scripts/timeline.sh r5_23 c4 1; cd $GRAFT_REPO_ROOT; grep -E "gemm64|period" gpurun_out/r5_23_timeline_c4.txt | cut -c1-150
