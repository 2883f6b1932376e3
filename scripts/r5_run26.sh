#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_pins.py -q -k "cnmfsc" 2>&1 | tail -2
python bench.py --workload c4sc --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r5_26_bench_c4sc.json 2>/dev/null
tail -1 gpurun_out/r5_26_bench_c4sc.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['phases_ms_per_iteration_whole_call'], d['cost_first_last'])"
bash scripts/prof_cmd.sh r5_26_c4sc python $GRAFT_REPO_ROOT/bench.py --workload c4sc --steps 20 --warmup 5 --no-cpu-baseline
cd $GRAFT_REPO_ROOT; grep -E "w_slices" gpurun_out/r5_26_c4sc_kernel_stats.md | cut -c1-160
