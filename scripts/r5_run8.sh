#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -q -k "cnmfsc" 2>&1 | tail -15 > gpurun_out/r5_08_cnmfsc.log
python bench.py --workload c4sc --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5_08_bench_c4sc.json 2> gpurun_out/r5_08_bench_c4sc.err
bash scripts/prof_cmd.sh r5_08_c4sc python $GRAFT_REPO_ROOT/bench.py --workload c4sc --steps 10 --warmup 5 --no-cpu-baseline
cd $GRAFT_REPO_ROOT
tail -6 gpurun_out/r5_08_cnmfsc.log
tail -3 gpurun_out/r5_08_bench_c4sc.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r5_08_bench_c4sc.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['line_search_tries_H'], d['cost_first_last'], d['roofline']['tag_frac_of_peak'], d['roofline']['phases_ms_per_iteration_whole_call'])
PY
head -16 gpurun_out/r5_08_c4sc_kernel_stats.md | cut -c1-160
