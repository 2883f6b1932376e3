#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -q -k "cnmfsc" 2>&1 | tail -8 > gpurun_out/r5_06_cnmfsc.log
python bench.py --workload c4sc --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5_06_bench_c4sc.json 2> gpurun_out/r5_06_bench_c4sc.err
NMFX_SC_NO_F18=1 python bench.py --workload c4sc --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5_06_bench_c4sc_nof18.json 2> /dev/null
bash scripts/prof_cmd.sh r5_06_c4sc python bench.py --workload c4sc --steps 10 --warmup 5 --no-cpu-baseline
tail -4 gpurun_out/r5_06_cnmfsc.log
python - <<PY
import json
for f in ('r5_06_bench_c4sc','r5_06_bench_c4sc_nof18'):
    d=json.loads(open('gpurun_out/%s.json'%f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'), d.get('per_tag_ms') or d.get('tags') or '')
PY
head -16 gpurun_out/r5_06_c4sc_kernel_stats.md | cut -c1-160
