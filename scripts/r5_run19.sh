#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for i in 1 2; do
python bench.py --workload c3 --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r5_19_bench_c3_$i.json 2> gpurun_out/r5_19_bench_c3.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r5_19_bench_c3_$i.json').read().strip().splitlines()[-1]); print('c3', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('phases_ms_per_step'))
PY
done
python bench.py --workload c4kl --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r5_19_bench_c4kl.json 2>/dev/null
python - <<PY
import json
d=json.loads(open('gpurun_out/r5_19_bench_c4kl.json').read().strip().splitlines()[-1]); print('c4kl', d['value'], d['ms_per_step'], d['roofline'].get('phases_ms_per_step'))
PY
python -m pytest tests/test_gpu_golden.py tests/test_gpu_pins.py tests/test_gpu_fullsize_oracle.py tests/test_gpu_parity.py -x -q -k "kl or KL or lnmf or golden or pins or fixed_points" 2>&1 | tail -3
python scripts/fuzz_campaign_fixed_factor.py 391 300 kl > gpurun_out/r5_19_fuzz_kl.log 2>&1
tail -n 1 gpurun_out/r5_19_fuzz_kl.log | cut -c1-400
python - <<PY
import json
d=json.load(open('gpurun_out/parity_errors.json')); print({k:v for k,v in d['worst'].items() if k in ('W','H','WH','cost')})
PY
