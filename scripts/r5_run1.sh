#!/bin/bash
# round 5, first GPU contact: gemm64 + the conditioning classes on the new build, the same tests on the round-4 build (_r4ref), bench lines
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py -q -k gemm64 -x 2>&1 | tail -5 > gpurun_out/r5_01_gemm64.log
python -m pytest tests/test_gpu_conditioning.py -q 2>&1 | tail -40 > gpurun_out/r5_01_conditioning_new.log
cp gpurun_out/parity_errors.json gpurun_out/r5_01_conditioning_new_errors.json 2>/dev/null
(cd _r4ref && python -m pytest tests/test_gpu_conditioning.py -q 2>&1 | tail -60 > ../gpurun_out/r5_01_conditioning_r4build.log)
for w in c3 c2 c4; do python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r5_01_bench_$w.json 2> gpurun_out/r5_01_bench_$w.err; done
python -m pytest tests/test_gpu_golden.py tests/test_gpu_pins.py -q -x 2>&1 | tail -5 > gpurun_out/r5_01_golden_pins.log
cat gpurun_out/r5_01_gemm64.log gpurun_out/r5_01_conditioning_new.log; tail -5 gpurun_out/r5_01_conditioning_r4build.log; cat gpurun_out/r5_01_golden_pins.log
for w in c3 c2 c4; do python -c "import json,sys; d=json.loads(open('gpurun_out/r5_01_bench_$w.json').read().strip().splitlines()[-1]); print('$w', d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'))"; done
