#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_gpu_conditioning.py tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q 2>&1 | tail -4
python scripts/fuzz_campaign_fixed_factor.py 291 900 > gpurun_out/r5_18_fuzz_fixed_factor_a.log 2>&1 &
python scripts/fuzz_campaign_fixed_factor.py 292 900 > gpurun_out/r5_18_fuzz_fixed_factor_b.log 2>&1 &
wait
tail -n 1 gpurun_out/r5_18_fuzz_fixed_factor_a.log gpurun_out/r5_18_fuzz_fixed_factor_b.log | cut -c1-500
grep BAD gpurun_out/r5_18_fuzz_fixed_factor_*.log | cut -c1-250
