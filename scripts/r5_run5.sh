#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_gpu_sharded.py -q -k "rccl" 2>&1 | tail -15 > gpurun_out/r5_05_rccl.log
python scripts/fuzz_campaign_fixed_factor.py 91 420 > gpurun_out/r5_05_fuzz_fixed_factor.log 2>&1
python scripts/fuzz_campaign_sc.py 92 240 > gpurun_out/r5_05_fuzz_sc_default.log 2>&1
python scripts/fuzz_campaign_r3.py 93 240 > gpurun_out/r5_05_fuzz_r3.log 2>&1
tail -6 gpurun_out/r5_05_rccl.log; tail -12 gpurun_out/r5_05_fuzz_fixed_factor.log | cut -c1-400; tail -5 gpurun_out/r5_05_fuzz_sc_default.log | cut -c1-400; tail -5 gpurun_out/r5_05_fuzz_r3.log | cut -c1-400
