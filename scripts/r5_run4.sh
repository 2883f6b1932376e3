#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_gpu_conditioning.py -q 2>&1 | tail -25 > gpurun_out/r5_04_conditioning.log
python -m pytest tests/test_gpu_sharded.py -q -k "rccl or stop_rule_equals or blocking_api_n_gpus" 2>&1 | tail -25 > gpurun_out/r5_04_rccl.log
python -m pytest tests/test_gpu_parity.py -q -k "nmfsc" 2>&1 | tail -15 > gpurun_out/r5_04_nmfsc.log
python -m pytest tests/test_gpu_golden.py tests/test_gpu_pins.py tests/test_mex_gateway.py -q 2>&1 | tail -8 > gpurun_out/r5_04_golden.log
tail -12 gpurun_out/r5_04_conditioning.log; tail -25 gpurun_out/r5_04_rccl.log; tail -8 gpurun_out/r5_04_nmfsc.log; tail -8 gpurun_out/r5_04_golden.log
