#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python scripts/bench_gemm64.py > gpurun_out/r5_15_gemm64_direct.txt 2>&1
NMFX_GEMM64_LDS=1 python scripts/bench_gemm64.py > gpurun_out/r5_15_gemm64_lds.txt 2>&1
cat gpurun_out/r5_15_gemm64_direct.txt gpurun_out/r5_15_gemm64_lds.txt
python -m pytest tests/test_gpu_kernels.py -q -k gemm64 2>&1 | tail -2
