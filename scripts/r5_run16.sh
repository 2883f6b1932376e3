#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python scripts/bench_gemm64.py 2>&1 | grep gemm64
python -m pytest tests/test_gpu_kernels.py -q -k gemm64 2>&1 | tail -2
