#!/bin/bash
cd $GRAFT_REPO_ROOT
for m in 2 5; do echo "mode $m"; for r in 1 2; do NMFX_G64_MODE=$m python scripts/bench_gemm64.py 2>&1 | grep gemm64 | head -4; done; done
