#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python scripts/fuzz_campaign_dual.py 701 300 > gpurun_out/r5_40_fuzz_dual_a.log 2>&1 &
python scripts/fuzz_campaign_dual.py 702 300 > gpurun_out/r5_40_fuzz_dual_b.log 2>&1 &
wait
tail -n 1 gpurun_out/r5_40_fuzz_dual_a.log gpurun_out/r5_40_fuzz_dual_b.log | cut -c1-400; grep -h BAD gpurun_out/r5_40_fuzz_dual_*.log | cut -c1-300 | head
