#!/bin/bash
# closing campaigns on the final build of round 5, four processes sharing the GPU
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python scripts/fuzz_campaign.py 591 480 > gpurun_out/r5_29_fuzz_r2.log 2>&1 &
python scripts/fuzz_campaign_fixed_factor.py 592 480 > gpurun_out/r5_29_fuzz_fixed_factor.log 2>&1 &
python scripts/fuzz_campaign_r3.py 593 480 > gpurun_out/r5_29_fuzz_r3.log 2>&1 &
NMFX_FUZZ_PATH=2 python scripts/fuzz_campaign_sc.py 594 480 > gpurun_out/r5_29_fuzz_sc_fused.log 2>&1 &
wait
python scripts/fuzz_campaign_sc.py 595 240 > gpurun_out/r5_29_fuzz_sc.log 2>&1
tail -n 1 gpurun_out/r5_29_fuzz_*.log | cut -c1-500
grep -h BAD gpurun_out/r5_29_fuzz_*.log | cut -c1-260 | head
