#!/bin/bash
# the very last build of round 5: five campaigns side by side
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python scripts/fuzz_campaign.py 801 540 > gpurun_out/r5_45_fuzz_r2.log 2>&1 &
python scripts/fuzz_campaign_fixed_factor.py 802 540 > gpurun_out/r5_45_fuzz_fixed_factor.log 2>&1 &
python scripts/fuzz_campaign_r3.py 803 540 > gpurun_out/r5_45_fuzz_r3.log 2>&1 &
NMFX_FUZZ_PATH=2 python scripts/fuzz_campaign_sc.py 804 540 > gpurun_out/r5_45_fuzz_sc_fused.log 2>&1 &
python scripts/fuzz_campaign_dual.py 805 540 > gpurun_out/r5_45_fuzz_dual.log 2>&1 &
wait
tail -n 1 gpurun_out/r5_45_fuzz_*.log | cut -c1-520
grep -h "BAD\|TRIES" gpurun_out/r5_45_fuzz_*.log | cut -c1-300 | head
