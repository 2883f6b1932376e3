#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
t0=$(date +%s)
python -m pytest tests/ -x -q -m gpu --durations=10 > gpurun_out/r5_17_gputests.log 2>&1
echo "rc $? wall $(( $(date +%s) - t0 )) s" >> gpurun_out/r5_17_gputests.log
cp gpurun_out/parity_errors.json gpurun_out/r5_17_parity_errors.json 2>/dev/null
tail -16 gpurun_out/r5_17_gputests.log | cut -c1-180
python scripts/fuzz_campaign_fixed_factor.py 191 420 > gpurun_out/r5_17_fuzz_fixed_factor.log 2>&1 &
python scripts/fuzz_campaign_r3.py 193 420 > gpurun_out/r5_17_fuzz_r3.log 2>&1 &
python scripts/fuzz_campaign_sc.py 192 420 > gpurun_out/r5_17_fuzz_sc.log 2>&1 &
NMFX_FUZZ_PATH=2 python scripts/fuzz_campaign_sc.py 194 420 > gpurun_out/r5_17_fuzz_sc_fused.log 2>&1 &
wait
tail -n 2 gpurun_out/r5_17_fuzz_fixed_factor.log gpurun_out/r5_17_fuzz_r3.log gpurun_out/r5_17_fuzz_sc.log gpurun_out/r5_17_fuzz_sc_fused.log | cut -c1-400
grep -c BAD gpurun_out/r5_17_fuzz_*.log
