#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
t0=$(date +%s)
python -m pytest tests/ -x -q -m gpu --durations=20 > gpurun_out/r5_09_gputests.log 2>&1
echo "rc $? wall $(( $(date +%s) - t0 )) s" >> gpurun_out/r5_09_gputests.log
cp gpurun_out/parity_errors.json gpurun_out/r5_09_parity_errors.json 2>/dev/null
tail -32 gpurun_out/r5_09_gputests.log | cut -c1-180
