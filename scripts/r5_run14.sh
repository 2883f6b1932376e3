#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/fullsize_golden
python tests/golden/make_fullsize_golden.py gpurun_out/fullsize_golden > gpurun_out/r5_14_make_fullsize_golden.log 2>&1
cat gpurun_out/r5_14_make_fullsize_golden.log | tail -10
cp gpurun_out/fullsize_golden/*.npz tests/golden/
ls -la tests/golden/fullsize_*
t0=$(date +%s)
python -m pytest tests/ -x -q -m gpu --durations=25 > gpurun_out/r5_14_gputests.log 2>&1
echo "rc $? wall $(( $(date +%s) - t0 )) s" >> gpurun_out/r5_14_gputests.log
cp gpurun_out/parity_errors.json gpurun_out/r5_14_parity_errors.json 2>/dev/null
tail -34 gpurun_out/r5_14_gputests.log | cut -c1-180
