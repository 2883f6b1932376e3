#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
t0=$(date +%s)
python -m pytest tests/ -x -q -m gpu --durations=8 > gpurun_out/r5_31_gputests.log 2>&1
echo "rc $? wall $(( $(date +%s) - t0 )) s" >> gpurun_out/r5_31_gputests.log
cp gpurun_out/parity_errors.json gpurun_out/r5_31_parity_errors.json 2>/dev/null
tail -14 gpurun_out/r5_31_gputests.log | cut -c1-160
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5_31_bench_c3.json 2>/dev/null
tail -1 gpurun_out/r5_31_bench_c3.json | cut -c1-1500
rocprofv3 --version > /dev/null 2>&1
bash scripts/prof_cmd.sh r5_31_c3 python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline
cd $GRAFT_REPO_ROOT; head -12 gpurun_out/r5_31_c3_kernel_stats.md | cut -c1-160
