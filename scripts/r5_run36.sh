#!/bin/bash
# after the last change to engine.hip (IS / alpha-beta cnmf on the fused passes): the whole suite, the PMC passes again on the final sources, c4is lines
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
t0=$(date +%s)
python -m pytest tests/ -x -q -m gpu --durations=6 > gpurun_out/r5_36_gputests.log 2>&1
echo "rc $? wall $(( $(date +%s) - t0 )) s" >> gpurun_out/r5_36_gputests.log
cp gpurun_out/parity_errors.json gpurun_out/r5_36_parity_errors.json 2>/dev/null
tail -10 gpurun_out/r5_36_gputests.log | cut -c1-160
for w in c3 c2 c4 c4kl c5 c4sc c2is256 c4is; do bash scripts/pmc_passes.sh $w r5_36; done
cd $GRAFT_REPO_ROOT
python bench.py --workload c4is --steps 20 --warmup 5 > gpurun_out/r5_36_bench_c4is.json 2>/dev/null
tail -1 gpurun_out/r5_36_bench_c4is.json | cut -c1-300
for w in c3 c2 c4 c4kl c5 c4sc c2is256 c4is; do echo "== $w"; grep -E "^## |HBM traffic" gpurun_out/r5_36_${w}_pmc.md | grep -A1 "fused_kernel<\|gemm_pipe_kernel<128, 128, true, true" | grep -v "^--" | cut -c1-210; done
