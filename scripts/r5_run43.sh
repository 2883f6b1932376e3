#!/bin/bash
# the tree after the last change to engine.hip (IS / alpha-beta with K > 256 in column blocks): whole suite, PMC passes, the driver's line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
t0=$(date +%s)
python -m pytest tests/ -x -q -m gpu --durations=6 > gpurun_out/r5_43_gputests.log 2>&1
echo "rc $? wall $(( $(date +%s) - t0 )) s" >> gpurun_out/r5_43_gputests.log
cp gpurun_out/parity_errors.json gpurun_out/r5_43_parity_errors.json 2>/dev/null
tail -10 gpurun_out/r5_43_gputests.log | cut -c1-160
for w in c3 c2 c4 c4kl c5 c4sc c2is256 c4is c2is512; do bash scripts/pmc_passes.sh $w r5_43; done
cd $GRAFT_REPO_ROOT
for w in c3 c2 c4 c4kl c5 c4sc c2is256 c4is c2is512; do echo "== $w"; grep -E "^## |HBM traffic" gpurun_out/r5_43_${w}_pmc.md | grep -A1 "fused_kernel<\|gemm_pipe_kernel<128, 128, true, true" | grep -v "^--" | grep "HBM\|##" | cut -c1-200; done
