#!/bin/bash
# usage: pmc_run.sh <workload> <tag>   -- three separate --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ group), summaries into gpurun_out/
set -e
W=$1; TAG=$2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python $R/bench.py --workload $W --steps 5 --warmup 1 --no-cpu-baseline --no-profile --spinup-ms 0"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_${W}_f -o f -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmc_${W}_w -o w -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT -d /tmp/pmc_${W}_s -o s -- $CMD > /dev/null 2>&1
python $R/profiles/summarize_pmc.py "rocprofv3 --kernel-trace --pmc <group> -- python bench.py --workload $W --steps 5 --warmup 1 --no-cpu-baseline --no-profile --spinup-ms 0; groups (separate passes): FETCH_SIZE | WRITE_SIZE | SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT" /tmp/pmc_${W}_f/f_results.db /tmp/pmc_${W}_w/w_results.db /tmp/pmc_${W}_s/s_results.db > $R/gpurun_out/${TAG}_${W}_pmc.md
