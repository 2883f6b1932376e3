#!/bin/bash
# rocprofv3 --kernel-trace --stats of an arbitrary command, summarised: scripts/prof_cmd.sh <tag> <command ...>   -> gpurun_out/<tag>_kernel_stats.md
R=$GRAFT_REPO_ROOT; TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pc_$TAG
rocprofv3 --kernel-trace --stats -d /tmp/pc_$TAG -o ks -- "$@" > /tmp/pc_$TAG.out 2>&1
python $R/profiles/summarize_rocprof.py /tmp/pc_$TAG/ks_results.db "$*" > $R/gpurun_out/${TAG}_kernel_stats.md
tail -3 /tmp/pc_$TAG.out
