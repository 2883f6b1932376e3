#!/bin/bash
# scripts/trace_by_grid.sh <tag> <workload> <steps>  -> gpurun_out/<tag>_<workload>_kernel_stats_by_grid.md
R=$GRAFT_REPO_ROOT; TAG=$1; W=$2; ST=${3:-300}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tg_$W
CMD="python bench.py --workload $W --steps $ST --warmup 20 --no-cpu-baseline --no-profile"
rocprofv3 --kernel-trace --output-format csv -d /tmp/tg_$W -- python $R/bench.py --workload $W --steps $ST --warmup 20 --no-cpu-baseline --no-profile > /tmp/tg_$W.out 2>&1
python $R/profiles/summarize_trace_by_grid.py /tmp/tg_$W "$CMD   (rocprofv3 --kernel-trace --output-format csv)" > $R/gpurun_out/${TAG}_${W}_kernel_stats_by_grid.md
head -12 $R/gpurun_out/${TAG}_${W}_kernel_stats_by_grid.md | cut -c1-230
