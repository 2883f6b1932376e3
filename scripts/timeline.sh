#!/bin/bash
# one iteration of a workload as a kernel timeline: scripts/timeline.sh <tag> <workload> [launches of the heaviest kernel per iteration]  -> gpurun_out/<tag>_timeline_<workload>.txt
R=$GRAFT_REPO_ROOT; TAG=$1; W=$2; PER=${3:-1}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tl_$W
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$W -- python $R/bench.py --workload $W --steps 10 --warmup 3 --no-cpu-baseline --no-profile > /tmp/tl_$W.out 2>&1
python $R/scripts/kernel_timeline.py /tmp/tl_$W $PER > $R/gpurun_out/${TAG}_timeline_$W.txt
tail -1 /tmp/tl_$W.out | cut -c1-200
