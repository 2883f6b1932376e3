#!/bin/bash
# one steady-state iteration of a workload as a kernel timeline (gaps included): scripts/timeline.sh c2 [launches of the heaviest kernel per iteration]
R=$GRAFT_REPO_ROOT; W=$1; PER=${2:-1}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tl_$W
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$W -- python $R/bench.py --workload $W --steps 10 --warmup 3 --no-cpu-baseline --no-profile > /dev/null 2>&1
python $R/scripts/kernel_timeline.py /tmp/tl_$W $PER
