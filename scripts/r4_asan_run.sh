#!/bin/bash
# host-sanitizer campaign on the GPU box: scripts/r4_asan_run.sh <seconds> <seed> <kinds> <pool: on|off> <tag>
secs=${1:-300}; seed=${2:-12}; kinds=${3:-e}; pool=${4:-off}; tag=${5:-a}
mkdir -p gpurun_out
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=0:print_stacktrace=1:symbolize=1
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export ASAN_SYMBOLIZER_PATH=/opt/rocm/lib/llvm/bin/llvm-symbolizer
[ "$pool" = off ] && export NMFX_NO_POOL=1
timeout $((secs + 120)) ./tests/host_asan/fuzz_multi "$secs" "$seed" "$kinds" > gpurun_out/r4_asan_${tag}.log 2>&1
echo "exit $?" >> gpurun_out/r4_asan_${tag}.log
tail -5 gpurun_out/r4_asan_${tag}.log
