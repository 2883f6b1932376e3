#!/bin/bash
# host-sanitizer campaign on the GPU box: scripts/r4_asan_run.sh <seconds> <seed> <kinds> <pool: on|off> <tag> [binary]
# a run that stops making progress is dumped with rocgdb (all threads) before it is killed
secs=${1:-300}; seed=${2:-12}; kinds=${3:-e}; pool=${4:-off}; tag=${5:-a}; bin=${6:-./tests/host_asan/fuzz_multi}
mkdir -p gpurun_out
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=0:print_stacktrace=1:symbolize=1
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export ASAN_SYMBOLIZER_PATH=/opt/rocm/lib/llvm/bin/llvm-symbolizer
[ "$pool" = off ] && export NMFX_NO_POOL=1
log=gpurun_out/r4_asan_${tag}.log
FUZZ_NO_WATCHDOG=1 $bin "$secs" "$seed" "$kinds" > $log 2>&1 &
pid=$!
t0=$(date +%s); last_size=-1; quiet=0
while kill -0 $pid 2>/dev/null; do
  sleep 10
  size=$(stat -c %s $log)
  if [ "$size" = "$last_size" ]; then quiet=$((quiet + 10)); else quiet=0; last_size=$size; fi
  now=$(date +%s)
  if [ $quiet -ge 150 ] || [ $((now - t0)) -ge $((secs + 150)) ]; then
    echo "STALL: no output for ${quiet}s (elapsed $((now - t0))s): rocgdb backtraces of all threads" >> $log
    timeout 120 /opt/rocm/bin/rocgdb -p $pid -batch -ex "set pagination off" -ex "thread apply all bt 25" >> $log 2>&1
    kill -9 $pid
    break
  fi
done
wait $pid 2>/dev/null
echo "exit $?" >> $log
grep -v "^\.\.\." $log | tail -4
