#!/bin/bash
# ONE parametrised runner for everything that goes to the GPU box (replaces the 45 one-off scripts/r5_run*.sh of round 5):
#
#   gpurun --timeout N -- 'bash scripts/gpu_job.sh <tag> <step> [<step> ...]'        outputs: gpurun_out/<tag>_*
#
# steps (run in the order given; a variant is an A/B build of the kernel switches, nmf_toolbox_amd/build.py --variant):
#   ubench[:tiles]                  scripts/ubench_mfma (the fused kernel's MFMA / LDS / VALU skeleton)            -> <tag>_ubench.jsonl
#   tests[:pytest args]             python -m pytest tests -m gpu -q <args>                                        -> <tag>_gputests.log, <tag>_parity_errors.json
#   bench:<workload>[:variant[:extra bench.py args]]   the driver's line (--steps 20 --warmup 5, CPU baseline only for the default library)
#                                                                                                                  -> <tag>_bench_<workload>[_<variant>].json
#   steady:<workload>[:variant]     --steps 200 --no-cpu-baseline                                                  -> <tag>_bench_<workload>[_<variant>]_steady.json
#   prof:<workload>[:variant]       rocprofv3 --kernel-trace --stats of the bench command                          -> <tag>_<workload>[_<variant>]_kernel_stats.md
#   pmc:<workload>[:variant]        the three separate --pmc passes (scripts/pmc_passes.sh)                        -> <tag>_<workload>_pmc.md
#   campaign:<script>:<seed>:<seconds>[:args]   scripts/<script>.py in the BACKGROUND (joined at the end)          -> <tag>_<script>_<seed>.log
#   py:<file>[:args]                python <file> <args>                                                           -> <tag>_<basename>.log
#   bgpy:<file>[:args]              the same in the background (host-only work next to the GPU steps: the full-size oracle fixtures)
#   asan:<seconds>:<seed>:<kinds>:<pool on|off>   tests/host_asan/fuzz_multi against libnmfx_asan.so (python -m nmf_toolbox_amd.build --sanitize; both must be let
#                                   through .gpurunignore for the run)                                             -> <tag>_asan_<kinds>_<pool>.log
#   wait                            join the background campaigns here instead of at the end
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
TAG=$1; shift
for step in "$@"; do
  IFS=':' read -r kind a b c d <<< "$step"
  t0=$(date +%s)
  case $kind in
    ubench) scripts/ubench_mfma ${a:-3000} > gpurun_out/${TAG}_ubench.jsonl 2>&1; cat gpurun_out/${TAG}_ubench.jsonl | cut -c1-260 ;;
    tests)
      python -m pytest tests -m gpu -q --durations=8 $a > gpurun_out/${TAG}_gputests.log 2>&1
      echo "rc $? wall $(( $(date +%s) - t0 )) s" >> gpurun_out/${TAG}_gputests.log
      cp gpurun_out/parity_errors.json gpurun_out/${TAG}_parity_errors.json 2>/dev/null
      tail -14 gpurun_out/${TAG}_gputests.log | cut -c1-220 ;;
    bench|steady)
      sfx=""; [ -n "$b" ] && sfx="_$b"
      if [ $kind = steady ]; then args="--steps 200 --warmup 5 --no-cpu-baseline"; sfx="${sfx}_steady"; else args="--steps 20 --warmup 5"; [ -n "$b" ] && args="$args --no-cpu-baseline"; fi
      NMFX_LIB_VARIANT=$b python bench.py --workload $a $args $c > gpurun_out/${TAG}_bench_${a}${sfx}.json 2> gpurun_out/${TAG}_bench_${a}${sfx}.err
      tail -1 gpurun_out/${TAG}_bench_${a}${sfx}.json | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}
    print('$a$sfx', d['value'], d['unit'], d['ms_per_step'], 'ms |', str(r.get('kernel'))[:48], 'frac', r.get('frac'), 'avg ms', r.get('avg_launch_ms'), '| phases', d.get('phases_ms_per_step'), '| cpu', (d.get('cpu_baseline') or {}).get('value'))
except Exception as e: print('$a$sfx: no line', e)" ;;
    prof)
      sfx=""; [ -n "$b" ] && sfx="_$b"
      (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ks_$a$sfx && NMFX_LIB_VARIANT=$b rocprofv3 --kernel-trace --stats -d /tmp/ks_$a$sfx -o ks -- python $R/bench.py --workload $a --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
       python $R/profiles/summarize_rocprof.py /tmp/ks_$a$sfx/ks_results.db "python bench.py --workload $a --steps 20 --warmup 5 --no-cpu-baseline   (rocprofv3 --kernel-trace --stats; 5 warm-up + 20 timed iterations, averages include the warm-up launches; library variant: ${b:-default})" > $R/gpurun_out/${TAG}_${a}${sfx}_kernel_stats.md)
      head -12 gpurun_out/${TAG}_${a}${sfx}_kernel_stats.md | cut -c1-200 ;;
    pmc) NMFX_LIB_VARIANT=$b bash scripts/pmc_passes.sh $a $TAG; cd $R; grep -E "^## |HBM traffic|MFMA" gpurun_out/${TAG}_${a}_pmc.md | head -12 | cut -c1-220 ;;
    campaign) python scripts/$a.py $b $c $d > gpurun_out/${TAG}_${a}_$b.log 2>&1 & ;;
    bgpy) python $a $b $c $d > gpurun_out/${TAG}_$(basename $a .py).log 2>&1 & ;;
    py) python $a $b $c $d > gpurun_out/${TAG}_$(basename $a .py).log 2>&1; tail -5 gpurun_out/${TAG}_$(basename $a .py).log | cut -c1-300 ;;
    asan)
      log=gpurun_out/${TAG}_asan_${c}_${d:-off}.log
      ( export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=0:print_stacktrace=1:symbolize=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 ASAN_SYMBOLIZER_PATH=/opt/rocm/lib/llvm/bin/llvm-symbolizer
        [ "${d:-off}" = off ] && export NMFX_NO_POOL=1
        FUZZ_NO_WATCHDOG=1 timeout $(( a + 180 )) ./tests/host_asan/fuzz_multi "$a" "$b" "$c" > $log 2>&1; echo "exit $?" >> $log )
      echo "sanitizer reports: $(grep -c 'ERROR: AddressSanitizer\|runtime error' $log)"; grep "fuzz_multi seed\|^exit\|BAD" $log | tail -3 | cut -c1-300 ;;
    wait) wait ;;
    *) echo "unknown step $step" ;;
  esac
  echo "-- $step: $(( $(date +%s) - t0 )) s"
done
wait
for f in gpurun_out/${TAG}_fuzz_*.log; do [ -f "$f" ] && { echo "== $f"; grep -h "BAD\|TRIES" $f | cut -c1-300 | head -8; tail -n 1 $f | cut -c1-520; }; done
exit 0
