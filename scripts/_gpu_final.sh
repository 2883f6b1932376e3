# round-2 final evidence: bench lines for every workload + kernel traces of the two cnmf workloads that changed last
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2_final
for w in c3 c2 c4 c4kl c5 c2is; do
  python bench.py --workload $w --steps 20 --warmup 5 > gpurun_out/r2_final/bench_$w.json 2> gpurun_out/r2_final/bench_$w.err
  tail -1 gpurun_out/r2_final/bench_$w.json | cut -c1-400
done
cd /tmp && export TMPDIR=/tmp
for w in c4 c4kl c3; do
  rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2_final/kt_$w -o kt -- python $GRAFT_REPO_ROOT/bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
  python $GRAFT_REPO_ROOT/profiles/summarize_rocprof.py $GRAFT_REPO_ROOT/gpurun_out/r2_final/kt_$w/kt_results.db "python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline   (rocprofv3 --kernel-trace --stats; 5 warm-up + 20 timed iterations, averages include the warm-up launches)" > $GRAFT_REPO_ROOT/gpurun_out/r2_final/${w}_kernel_stats.md
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/r2_final/kt_$w
done
