cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2_prof; mkdir -p $O
for w in c3 c2 c4 c5; do
  python bench.py --workload $w --steps 20 --warmup 5 > $O/bench_$w.json 2> $O/bench_$w.err
  tail -c 400 $O/bench_$w.json | head -c 400; echo
done
python bench.py --workload c4kl --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c4kl.json 2>/dev/null
for w in c3 c4 c5 c2; do
  rocprofv3 --kernel-trace --stats -d $O/kt_$w -o $w -- python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline > $O/kt_$w.log 2>&1
done
for w in c3 c4; do
  for grp in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT"; do
    tag=$(echo $grp | cut -d' ' -f1)
    rocprofv3 --kernel-trace --pmc $grp -d $O/pmc_${w}_$tag -o $w -- python bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline > $O/pmc_${w}_$tag.log 2>&1
  done
done
ls $O | head -40
