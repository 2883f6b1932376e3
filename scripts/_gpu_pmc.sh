# PMC passes (each counter group in its own run, kernel-trace only) of one bench workload: bash scripts/_gpu_pmc.sh <workload> <tag>
cd /tmp && export TMPDIR=/tmp
W=$1; TAG=$2; OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT
for G in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT"; do
  N=$(echo $G | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $G -d $OUT/$N -o p -- python $GRAFT_REPO_ROOT/bench.py --workload $W --steps 5 --warmup 1 --no-cpu-baseline --no-profile > $OUT/$N.log 2>&1
done
cd $GRAFT_REPO_ROOT
python profiles/summarize_pmc.py "rocprofv3 --kernel-trace --pmc <group> -- python bench.py --workload $W --steps 5 --warmup 1 --no-cpu-baseline --no-profile; groups (separate passes): FETCH_SIZE | WRITE_SIZE | SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT" $OUT/FETCH_SIZE/p_results.db $OUT/WRITE_SIZE/p_results.db $OUT/SQ_WAVE_CYCLES/p_results.db > gpurun_out/pmc_$TAG.md
rm -rf $OUT/*/p_results.db
