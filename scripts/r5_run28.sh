#!/bin/bash
# round-5 host code under ASan + UBSan (pool off): the RCCL backend, cnmfsc's Gram-form W branch, the float64 nmfsc, path 1, next to the round-4 kinds
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash scripts/r4_asan_run.sh 420 61 rcp off r5_rcp
bash scripts/r4_asan_run.sh 420 62 esx1rcp off r5_mix
bash scripts/r4_asan_run.sh 200 63 esx1rcp on r5_mix_poolon
for t in r5_rcp r5_mix r5_mix_poolon; do mv gpurun_out/r4_asan_$t.log gpurun_out/r5_28_asan_$t.log; grep -c "ERROR: AddressSanitizer\|runtime error" gpurun_out/r5_28_asan_$t.log; grep "fuzz_multi seed\|^exit\|BAD" gpurun_out/r5_28_asan_$t.log | tail -4; done
