#!/bin/bash
# Everything the FIRST lease of an 8-GPU MI355X node should produce, in one shot (no scaling curve has ever been measured -- this pool has
# 1-GPU boxes -- so the first contact must be self-describing, not a debug session).   bash scripts/first_8gpu_lease.sh [outdir]     (~6 min)
#
# Every line is bench.py's JSON line; for N > 1 it carries world_size_seen, per_rank_ms_per_step, all_ranks_same_path, allreduce_ms_per_step,
# allreduce_busbw_GBps (+ the one-link 153 GB/s and seven-link 1071 GB/s bounds), single_gpu_its_same_run and strong_scaling_eff = value / (N * single).
#   1. the headline (c3: nmf KL 16384 x 65536, K = 256) at N = 1, 2, 4, 8 over RCCL (one process per GPU, torch.distributed nccl)   -> the scaling curve
#   2. N = 8 with the row-chunked W step overlapping the all-reduce: --overlap 2, 4                                                  -> is overlap worth it on xGMI?
#   3. the blocking C-ABI call (single process, 8 devices: what a MEX caller gets) with both exchanges: RCCL (ncclCommInitAll) and the peer
#      reduce-scatter + all-gather                                                                                                   -> A/B of the two backends
#   4. cnmf (c4: halo exchange) and nmfsc (c5: distributed projfunc) at N = 2 and 8
#   5. multi-GPU parity: the sharded tests on REAL devices (nmfx_gpus = [0..N-1]) through scripts/multi_gpu_parity.py
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
OUT=${1:-gpurun_out/first_8gpu}; mkdir -p $OUT
export MASTER_ADDR=127.0.0.1 HSA_ENABLE_IPC_MODE_LEGACY=0
NG=$(python -c "import torch; print(torch.cuda.device_count())")
echo "devices visible: $NG" | tee $OUT/00_devices.txt
rocm-smi --showtopo >> $OUT/00_devices.txt 2>&1
line() { tail -1 "$1" | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read())
    print('%-28s N=%s %8.2f it/s %8.3f ms | ranks %s same_path %s | allreduce %s ms busbw %s GB/s | single %s eff %s' % ('$2', d.get('world_size_seen'), d['value'], d['ms_per_step'],
          (d.get('per_rank_ms_per_step') or {}).get('all'), d.get('all_ranks_same_path'), d.get('allreduce_ms_per_step'), d.get('allreduce_busbw_GBps'), d.get('single_gpu_its_same_run'), d.get('strong_scaling_eff')))
except Exception as e: print('$2: no JSON line (%s)' % e)"; }
run() { # run <name> <bench args...>
  name=$1; shift
  timeout 600 python bench.py "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "rc $?" >> $OUT/$name.err; line $OUT/$name.json $name; }
for N in 1 2 4 8; do [ $N -le $NG ] && run c3_n$N --gpus $N --steps 20 --warmup 5 --no-cpu-baseline; done
if [ $NG -ge 8 ]; then
  for C in 2 4; do run c3_n8_overlap$C --gpus 8 --steps 20 --warmup 5 --no-cpu-baseline --overlap $C; done
  run c3_blocking_n8 --gpus 8 --api blocking --backends rccl,peer --steps 10 --warmup 2 --no-cpu-baseline
  for f in $OUT/c3_blocking_n8.json; do cat $f | cut -c1-600; done
fi
for N in 2 8; do
  [ $N -le $NG ] || continue
  run c4_n$N --workload c4 --gpus $N --steps 20 --warmup 5 --no-cpu-baseline
  run c4kl_n$N --workload c4kl --gpus $N --steps 20 --warmup 5 --no-cpu-baseline
  run c5_n$N --workload c5 --gpus $N --steps 20 --warmup 5 --no-cpu-baseline
  run c2_n$N --workload c2 --gpus $N --steps 20 --warmup 5 --no-cpu-baseline
done
if [ $NG -ge 2 ]; then timeout 900 python scripts/multi_gpu_parity.py $NG > $OUT/parity_real_devices.log 2>&1; tail -5 $OUT/parity_real_devices.log; fi
echo "done: $OUT"
