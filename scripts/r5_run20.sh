#!/bin/bash
# final evidence of round 5 on the frozen kernel sources: PMC passes, bench lines + kernel stats, the default driver command, the peer exchange on one device
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for w in c3 c2 c4 c4kl c5 c4sc c2is256; do bash scripts/pmc_passes.sh $w r5_20; done
cd $GRAFT_REPO_ROOT
bash scripts/final_evidence.sh r5_20
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/r5_20_bench_default.json 2> gpurun_out/r5_20_bench_default.err
tail -1 gpurun_out/r5_20_bench_default.json | cut -c1-600
NMFX_BENCH_ONE_DEVICE=1 python bench.py --workload c3 --api blocking --gpus 8 --backends peer --steps 10 --host-dtype f32 > gpurun_out/r5_20_blocking_c3_peer_8shards_one_device.json 2> gpurun_out/r5_20_blocking.err
tail -1 gpurun_out/r5_20_blocking_c3_peer_8shards_one_device.json | cut -c1-900
for w in c2 c4 c5 c4sc; do
python bench.py --workload $w --steps 2000 --warmup 100 --no-cpu-baseline > gpurun_out/r5_20_bench_${w}_steady.json 2>/dev/null
tail -1 gpurun_out/r5_20_bench_${w}_steady.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['config']['name'], 'steady', d['value'], d['ms_per_step'])"
done
ls gpurun_out | grep r5_20 | head -60
