#!/bin/bash
# experiment: cap on the fp32 accumulation chain of the fused passes (NMFX_CHAIN_MAX) -- parity of the fixed-factor class and cost at C3 / C2
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for CH in 0 2048 1024 512 256 128; do
  export NMFX_CHAIN_MAX=$CH
  python -m pytest tests/test_gpu_conditioning.py -q -k "one_factor or (fused_gram and 1024)" 2>&1 | tail -3 > gpurun_out/r5_02_chain_${CH}.log
  cp gpurun_out/parity_errors.json gpurun_out/r5_02_chain_${CH}_errors.json
  for w in c3 c2; do python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r5_02_bench_${w}_chain_${CH}.json 2>/dev/null; done
  echo "CHAIN $CH"; python - <<PY
import json
d=json.load(open('gpurun_out/r5_02_chain_${CH}_errors.json'))
for t,v in sorted(d['tests'].items()): print('  ', t.split('::')[1][:80], {k:'%.2e'%x for k,x in v.items()})
for w in ('c3','c2'):
    b=json.loads(open('gpurun_out/r5_02_bench_%s_chain_${CH}.json'%w).read().strip().splitlines()[-1]); print('  ', w, b['value'], b['ms_per_step'])
PY
done
