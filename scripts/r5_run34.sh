#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -q -k "still_materialised or cnmf_ab or cnmf_matches or cnmf_fused" 2>&1 | tail -15 | cut -c1-220
python - <<'PY'
import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np
from conftest import synth, rel_fro
import nmf_toolbox_amd as A
from oracle import nmf_oracle as O
for div, ab, m, n, K, T in [("is", None, 192, 512, 64, 8), ("ab", (0.5, 1.5), 130*4, 777, 32, 4), ("ab", (1.0, 0.5), 256, 1000, 64, 4), ("is", None, 4096, 2048, 64, 8)]:
    V, W0, H0 = synth(m, n, K, T=T)
    cfg = dict(divergence=div, W_init=W0, H_init=H0, maxiter=5, tolerance=1e-12, W_sparsity=0.01)
    if ab: cfg["alpha"], cfg["beta"] = ab
    ref = O.cnmf(V, K, T, cfg)
    got = A.cnmf(V, K, T, dict(cfg, nmfx_path=2)); gen = A.cnmf(V, K, T, dict(cfg, nmfx_path=1))
    print(div, ab, m, n, K, T, "fused vs oracle", rel_fro(got[0], ref[0]), rel_fro(got[1], ref[1]), rel_fro(got[2], ref[2]), "| gemm vs oracle", rel_fro(gen[0], ref[0]), rel_fro(gen[1], ref[1]), rel_fro(gen[2], ref[2]), flush=True)
PY
python bench.py --workload c4is --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5_34_bench_c4is.json 2> gpurun_out/r5_34_bench_c4is.err
tail -1 gpurun_out/r5_34_bench_c4is.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('c4is', d['value'], d['ms_per_step'], d['roofline']['phases_ms_per_step'], d['cost_first_last'])"
tail -2 gpurun_out/r5_34_bench_c4is.err
