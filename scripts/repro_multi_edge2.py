"""Repro aid (rare host-heap corruption, scripts/fuzz_campaign_r3.py multi_edge seed 16): the first three cases of that seed, with and without the oracle."""
import sys, os
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from conftest import synth
import nmf_toolbox_amd as A
from oracle import nmf_oracle as O
which = sys.argv[1] if len(sys.argv) > 1 else "all"
oracle = (sys.argv[2] if len(sys.argv) > 2 else "1") == "1"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
cases = {"a": ("lnmf", 14, 541, 51, 1, "kl", 4, 2), "b": ("cnmf", 162, 102, 48, 3, "kl", 8, 4), "c": ("nmf", 212, 175, 30, 1, "euclidean", 3, 3)}
for r in range(reps):
    for key, (alg, m, n, K, T, div, N, it) in cases.items():
        if which != "all" and key not in which: continue
        V, W0, H0 = synth(m, n, K, T=(T if alg == "cnmf" else None))
        if alg == "cnmf":
            cfg = dict(divergence=div, W_init=W0, H_init=H0, maxiter=it, tolerance=1e-300)
            if oracle: O.cnmf(V, K, T, cfg)
            A.cnmf(V, K, T, dict(cfg, nmfx_gpus=[0] * N))
        elif alg == "lnmf":
            cfg = dict(W_init=W0 / W0.sum(0), H_init=H0, maxiter=it, tolerance=1e-300)
            if oracle: O.lnmf(V, K, cfg)
            A.lnmf(V, K, dict(cfg, nmfx_gpus=[0] * N))
        else:
            cfg = dict(divergence=div, W_init=W0, H_init=H0, maxiter=it, tolerance=1e-300)
            if oracle: O.nmf(V, K, cfg)
            A.nmf(V, K, dict(cfg, nmfx_gpus=[0] * N))
    print("rep", r, flush=True)
print("done")
