"""Repro aid for a rare host-heap corruption seen in scripts/fuzz_campaign_r3.py multi_edge: the blocking multi-GPU call alone, in a loop, on the shapes of the crashing runs."""
import sys, os
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from conftest import synth
import nmf_toolbox_amd as A
mode = sys.argv[1] if len(sys.argv) > 1 else "gpu"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
cases = [("nmf", 100, 90, 45, "kl", 5, 5), ("lnmf", 50, 88, 10, "kl", 7, 3), ("nmf", 91, 232, 8, "kl", 7, 1), ("nmf", 234, 99, 32, "is", 8, 3), ("nmf", 104, 133, 27, "euclidean", 5, 6)]
data = [(c, synth(c[1], c[2], c[3])) for c in cases]
for r in range(reps):
    for (alg, m, n, K, div, N, it), (V, W0, H0) in data:
        if mode == "oracle":
            from oracle import nmf_oracle as O
            (O.lnmf(V, K, dict(W_init=W0 / W0.sum(0), H_init=H0, maxiter=it, tolerance=1e-300)) if alg == "lnmf" else O.nmf(V, K, dict(divergence=div, W_init=W0, H_init=H0, maxiter=it, tolerance=1e-300)))
        elif alg == "lnmf":
            A.lnmf(V, K, dict(W_init=W0 / W0.sum(0), H_init=H0, maxiter=it, tolerance=1e-300, nmfx_gpus=([0] * N if mode == "gpu" else None)))
        else:
            A.nmf(V, K, dict(divergence=div, W_init=W0, H_init=H0, maxiter=it, tolerance=1e-300, nmfx_gpus=([0] * N if mode == "gpu" else None)))
print("done", mode, reps)
