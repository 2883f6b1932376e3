"""Dev aid: per-iteration time of cnmf with K not a multiple of 32 through the blocking call (padded onto the fused shift-sum passes) against the two-operand GEMM path."""
import sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from conftest import synth
import nmf_toolbox_amd as A
from nmf_toolbox_amd import _lib
for (m, n, K, T) in [(513, 4000, 20, 8), (1025, 20000, 40, 8), (513, 8000, 25, 16), (1025, 10000, 100, 4)]:
    V, W0, H0 = synth(m, n, K, T=T)
    for div in ("euclidean", "kl"):
        out = []
        for path in (0, 1):
            cfg = dict(divergence=div, W_init=W0, H_init=H0, maxiter=50, nmfx_disable_stop=True, nmfx_path=path)
            A.cnmf(V, K, T, dict(cfg, maxiter=2))
            A.cnmf(V, K, T, cfg)
            out.append(1e3 * _lib.last_call_timing()["iterate_s"] / 50)
        print("cnmf %s %dx%d K=%d T=%d: fused passes on padded K %.3f ms / iteration, GEMM path %.3f ms" % (div, m, n, K, T, out[0], out[1]), flush=True)
