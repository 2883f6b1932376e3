import sys, os, subprocess
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from conftest import synth, rel_fro
import nmf_toolbox_amd as A
from oracle import nmf_oracle as O
mode = sys.argv[1]
for (m, n, K, it, extra) in [(385, 1459, 640, 11, dict(W_sparsity=0.0993, H_sparsity=0.0551)), (364, 226, 448, 10, {}), (190, 1043, 512, 11, {})]:
    for planted in (False, True):
        V, W0, H0 = synth(m, n, K, planted=planted)
        cfg = dict(divergence="euclidean", W_init=W0, H_init=H0, maxiter=it, tolerance=1e-300, H_fixed=True, **extra)
        ref = O.nmf(V, K, cfg)
        got = A.nmf(V, K, dict(cfg, nmfx_path=1) if mode == "path1" else cfg)
        # the float64 algorithm on inputs rounded to fp32: what any fp32-storage implementation starts from
        cfg32 = dict(cfg, W_init=W0.astype(np.float32).astype(np.float64), H_init=H0.astype(np.float32).astype(np.float64))
        r32 = O.nmf(V.astype(np.float32).astype(np.float64), K, cfg32)
        print(mode, m, n, K, planted, "W err %.2e  cost err %.2e   intrinsic (f64 on fp32-rounded inputs) W %.2e" % (rel_fro(got[0], ref[0]), rel_fro(got[2], ref[2]), rel_fro(r32[0], ref[0])), flush=True)
