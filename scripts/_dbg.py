import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from conftest import synth, rel_fro
import nmf_toolbox_amd as A
from oracle import nmf_oracle as O
for (m, n, K) in [(129, 131, 96), (256, 512, 96), (256, 131, 96), (129, 512, 96), (129, 131, 64), (129, 131, 128), (129, 131, 32), (129,192,96), (129,193,96)]:
    V, W0, H0 = synth(m, n, K)
    for it in (1,):
        cfg = dict(divergence="is", W_init=W0, H_init=H0, maxiter=it, tolerance=1e-12)
        ref = O.nmf(V, K, cfg)
        f = A.nmf(V, K, dict(cfg, nmfx_path=2))
        hf = A.nmf(V, K, dict(cfg, nmfx_path=2, W_fixed=True))
        hr = O.nmf(V, K, dict(cfg, W_fixed=True))
        print(m, n, K, "W %.2e H %.2e cost %.2e | H-only: H %.2e" % (rel_fro(f[0], ref[0]), rel_fro(f[1], ref[1]), rel_fro(f[2], ref[2]), rel_fro(hf[1], hr[1])))
