#!/usr/bin/env python3
"""Time nmfsc (BASELINE config 5: V=8192x32768, K=128, H_sparsity=0.5) on device-resident data (nmfx_nmfsc_dev, one rank).

Two runs with different maxiter are differenced to get the steady-state time per outer iteration (line-search tries vary
per iteration; they are printed).  Inputs are resident in HBM before the clock starts."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmf_toolbox_amd.engine import nmfsc_sharded  # noqa: E402

m, n, K = (8192, 32768, 128) if len(sys.argv) < 2 else tuple(int(x) for x in sys.argv[1:4])
sH = 0.5 if len(sys.argv) < 5 else float(sys.argv[4])
sW = 0.0 if len(sys.argv) < 6 else float(sys.argv[5])
dev = "cuda:0"
g = torch.Generator(device=dev)
g.manual_seed(1000)
V = torch.rand((n, m), generator=g, device=dev)
g.manual_seed(1)
W0 = torch.rand((K, m), generator=g, device=dev)
g.manual_seed(2)
H0 = torch.rand((n, K), generator=g, device=dev)
res = {}
for iters in (2, 3, 13, 3, 13):   # the first call only warms up
    W, H = W0.clone(), H0.clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    c, info = nmfsc_sharded(V, W, H, W_sparsity=sW, H_sparsity=sH, maxiter=iters, tolerance=-1.0)
    torch.cuda.synchronize()
    res.setdefault(iters, []).append((time.perf_counter() - t0, info["triesH"], info["triesW"], c))
dt = min((a[0] - b[0]) / 10.0 for a in res[13] for b in res[3])
print("nmfsc %dx%d K=%d sW=%g sH=%g: %.3f ms per outer iteration (iterations 4-13, tries H %s W %s), %.1f it/s; cost %.6g -> %.6g"
      % (m, n, K, sW, sH, 1e3 * dt, res[13][0][1][3:], res[13][0][2][3:], 1.0 / dt, res[13][0][3][0], res[13][0][3][-1]))
