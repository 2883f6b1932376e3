"""One-off campaign (not part of the suite) for the two paths added at the end of round 5: IS / alpha-beta cnmf on the fused passes (engine.fusedT_dual, the eight
(K, T) pairs) and IS / alpha-beta nmf with K > 256 in column blocks (engine.dualw; one GPU and column shards) -- random shapes, sparsity terms, fixed factors, 1-7
iterations, against the float64 oracle.     scripts/fuzz_campaign_dual.py <seed> <seconds>"""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from conftest import synth, rel_fro
import nmf_toolbox_amd as A
from oracle import nmf_oracle as O
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 300.0
rs = np.random.RandomState(seed)
PAIRS = [(64, 8), (64, 4), (32, 8), (32, 16), (64, 2), (32, 4), (128, 2), (128, 4)]
ABS = [(0.5, 1.5), (2.0, -0.5), (1.0, 0.5), (1.5, -1.5), (0.5, 0.5), (1.0, 1.0)]
t0 = time.time(); counts = {}; worst = dict(W=0.0, H=0.0, cost=0.0); bad = 0
while time.time() - t0 < budget:
    kind = str(rs.choice(["cnmf", "wide", "wide_shards"]))
    div = "is" if rs.rand() < 0.5 else "ab"
    cfg = dict(divergence=div, maxiter=int(rs.randint(1, 8)), tolerance=1e-300)
    if div == "ab": cfg["alpha"], cfg["beta"] = ABS[rs.randint(len(ABS))]
    if rs.rand() < 0.4: cfg["W_sparsity"], cfg["H_sparsity"] = float(rs.rand() * 0.1), float(rs.rand() * 0.1)
    if rs.rand() < 0.15: cfg["W_fixed" if rs.rand() < 0.5 else "H_fixed"] = True
    extra = {}
    if kind == "cnmf":
        K, T = PAIRS[rs.randint(len(PAIRS))]
        m, n = 4 * int(rs.randint(16, 150)), int(rs.randint(max(64, 4 * T), 1500))
        V, W0, H0 = synth(m, n, K, T=T)
        cfg.update(W_init=W0, H_init=H0)
        ref = O.cnmf(V, K, T, cfg); got = A.cnmf(V, K, T, cfg)
    else:
        K = int(rs.choice([288, 320, 384, 448, 512, 544, 640]))
        m, n = int(rs.randint(64, 400)), int(rs.randint(200, 1500))
        if kind == "wide_shards" and n >= 400: extra["nmfx_gpus"] = [0] * int(rs.randint(2, 5))
        V, W0, H0 = synth(m, n, K)
        cfg.update(W_init=W0, H_init=H0)
        ref = O.nmf(V, K, cfg); got = A.nmf(V, K, dict(cfg, **extra))
    counts[kind] = counts.get(kind, 0) + 1
    fin = all(np.isfinite(x).all() for x in ref)
    e = dict(W=rel_fro(got[0], ref[0]), H=rel_fro(got[1], ref[1]), cost=rel_fro(got[2], ref[2]) if len(got[2]) == len(ref[2]) else 1.0) if fin else dict(W=0.0, H=0.0, cost=0.0)
    for k in worst: worst[k] = max(worst[k], e[k])
    if not (e["W"] <= 1e-5 and e["H"] <= 1e-5 and e["cost"] <= 1e-6):
        bad += 1; print("BAD", (kind, div, m, n, K, {k: v for k, v in cfg.items() if k not in ("W_init", "H_init")}, extra), e, flush=True)
print("seed", seed, "cases", counts, "total", sum(counts.values()), "worst", worst, "bad", bad)
