"""One-off campaign (not part of the suite) for the class that was outside the contract until round 4: nmf / cnmf with ONE factor fixed for 9-14 iterations,
over-complete or not, random or planted data, on every kernel path (register-stationary kernels, K > 256 in column blocks, Gram form on the GEMM, materialised,
column shards) -- against the float64 oracle.     scripts/fuzz_campaign_fixed_factor.py <seed> <seconds> [kind,kind,...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from conftest import synth, rel_fro
import nmf_toolbox_amd as A
from oracle import nmf_oracle as O
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 300.0
KINDS = sys.argv[3].split(",") if len(sys.argv) > 3 else ["wide", "wide", "fused", "fused", "gram_small", "path1", "cnmf", "shards", "kl"]
rs = np.random.RandomState(seed)
t0 = time.time(); counts = {}; worst = dict(W=0.0, H=0.0, cost=0.0); bad = 0; cost_only = 0
while time.time() - t0 < budget:
    kind = str(rs.choice(KINDS))
    planted = bool(rs.rand() < 0.5)
    fixed = "H_fixed" if rs.rand() < 0.8 else "W_fixed"
    it = int(rs.randint(9, 15))
    T = 1
    if kind in ("wide", "shards"):
        K = int(rs.choice([257, 288, 300, 320, 384, 400, 448, 512, 520, 640])); m, n = int(rs.randint(64, 500)), int(rs.randint(200, 1500))
    elif kind == "fused":
        K = int(rs.choice([32, 64, 100, 128, 192, 250, 256])); m, n = int(rs.randint(64, 600)), int(rs.randint(64, 2500))
    elif kind == "gram_small":
        K = int(rs.randint(20, 300)); m, n = int(rs.randint(8, 64)), int(rs.randint(100, 800))
    elif kind == "path1":
        K = int(rs.choice([40, 128, 200, 320])); m, n = int(rs.randint(64, 400)), int(rs.randint(100, 900))
    elif kind == "kl":
        K = int(rs.choice([64, 128, 256, 320])); m, n = int(rs.randint(64, 500)), int(rs.randint(100, 1500))
    else:
        K, T = [(64, 8), (64, 4), (32, 4), (32, 8), (128, 2), (20, 3), (10, 5)][rs.randint(7)]; m, n = int(rs.randint(64, 400)), int(rs.randint(max(128, 4 * T), 1200))
    V, W0, H0 = synth(m, n, K, T=(T if kind == "cnmf" else None), planted=planted and kind != "cnmf")
    cfg = dict(divergence="kl" if kind == "kl" else "euclidean", W_init=W0, H_init=H0, maxiter=it, tolerance=1e-300)
    cfg[fixed] = True
    if rs.rand() < 0.3: cfg["W_sparsity"], cfg["H_sparsity"] = float(rs.rand() * 0.1), float(rs.rand() * 0.1)
    extra = {}
    if kind == "path1": extra["nmfx_path"] = 1
    if kind == "shards" and n >= 400: extra["nmfx_gpus"] = [0] * int(rs.randint(2, 5))
    if kind == "cnmf": ref = O.cnmf(V, K, T, cfg); got = A.cnmf(V, K, T, dict(cfg, **extra))
    else: ref = O.nmf(V, K, cfg); got = A.nmf(V, K, dict(cfg, **extra))
    counts[kind] = counts.get(kind, 0) + 1
    e = dict(W=rel_fro(got[0], ref[0]), H=rel_fro(got[1], ref[1]), cost=rel_fro(got[2], ref[2]) if len(got[2]) == len(ref[2]) else 1.0)
    for k in worst: worst[k] = max(worst[k], e[k])
    if e["W"] <= 1e-5 and e["H"] <= 1e-5 and 1e-6 < e["cost"] <= 1e-5:
        cost_only += 1   # inside north_star's 1e-5 on every output; past this repo's own 1e-6 on the cost (KL, W fixed, near-perfect fits: cost << sum(V))
    if max(e["W"], e["H"]) > float(os.environ.get("NMFX_FUZZ_REPORT_ABOVE", "1")):   # (inside the contract, but worth a look)
        print("NOTE", (kind, m, n, K, T, planted, fixed, it, extra, {k: v for k, v in cfg.items() if k.endswith("sparsity")}), e, flush=True)
    if not (e["W"] <= 1e-5 and e["H"] <= 1e-5 and e["cost"] <= 1e-6):
        bad += 1; print("BAD", (kind, m, n, K, T, planted, fixed, it, extra, {k: v for k, v in cfg.items() if k.endswith("sparsity")}), e, flush=True)
print("seed", seed, "cases", counts, "total", sum(counts.values()), "worst", worst, "bad", bad, "of which only the cost, between 1e-6 and 1e-5:", cost_only)
