"""One-off fuzz campaign for the paths that are new in round 3 (not part of the suite): KL with K > 256 in column blocks (one GPU and column shards, nmf /
lnmf, sources, sparsity, fixed flags), cnmf / nmfsc / nmf on N shards behind the blocking call, cnmfsc on the fused passes, the Gram-form cost of the
euclidean fused paths over residual levels -- against the float64 oracle.   scripts/fuzz_campaign_r3.py <seed> <seconds>"""
import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from conftest import synth, rel_fro
import nmf_toolbox_amd as A
from oracle import nmf_oracle as O
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 300.0
rs = np.random.RandomState(seed)
KINDS = sys.argv[3].split(",") if len(sys.argv) > 3 else ["klw", "klw", "eucw", "multi_cnmf", "multi_nmfsc", "multi_nmf", "cnmfsc", "gramcost", "cnmf_pad", "is_wide", "multi_edge"]
PAIRS = [(64, 8), (64, 4), (64, 2), (32, 4), (32, 8), (32, 16), (128, 2), (128, 4)]
t0 = time.time(); counts = {}; worst = dict(W=0.0, H=0.0, cost=0.0); bad = []
cat = lambda x: np.concatenate([np.asarray(a).reshape(-1) for a in x]) if isinstance(x, (list, tuple)) else np.asarray(x).reshape(-1)
while time.time() - t0 < budget:
    kind = str(rs.choice(KINDS))
    tries_ok = True
    if kind == "klw":
        K = int(rs.choice([257, 288, 300, 320, 384, 400, 448, 512, 520, 640]))
        m, n = int(rs.randint(64, 500)), int(rs.randint(200, 1500))
        V, W0, H0 = synth(m, n, K)
        cfg = dict(divergence="kl", W_init=W0, H_init=H0, maxiter=int(rs.randint(1, 6)), tolerance=1e-300)
        if rs.rand() < 0.5: cfg["W_sparsity"], cfg["H_sparsity"] = float(rs.rand() * 0.1), float(rs.rand() * 0.1)
        r = rs.rand()
        if r < 0.15: cfg["W_fixed"] = True
        elif r < 0.3: cfg["H_fixed"] = True
        Ks = K
        if rs.rand() < 0.25:
            k1 = int(rs.randint(1, K)); Ks = [k1, K - k1]
            cfg["W_init"] = [W0[:, :k1], W0[:, k1:]]; cfg["H_init"] = [H0[:k1], H0[k1:]]
            cfg["W_fixed"] = [bool(rs.rand() < 0.5), False]; cfg["H_sparsity"] = [0.0, 0.03]; cfg["W_sparsity"] = 0.0; cfg.pop("H_fixed", None)
        extra = {}
        if rs.rand() < 0.3 and n >= 400: extra["nmfx_gpus"] = [0] * int(rs.randint(2, 5))
        if rs.rand() < 0.2 and not isinstance(Ks, list):
            c2 = dict(W_init=W0 / W0.sum(0), H_init=H0, maxiter=cfg["maxiter"], tolerance=1e-300)
            ref = O.lnmf(V, K, c2); got = A.lnmf(V, K, dict(c2, **extra)); kind = "klw_lnmf"
        else:
            ref = O.nmf(V, Ks, cfg); got = A.nmf(V, Ks, dict(cfg, **extra))
        tag = (kind, m, n, K, extra, {k: v for k, v in cfg.items() if k not in ("W_init", "H_init")})
    elif kind == "eucw":   # euclidean above K = 256 (column blocks, Gram-form cost and its switch), one GPU or shards
        K = int(rs.choice([257, 288, 300, 320, 384, 400, 448, 512, 520, 640]))
        m, n = int(rs.randint(64, 500)), int(rs.randint(200, 1500))
        V, W0, H0 = synth(m, n, K, planted=bool(rs.rand() < 0.5))
        cfg = dict(divergence="euclidean", W_init=W0, H_init=H0, maxiter=int(rs.randint(1, 12)), tolerance=1e-300)
        if rs.rand() < 0.5: cfg["W_sparsity"], cfg["H_sparsity"] = float(rs.rand() * 0.1), float(rs.rand() * 0.1)
        r = rs.rand()
        if r < 0.15: cfg["W_fixed"] = True
        elif r < 0.3: cfg["H_fixed"] = True
        extra = {}
        if rs.rand() < 0.3 and n >= 400: extra["nmfx_gpus"] = [0] * int(rs.randint(2, 5))
        ref = O.nmf(V, K, cfg); got = A.nmf(V, K, dict(cfg, **extra))
        tag = (kind, m, n, K, extra, {k: v for k, v in cfg.items() if k not in ("W_init", "H_init")})
    elif kind == "cnmf_pad":   # cnmf with any K padded onto an instantiated (K, T) pair, the round-3 context lengths included
        T = int(rs.randint(2, 17))
        kmax = {2: 256, 3: 128, 4: 128, 5: 64, 6: 64, 7: 64, 8: 64}.get(T, 32)      # the largest instantiated K for that context length: anything below pads up to a pair
        K = int(rs.randint(2, kmax + 1))
        m, n = int(rs.randint(64, 600)), int(rs.randint(max(64, 2 * T), 1200))
        div = str(rs.choice(["euclidean", "kl", "frobenius"]))
        V, W0, H0 = synth(m, n, K, T=T)
        cfg = dict(divergence=div, W_init=W0, H_init=H0, maxiter=int(rs.randint(1, 9)), tolerance=1e-300, nmfx_path=2)
        if rs.rand() < 0.5: cfg["W_sparsity"], cfg["H_sparsity"] = float(rs.rand() * 0.05), float(rs.rand() * 0.05)
        r = rs.rand()
        if r < 0.15: cfg["W_fixed"] = True
        elif r < 0.3: cfg["H_fixed"] = True
        ref = O.cnmf(V, K, T, cfg); got = A.cnmf(V, K, T, cfg)
        tag = (kind, m, n, K, T, div, {k: v for k, v in cfg.items() if k not in ("W_init", "H_init")})
    elif kind == "is_wide":    # IS / alpha-beta on the dual-map kernels above K = 128
        K = int(rs.choice([130, 150, 160, 176, 192]))
        m, n = int(rs.randint(64, 500)), int(rs.randint(64, 1200))
        div = str(rs.choice(["is", "ab"]))
        V, W0, H0 = synth(m, n, K)
        cfg = dict(divergence=div, W_init=W0, H_init=H0, maxiter=int(rs.randint(1, 7)), tolerance=1e-300, nmfx_path=2)
        if div == "ab": cfg["alpha"], cfg["beta"] = [(0.5, 1.5), (2.0, -0.5), (1.0, 0.5), (1.5, 0.2)][rs.randint(4)]
        if rs.rand() < 0.5: cfg["W_sparsity"], cfg["H_sparsity"] = float(rs.rand() * 0.05), float(rs.rand() * 0.05)
        ref = O.nmf(V, K, cfg); got = A.nmf(V, K, cfg)
        tag = (kind, m, n, K, div, {k: v for k, v in cfg.items() if k not in ("W_init", "H_init")})
    elif kind == "dual2":      # round 4: IS / alpha-beta with 192 < K <= 256 as two single-map passes per half-iteration (any K in between through the padding),
        K = int(rs.choice([193, 200, 224, 225, 240, 250, 256]))   # one GPU or ragged column shards, sources / sparsity / fixed factors
        m, n = int(rs.randint(64, 500)), int(rs.randint(200, 1500))
        div = str(rs.choice(["is", "ab"]))
        V, W0, H0 = synth(m, n, K)
        cfg = dict(divergence=div, W_init=W0, H_init=H0, maxiter=int(rs.randint(1, 7)), tolerance=1e-300)
        if div == "ab": cfg["alpha"], cfg["beta"] = [(0.5, 1.5), (2.0, -0.5), (1.0, 0.5), (1.5, 0.2)][rs.randint(4)]
        if rs.rand() < 0.5: cfg["W_sparsity"], cfg["H_sparsity"] = float(rs.rand() * 0.05), float(rs.rand() * 0.05)
        r = rs.rand()
        if r < 0.15: cfg["W_fixed"] = True
        elif r < 0.3: cfg["H_fixed"] = True
        Ks = K
        if rs.rand() < 0.25:
            k1 = int(rs.randint(1, K)); Ks = [k1, K - k1]
            cfg["W_init"] = [W0[:, :k1], W0[:, k1:]]; cfg["H_init"] = [H0[:k1], H0[k1:]]
            cfg["W_fixed"] = [bool(rs.rand() < 0.5), False]; cfg["H_sparsity"] = [0.0, 0.03]; cfg["W_sparsity"] = 0.0; cfg.pop("H_fixed", None)
        extra = {}
        if rs.rand() < 0.4 and n >= 400: extra["nmfx_gpus"] = [0] * int(rs.randint(2, 5))
        ref = O.nmf(V, Ks, cfg); got = A.nmf(V, Ks, dict(cfg, **extra))
        tag = (kind, m, n, K, div, extra, {k: v for k, v in cfg.items() if k not in ("W_init", "H_init")})
    elif kind == "multi_edge":   # the blocking multi-GPU call on awkward geometry: tiny and uneven shards, ragged m, any K, nmf / cnmf / lnmf
        N = int(rs.randint(2, 9))
        alg = str(rs.choice(["nmf", "cnmf", "lnmf"]))
        T = int(rs.randint(2, 6)) if alg == "cnmf" else 1
        m = int(rs.randint(8, 300))
        n = int(rs.randint(N * max(T, 2), 700))
        K = int(rs.randint(2, 70))
        div = "kl" if alg == "lnmf" else str(rs.choice(["euclidean", "kl", "is"] if alg == "nmf" else ["euclidean", "kl"]))
        V, W0, H0 = synth(m, n, K, T=(T if alg == "cnmf" else None))
        it = int(rs.randint(1, 7))
        if os.environ.get("NMFX_FUZZ_VERBOSE"): print("case", alg, m, n, K, T, div, N, it, flush=True)
        skip = os.environ.get("NMFX_FUZZ_SKIP", "")      # repro aid: "gpu" runs the oracle alone, "oracle" the library alone
        if alg == "cnmf":
            cfg = dict(divergence=div, W_init=W0, H_init=H0, maxiter=it, tolerance=1e-300)
            ref = O.cnmf(V, K, T, cfg) if skip != "oracle" else None
            got = A.cnmf(V, K, T, dict(cfg, nmfx_gpus=[0] * N)) if skip != "gpu" else ref
        elif alg == "lnmf":
            cfg = dict(W_init=W0 / W0.sum(0), H_init=H0, maxiter=it, tolerance=1e-300)
            ref = O.lnmf(V, K, cfg) if skip != "oracle" else None
            got = A.lnmf(V, K, dict(cfg, nmfx_gpus=[0] * N)) if skip != "gpu" else ref
        else:
            cfg = dict(divergence=div, W_init=W0, H_init=H0, maxiter=it, tolerance=1e-300)
            if rs.rand() < 0.4: cfg["W_sparsity"], cfg["H_sparsity"] = float(rs.rand() * 0.1), float(rs.rand() * 0.1)
            ref = O.nmf(V, K, cfg) if skip != "oracle" else None
            got = A.nmf(V, K, dict(cfg, nmfx_gpus=[0] * N)) if skip != "gpu" else ref
        if ref is None: ref = got
        tag = (kind, alg, m, n, K, T, div, N, it)
    elif kind == "multi_cnmf":
        K, T = PAIRS[rs.randint(len(PAIRS))] if rs.rand() < 0.7 else (int(rs.randint(3, 20)), int(rs.randint(2, 6)))
        N = int(rs.randint(2, 6))
        m, n = int(rs.randint(64, 400)), int(rs.randint(max(128, N * T * 2), 1200))
        div = str(rs.choice(["euclidean", "kl"]))
        V, W0, H0 = synth(m, n, K, T=T)
        cfg = dict(divergence=div, W_init=W0, H_init=H0, maxiter=int(rs.randint(1, 7)), tolerance=1e-300)
        if rs.rand() < 0.5: cfg["W_sparsity"], cfg["H_sparsity"] = float(rs.rand() * 0.05), float(rs.rand() * 0.05)
        ref = O.cnmf(V, K, T, cfg); got = A.cnmf(V, K, T, dict(cfg, nmfx_gpus=[0] * N))
        tag = (kind, m, n, K, T, div, N)
    elif kind == "multi_nmf":
        K = int(rs.choice([16, 40, 64, 128, 200, 256]))
        N = int(rs.randint(2, 9))
        m, n = int(rs.randint(64, 500)), int(rs.randint(64 * N, 64 * N + 1500))
        div = str(rs.choice(["euclidean", "kl", "is"]))
        if div == "is" and K > 128: K = 128
        V, W0, H0 = synth(m, n, K)
        cfg = dict(divergence=div, W_init=W0, H_init=H0, maxiter=int(rs.randint(1, 8)), tolerance=1e-300)
        ref = O.nmf(V, K, cfg); got = A.nmf(V, K, dict(cfg, nmfx_gpus=[0] * N))
        tag = (kind, m, n, K, div, N)
    elif kind == "multi_nmfsc":
        K = int(rs.choice([20, 32, 64, 100, 128]))
        N = int(rs.randint(2, 5))
        m, n = int(rs.randint(64, 400)), int(rs.randint(64 * N, 64 * N + 1000))
        sW, sH = [(0.0, 0.5), (0.3, 0.0), (0.4, 0.6), (0.0, 0.0)][rs.randint(4)]
        V, W0, H0 = synth(m, n, K)
        cfg = dict(W_init=W0, H_init=H0, maxiter=int(rs.randint(2, 7)), tolerance=1e-300, nmfx_path=2)
        if sW: cfg["W_sparsity"] = sW
        if sH: cfg["H_sparsity"] = sH
        i0, i1 = {}, {}
        ref = O.nmfsc(V, K, cfg, info=i0); got = A.nmfsc(V, K, dict(cfg, nmfx_gpus=[0] * N), info=i1)
        tries_ok = i0["triesH"] == i1["triesH"] and i0["triesW"] == i1["triesW"]
        tag = (kind, m, n, K, sW, sH, N)
    elif kind == "cnmfsc":
        K, T = PAIRS[rs.randint(len(PAIRS))]
        m, n = 4 * int(rs.randint(16, 100)), int(rs.randint(max(64, 2 * T), 900))
        sW, sH = [(0.0, 0.5), (0.0, 0.0), (0.0, 0.7), (0.0, 0.3)][rs.randint(4)]     # (the sparse-W branch ends by step-size underflow after 665 tries: seconds per problem)
        V, W0, H0 = synth(m, n, K, T=T)
        cfg = dict(W_init=W0, H_init=H0, maxiter=int(rs.randint(2, 6)), tolerance=1e-300)
        if sH: cfg["H_sparsity"] = sH
        r = rs.rand()
        if r < 0.15: cfg["W_fixed"] = True
        elif r < 0.3: cfg["H_fixed"] = True
        i0, i1 = {}, {}
        ref = O.cnmfsc(V, K, T, cfg, info=i0); got = A.cnmfsc(V, K, T, dict(cfg, nmfx_path=2), info=i1)
        tries_ok = i0["triesH"] == i1["triesH"] and i0["triesW"] == i1["triesW"]
        tag = (kind, m, n, K, T, sH, {k: v for k, v in cfg.items() if k not in ("W_init", "H_init")})
    else:   # gramcost: euclidean fused nmf / cnmf with the residual anywhere between 30 % and 0.01 % of ||V||^2 (the Gram-form cost and its switch)
        cn = rs.rand() < 0.4
        K, T = PAIRS[rs.randint(len(PAIRS))] if cn else (int(rs.choice([32, 64, 128, 256])), 1)
        m, n = int(rs.randint(64, 500)), int(rs.randint(max(128, 4 * T), 1200))
        V, W0, H0 = synth(m, n, K, T=(T if cn else None), planted=bool(rs.rand() < 0.7))
        if rs.rand() < 0.5: V = V + float(10 ** rs.uniform(-4, -0.5)) * V.mean() * rs.rand(m, n)
        cfg = dict(divergence="euclidean", W_init=W0, H_init=H0, maxiter=int(rs.randint(2, 30)), tolerance=1e-300, nmfx_path=2)
        if cn: ref = O.cnmf(V, K, T, cfg); got = A.cnmf(V, K, T, cfg)
        else: ref = O.nmf(V, K, cfg); got = A.nmf(V, K, cfg)
        tag = (kind, "cnmf" if cn else "nmf", m, n, K, T, cfg["maxiter"], float(ref[2][-1] / (0.5 * (V ** 2).sum())))
    counts[kind] = counts.get(kind, 0) + 1
    same_len = len(got[2]) == len(ref[2])
    fin = same_len and np.all(np.isfinite(ref[2])) and np.linalg.norm(ref[2]) > 0
    e = dict(W=rel_fro(cat(got[0]), cat(ref[0])), H=rel_fro(cat(got[1]), cat(ref[1])), cost=(rel_fro(got[2], ref[2]) if fin else (0.0 if same_len else 1.0)))
    for k in worst: worst[k] = max(worst[k], e[k])
    lim_c = 1e-5 if ((kind == "multi_nmf" and tag[4] == "is") or kind in ("is_wide", "dual2") or (kind == "multi_edge" and tag[6] == "is")) else 1e-6
    if not (e["W"] <= 1e-5 and e["H"] <= 1e-5 and e["cost"] <= lim_c and tries_ok):
        bad.append((tag, e, tries_ok)); print("BAD", tag, e, "tries_ok", tries_ok, flush=True)
print("seed", seed, "cases", counts, "worst", worst, "bad", len(bad))
