"""One-off extended fuzz campaign (not part of the suite): random problems on the paths that changed last in round 2 --
fused cnmf passes (euclidean with lag-form Gram products, kl with the S pass), fused nmf with the merged W update -- against the oracle."""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from conftest import synth, rel_fro
import nmf_toolbox_amd as A
from oracle import nmf_oracle as O
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 300.0
rs = np.random.RandomState(seed)
PAIRS = [(64, 8), (64, 4), (64, 2), (32, 4), (32, 8), (32, 16), (128, 2), (128, 4)]
t0 = time.time(); n_c = n_n = 0; worst = dict(W=0.0, H=0.0, cost=0.0); bad = []
while time.time() - t0 < budget:
    if rs.rand() < 0.6:
        K, T = PAIRS[rs.randint(len(PAIRS))]
        m, n = int(rs.randint(64, 700)), int(rs.randint(max(64, 2 * T), 900))
        div = str(rs.choice(["euclidean", "kl", "frobenius"]))
        V, W0, H0 = synth(m, n, K, T=T)
        cfg = dict(divergence=div, W_init=W0, H_init=H0, maxiter=int(rs.randint(1, 7)), tolerance=1e-300)
        if rs.rand() < 0.5: cfg["W_sparsity"], cfg["H_sparsity"] = float(rs.rand() * 0.05), float(rs.rand() * 0.05)
        r = rs.rand()
        if r < 0.15: cfg["W_fixed"] = True
        elif r < 0.3: cfg["H_fixed"] = True
        Ks = K
        if rs.rand() < 0.25:   # two sources
            k1 = int(rs.randint(1, K)); Ks = [k1, K - k1]
            cfg["W_init"] = [W0[:, :k1], W0[:, k1:]]; cfg["H_init"] = [H0[:k1], H0[k1:]]
            cfg["W_sparsity"] = [0.02, 0.0]; cfg["H_fixed"] = [False, bool(rs.rand() < 0.5)]; cfg.pop("W_fixed", None); cfg["H_sparsity"] = 0.0
        ref = O.cnmf(V, Ks, T, cfg)
        got = A.cnmf(V, Ks, T, dict(cfg, nmfx_path=2))
        tag = ("cnmf", m, n, K, T, div, {k: v for k, v in cfg.items() if k not in ("W_init", "H_init")})
        n_c += 1
    else:
        m, n = int(rs.randint(64, 900)), int(rs.randint(64, 1200))
        K = int(rs.choice([5, 16, 32, 40, 64, 96, 100, 128, 160, 200, 256]))
        div = str(rs.choice(["kl", "euclidean", "is"]))
        if div == "is" and K > 128: K = 128
        V, W0, H0 = synth(m, n, K)
        cfg = dict(divergence=div, W_init=W0, H_init=H0, maxiter=int(rs.randint(1, 8)), tolerance=1e-300)
        if rs.rand() < 0.5: cfg["W_sparsity"], cfg["H_sparsity"] = float(rs.rand() * 0.1), float(rs.rand() * 0.1)
        r = rs.rand()
        if r < 0.15: cfg["W_fixed"] = True
        elif r < 0.3: cfg["H_fixed"] = True
        Ks = K
        if rs.rand() < 0.25 and K > 2:
            k1 = int(rs.randint(1, K)); Ks = [k1, K - k1]
            cfg["W_init"] = [W0[:, :k1], W0[:, k1:]]; cfg["H_init"] = [H0[:k1], H0[k1:]]
            cfg["W_fixed"] = [bool(rs.rand() < 0.5), False]; cfg["H_sparsity"] = [0.0, 0.03]; cfg["W_sparsity"] = 0.0; cfg.pop("H_fixed", None)
        fn = "lnmf" if (div == "kl" and rs.rand() < 0.2 and not isinstance(Ks, list)) else "nmf"
        if fn == "lnmf":
            c2 = dict(W_init=W0, H_init=H0, maxiter=cfg["maxiter"], tolerance=1e-300)
            ref = O.lnmf(V, K, c2); got = A.lnmf(V, K, c2)
        else:
            ref = O.nmf(V, Ks, cfg); got = A.nmf(V, Ks, dict(cfg, nmfx_path=2))
        tag = (fn, m, n, K, div, {k: v for k, v in cfg.items() if k not in ("W_init", "H_init")})
        n_n += 1
    cat = lambda x: np.concatenate([np.asarray(a).reshape(-1) for a in x]) if isinstance(x, (list, tuple)) else np.asarray(x).reshape(-1)
    e = dict(W=rel_fro(cat(got[0]), cat(ref[0])), H=rel_fro(cat(got[1]), cat(ref[1])),
             cost=(rel_fro(got[2], ref[2]) if (len(got[2]) == len(ref[2]) and np.linalg.norm(ref[2]) > 0) else (0.0 if len(got[2]) == len(ref[2]) else 1.0)))
    for k in worst: worst[k] = max(worst[k], e[k])
    lim_c = 1e-5 if tag[0] != "cnmf" and tag[4] == "is" else 1e-6
    if not (e["W"] <= 1e-5 and e["H"] <= 1e-5 and e["cost"] <= lim_c):
        bad.append((tag, e)); print("BAD", tag, e, flush=True)
print("seed", seed, "cnmf cases", n_c, "nmf cases", n_n, "worst", worst, "bad", len(bad))
