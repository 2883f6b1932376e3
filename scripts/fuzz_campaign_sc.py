"""One-off fuzz campaign for the line-search algorithms (not part of the suite): nmfsc / cnmfsc on random problems against the oracle --
identical line-search try counts (the discrete branches of nmfsc.m:158-175, 209-226) and the 1e-5 / 1e-6 contract.
For every case outside the contract the float64 oracle is run a second time on inputs rounded to float32 -- what ANY fp32-storage
implementation starts from: "intrinsic" is how far the float64 algorithm itself moves under that 3e-8 perturbation (the Hoyer
projection after a gradient step amplifies perturbations by 10-1000x on some problems; cnmfsc's sparse-W branch is not even a descent
method, cnmfsc.m:235 compares against the shift-less product)."""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from conftest import synth, rel_fro
import nmf_toolbox_amd as A
from oracle import nmf_oracle as O
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 300.0
import os
FORCE_PATH = int(os.environ.get("NMFX_FUZZ_PATH", "0"))   # 2: nmfsc on the fused MFMA kernels by name (problems this small get the float64 VALU path by default)
rs = np.random.RandomState(seed)
t0 = time.time(); cnt = 0; worst = dict(W=0.0, H=0.0, cost=0.0); bad = 0; mism = 0
while time.time() - t0 < budget:
    conv = rs.rand() < 0.25
    m, n = int(rs.randint(64, 500)), int(rs.randint(64, 700))
    K = int(rs.choice([3, 8, 20, 32, 50, 64, 100, 128]))
    T = int(rs.randint(2, 5)) if conv else 1
    if conv: K = min(K, 32)
    sW, sH = float(rs.choice([0.0, 0.0, 0.3, 0.6])), float(rs.choice([0.0, 0.4, 0.5, 0.7]))
    V, W0, H0 = synth(m, n, K, T=T)
    if rs.rand() < 0.3: V = V * float(rs.choice([0.01, 3.0, 100.0]))
    cfg = dict(W_init=W0 if conv else W0.reshape(m, K), H_init=H0, maxiter=int(rs.randint(2, 8)), tolerance=1e-300)
    if sW: cfg["W_sparsity"] = sW
    if sH: cfg["H_sparsity"] = sH
    r = rs.rand()
    if r < 0.1: cfg["W_fixed"] = True
    elif r < 0.2: cfg["H_fixed"] = True
    i0, i1 = {}, {}
    if len(sys.argv) > 3: print("CASE", "cnmfsc" if conv else "nmfsc", m, n, K, T, sW, sH, {k: v for k, v in cfg.items() if k not in ("W_init", "H_init")}, flush=True)
    if conv:
        ref = O.cnmfsc(V, K, T, cfg, info=i0); got = A.cnmfsc(V, K, T, cfg, info=i1)
    else:
        ref = O.nmfsc(V, K, cfg, info=i0); got = A.nmfsc(V, K, dict(cfg, nmfx_path=FORCE_PATH) if FORCE_PATH else cfg, info=i1)
    cnt += 1
    tag = ("cnmfsc" if conv else "nmfsc", m, n, K, T, sW, sH, {k: v for k, v in cfg.items() if k not in ("W_init", "H_init")})
    if i1.get("triesH") != i0.get("triesH") or i1.get("triesW") != i0.get("triesW") or len(got[2]) != len(ref[2]):
        mism += 1; print("TRIES", tag, i0, i1, flush=True); continue
    e = dict(W=rel_fro(got[0], ref[0]), H=rel_fro(got[1], ref[1]), cost=rel_fro(got[2], ref[2]))
    for k in worst: worst[k] = max(worst[k], e[k])
    if not (e["W"] <= 1e-5 and e["H"] <= 1e-5 and e["cost"] <= 1e-6):
        f32 = lambda x: np.asarray(x, np.float64).astype(np.float32).astype(np.float64)
        cfg32 = dict(cfg, W_init=f32(cfg["W_init"]), H_init=f32(cfg["H_init"]))
        V32 = f32(V / V.max())
        r32 = O.cnmfsc(V32, K, T, cfg32) if conv else O.nmfsc(V32, K, cfg32)
        L = min(len(r32[2]), len(ref[2]))
        intr = dict(W=rel_fro(r32[0], ref[0]), H=rel_fro(r32[1], ref[1]), cost=rel_fro(r32[2][:L], ref[2][:L]))
        bad += 1; print("BAD", tag, e, "intrinsic (float64 algorithm, fp32-rounded inputs)", intr, flush=True)
print("seed", seed, "cases", cnt, "worst", worst, "bad", bad, "try-count mismatches", mism)
