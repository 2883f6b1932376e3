#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the float64 oracle (oracle/nmf_oracle.py).

The reference is MATLAB and cannot run in the build image (no MATLAB / Octave), and it ships no golden vectors of its
own, so these fixtures pin the ORACLE -- parity is "unpinned by the reference" (see DESIGN.md).
Inputs are regenerated from seeds (numpy.random.RandomState is a frozen legacy generator): V = max(U(0,1), eps) seed
1000, W_init seed 1, H_init seed 2, exactly as tests/conftest.py::synth.

    PYTHONPATH=. python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from conftest import synth  # noqa: E402
from oracle import nmf_oracle as O  # noqa: E402


def save(name, **kw):
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **kw)
    print(name, {k: np.shape(v) for k, v in kw.items()})


def main():
    # --- BASELINE.json configs[0]: nmf.m Euclidean MU, V=512x1024 K=16, 50 iterations (+ the KL twin) -----------------
    m, n, K = 512, 1024, 16
    V, W0, H0 = synth(m, n, K)
    for div in ("euclidean", "kl"):
        tr = []
        W, H, cost = O.nmf(V, K, dict(divergence=div, W_init=W0, H_init=H0, maxiter=50, tolerance=1e-12), trace=tr)
        # full matrices after 50 iterations would be ~200 KB each: keep every 4th row/column plus norms, and the
        # complete state after iterations 1 and 2 in strided form too
        save("nmf_c1_" + div, shape=np.array([m, n, K]), cost=cost, W50_sub=W[::4, :], H50_sub=H[:, ::4], W50_fro=np.linalg.norm(W),
             H50_fro=np.linalg.norm(H), WH50_fro=np.linalg.norm(W @ H), W1_sub=tr[0][0][::4, :], H1_sub=tr[0][1][:, ::4],
             W2_sub=tr[1][0][::4, :], H2_sub=tr[1][1][:, ::4])
    # --- small cases with complete outputs -----------------------------------------------------------------------------
    m, n, K = 96, 160, 8
    V, W0, H0 = synth(m, n, K)
    for div in ("euclidean", "kl", "is"):
        W, H, cost = O.nmf(V, K, dict(divergence=div, W_init=W0, H_init=H0, maxiter=30, tolerance=1e-12))
        save("nmf_small_" + div, shape=np.array([m, n, K]), W=W, H=H, cost=cost)
    cfg = dict(divergence="kl", W_init=[W0[:, :3], W0[:, 3:]], H_init=[H0[:3], H0[3:]], W_sparsity=[0.1, 0.0], H_sparsity=[0.0, 0.2],
               W_fixed=[False, True], H_fixed=[False, False], maxiter=20, tolerance=1e-12)
    W, H, cost = O.nmf(V, [3, 5], cfg)
    save("nmf_small_multi", shape=np.array([m, n, K]), W=np.hstack(W), H=np.vstack(H), cost=cost)
    Wp, Hp, costp = O.nmf(synth(m, n, K, planted=True)[0], K, dict(W_init=W0, H_init=H0, maxiter=400, tolerance=2e-2))
    save("nmf_small_stop", cost=costp, iters=np.array([len(costp)]))
    T = 4
    Vc, Wc0, Hc0 = synth(m, n, 6, T=T)
    for div in ("euclidean", "kl", "frobenius"):
        W, H, cost = O.cnmf(Vc, 6, T, dict(divergence=div, W_init=Wc0, H_init=Hc0, maxiter=20, tolerance=1e-12, W_sparsity=0.01, H_sparsity=0.02))
        save("cnmf_small_" + div, shape=np.array([m, n, 6, T]), W=W, H=H, cost=cost)
    Vs, Ws0, Hs0 = synth(64, 256, 8)
    for tag, sW, sH in (("h", 0.0, 0.5), ("wh", 0.4, 0.6), ("mu", 0.0, 0.0)):
        info = {}
        cfg = dict(W_init=Ws0, H_init=Hs0, maxiter=25, tolerance=1e-12)
        if sW:
            cfg["W_sparsity"] = sW
        if sH:
            cfg["H_sparsity"] = sH
        W, H, cost = O.nmfsc(3.0 * Vs, 8, cfg, info=info)
        save("nmfsc_small_" + tag, W=W, H=H, cost=cost, triesH=np.array(info["triesH"], dtype=np.int32), triesW=np.array(info["triesW"], dtype=np.int32),
             steps=np.array([info["stepsizeH"], info["stepsizeW"]]), sparsity=np.array([sW, sH]))
    Vl, Wl0, Hl0 = synth(96, 160, 8)
    W, H, cost = O.lnmf(Vl, 8, dict(W_init=Wl0, H_init=Hl0, maxiter=30, tolerance=1e-12))
    save("lnmf_small", W=W, H=H, cost=cost)
    Vq, Wq0, Hq0 = synth(48, 120, 5, T=3)
    for tag, sW, sH in (("mu", 0.0, 0.0), ("h", 0.0, 0.5), ("w", 0.3, 0.0)):
        info = {}
        cfg = dict(W_init=Wq0, H_init=Hq0, maxiter=10, tolerance=1e-12)
        if sW:
            cfg["W_sparsity"] = sW
        if sH:
            cfg["H_sparsity"] = sH
        W, H, cost = O.cnmfsc(2.0 * Vq, 5, 3, cfg, info=info)
        save("cnmfsc_small_" + tag, W=W, H=H, cost=cost, triesH=np.array(info["triesH"], dtype=np.int32), triesW=np.array(info["triesW"], dtype=np.int32),
             sparsity=np.array([sW, sH]))
    rs = np.random.RandomState(7)
    S = np.abs(rs.randn(6, 200))
    k1 = np.sqrt(200) - (np.sqrt(200) - 1) * 0.6
    outs, its = zip(*[O.projfunc(s, k1, 1.0, True) for s in S])
    sg = rs.randn(200)
    vs, it_s = O.projfunc(sg, 6.0, 1.0, False)
    save("projfunc", S=S, k1=np.array([k1]), V=np.array(outs), iters=np.array(its, dtype=np.int32), s_signed=sg, v_signed=vs, it_signed=np.array([it_s]))
    Wr, Hr = synth(40, 60, 5, T=3)[1:]
    save("reconstruct", W=Wr, H=Hr, V_hat=O.reconstruct_from_decomposition(Wr, Hr), V_hat_2d=O.reconstruct_from_decomposition(Wr[:, :, 0], Hr))
    # --- constrainednmf (labels -1 = unlabelled) and SortDictionary ----------------------------------
    Vk, Wk0, _ = synth(64, 120, 6)
    labels = np.random.RandomState(11).randint(-1, 4, size=120)
    labels[labels >= 0] = labels[labels >= 0] * 3 + 2          # non-consecutive class ids 2, 5, 8, 11
    nz = int(np.count_nonzero(labels == -1)) + 4
    Z0 = np.fmax(np.random.RandomState(12).rand(6, nz), 2.0 ** -52)
    for div in ("euclidean", "kl"):
        W, H, Z, A, cost = O.constrainednmf(Vk, labels, 6, dict(divergence=div, W_init=Wk0, Z_init=Z0, maxiter=20, tolerance=1e-12, Z_sparsity=0.05))
        save("constrainednmf_" + div, labels=labels, Z0=Z0, W=W, H=H, Z=Z, cost=cost, A_nnz_cols=np.argmax(A, axis=0))
    Wd = np.abs(np.random.RandomState(13).randn(50, 9)) * np.exp(-0.5 * ((np.arange(50)[:, None] - np.array([40, 5, 22, 22, 47, 1, 30, 12, 22])[None, :]) / 4.0) ** 2)
    Wd[:, 3] = Wd[:, 2]                                          # a tie: stable sort keeps 2 before 3
    Hd = np.random.RandomState(14).rand(9, 20)
    Ws, Hs = O.sort_dictionary(Wd, Hd)
    save("sort_dictionary", W=Wd, H=Hd, W_sorted=Ws, H_sorted=Hs)


if __name__ == "__main__":
    main()
