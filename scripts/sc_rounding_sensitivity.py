"""Which fp32 rounding point of the sparse-H line search of nmfsc costs what: the float64 algorithm with selected
intermediates rounded to float32 (V, H, dH, the stepped vector, W), on the K = 3 problems the fuzz campaign flagged.  CPU only."""
# which fp32 rounding point costs what: float64 nmfsc (sparse H, MU W) with selected intermediates rounded to float32
import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from conftest import synth, rel_fro
from oracle import nmf_oracle as O
f32 = lambda x: x.astype(np.float32).astype(np.float64)
def run(V, W, H, sH, iters, rd):
    K, n = H.shape; m = V.shape[0]
    V = V / V.max()
    if rd.get('V'): V = f32(V)
    L1s = np.sqrt(n) - (np.sqrt(n) - 1) * sH
    H = H.copy(); W = W.copy()
    for k in range(K): H[k] = O.projfunc(H[k], L1s, 1.0, True)[0]
    if rd.get('H'): H = f32(H)
    obj = lambda W, H: 0.5 * np.sum((V - W @ H) ** 2)
    step = 1.0
    for it in range(iters):
        Vh = W @ H
        if rd.get('S'): Vh = f32(Vh)
        dH = W.T @ (Vh - V)
        if rd.get('dH'): dH = f32(dH)
        beg = obj(W, H)
        while True:
            Hn = H - step * dH
            if rd.get('axpy'): Hn = f32(Hn)
            for k in range(K): Hn[k] = O.projfunc(Hn[k], L1s, 1.0, True)[0]
            if rd.get('H'): Hn = f32(Hn)
            if obj(W, Hn) <= beg: break
            step /= 2
        step *= 1.2; H = Hn
        Vh = W @ H
        W = W * (V @ H.T) / np.fmax(Vh @ H.T, O.EPS)
        if rd.get('W'): W = f32(W)
    return W, H
for (m, n, K, sH, it) in [(194, 635, 3, 0.7, 3), (431, 641, 3, 0.5, 7), (325, 202, 3, 0.7, 5), (300, 500, 32, 0.7, 5)]:
    V, W0, H0 = synth(m, n, K)
    Wr, Hr = run(V, W0, H0, sH, it, {})
    ref = O.nmfsc(V, K, dict(W_init=W0, H_init=H0, tolerance=1e-300, maxiter=it, H_sparsity=sH))
    print(m, n, K, "self-check vs oracle", rel_fro(Hr, ref[1]))
    for rd in ({'V':1}, {'H':1}, {'dH':1}, {'axpy':1}, {'W':1}, {'S':1}, {'V':1,'H':1,'dH':1,'axpy':1,'W':1}):
        W, H = run(V, W0, H0, sH, it, rd)
        print("   round", sorted(rd), "W %.2e H %.2e" % (rel_fro(W, Wr), rel_fro(H, Hr)))
