"""Pins that depend on NEITHER restatement of the reference (oracle/nmf_oracle.py, oracle/nmf_oracle.c): every expected value
below follows from the MATLAB lines themselves -- by exact rational arithmetic (`fractions`), by a closed form, or by a
fixed-point argument -- and is checked against whatever implementation is handed in (`impl.nmf / cnmf / projfunc` with the
toolbox call surface).  tests/test_oracle_pins.py runs them on the float64 oracle (tight tolerance) and on deliberately
MUTATED copies of it (each pin set must catch each mutation); tests/test_gpu_pins.py runs them on the HIP path.

The reference ships no vectors, so parity stays "unpinned by the reference"; these pins only remove the
common-mode risk of two restatements written by one reader.
"""
from fractions import Fraction as Fr
import math

import numpy as np

EPS = 2.0 ** -52


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


# ---------------------------------------------------------------------------------------------------------------------
# exact rational matrix helpers (lists of lists of Fraction); MATLAB operator names
# ---------------------------------------------------------------------------------------------------------------------
def fmat(a):
    return [[Fr(x) for x in row] for row in a]


def mt(A):                       # A'
    return [list(r) for r in zip(*A)]


def mm(A, B):                    # A * B
    Bt = mt(B)
    return [[sum((x * y for x, y in zip(r, c)), Fr(0)) for c in Bt] for r in A]


def ew(f, A, B=None):            # element-wise map
    if B is None:
        return [[f(x) for x in r] for r in A]
    return [[f(x, y) for x, y in zip(ra, rb)] for ra, rb in zip(A, B)]


def ddiag(M):                    # diag(diag(M))
    return [[M[i][j] if i == j else Fr(0) for j in range(len(M))] for i in range(len(M))]


def add(A, B):
    return ew(lambda x, y: x + y, A, B)


def ones(r, c):
    return [[Fr(1)] * c for _ in range(r)]


def tofloat(A):
    return np.array([[float(x) for x in r] for r in A], dtype=np.float64)


def fmax_eps(x):                 # max(x, eps): the inputs below keep every denominator far above eps
    assert x > Fr(1, 10 ** 6)
    return x


def nmf_w_step_exact(V, W, H, div, lam=Fr(0)):
    """nmf.m:149-156 + 168, literally (diag(diag(.)) chains and products against ones included), in exact rationals.
    Returns the UN-normalised W of line 168 and the squared column norms line 169 divides by."""
    m, n = len(V), len(V[0])
    Vh = mm(W, H)                                                     # nmf.m:139 / RFD.m:31
    if div == "euclidean":                                            # nmf.m:149-150
        neg = add(mm(V, mt(H)), mm(W, ddiag(mm(mm(H, mt(Vh)), W))))
        pos = add(mm(Vh, mt(H)), mm(W, ddiag(mm(mm(H, mt(V)), W))))
    elif div == "kl":                                                 # nmf.m:152-153
        R = ew(lambda v, s: v / s, V, Vh)
        neg = add(mm(R, mt(H)), mm(W, ddiag(mm(mm(H, ones(n, m)), W))))
        pos = add(mm(ones(m, n), mt(H)), mm(W, ddiag(mm(mm(H, mt(R)), W))))
    elif div == "is":                                                 # nmf.m:155-156
        A = ew(lambda v, s: v / (s * s), V, Vh)
        B = ew(lambda s: 1 / s, Vh)
        neg = add(mm(A, mt(H)), mm(W, ddiag(mm(mm(H, mt(B)), W))))
        pos = add(mm(B, mt(H)), mm(W, ddiag(mm(mm(H, mt(A)), W))))
    else:
        raise ValueError(div)
    U = ew(lambda w, q: w * q, W, ew(lambda a, b: a / fmax_eps(b + lam), neg, pos))   # nmf.m:168
    q = [sum((U[i][k] ** 2 for i in range(m)), Fr(0)) for k in range(len(U[0]))]        # nmf.m:169: sum(W.^2, 1)
    return U, q


def nmf_h_step_exact(V, W, H, div, lam=Fr(0)):
    """nmf.m:180-187 + 199 with V_hat = W*H (nmf.m:173), exact rationals."""
    m, n = len(V), len(V[0])
    Vh = mm(W, H)
    if div == "euclidean":                                            # nmf.m:180-181
        neg, pos = mm(mt(W), V), mm(mt(W), Vh)
    elif div == "kl":                                                 # nmf.m:183-184
        neg, pos = mm(mt(W), ew(lambda v, s: v / s, V, Vh)), mm(mt(W), ones(m, n))
    elif div == "is":                                                 # nmf.m:186-187
        neg, pos = mm(mt(W), ew(lambda v, s: v / (s * s), V, Vh)), mm(mt(W), ew(lambda s: 1 / s, Vh))
    else:
        raise ValueError(div)
    return ew(lambda h, q: h * q, H, ew(lambda a, b: a / fmax_eps(b + lam), neg, pos))   # nmf.m:199


def cost_exact_float(V, Vh, div):
    """nmf.m:206-212 on exact V_hat; the logs are taken in float64 of exactly-rounded ratios"""
    if div == "euclidean":
        return float(Fr(1, 2) * sum(((v - s) ** 2 for rv, rs in zip(V, Vh) for v, s in zip(rv, rs)), Fr(0)))
    if div == "kl":
        return sum(float(v) * math.log(float(v / s)) - float(v) + float(s) for rv, rs in zip(V, Vh) for v, s in zip(rv, rs))
    return sum(math.log(float(s / v)) + float(v / s) - 1.0 for rv, rs in zip(V, Vh) for v, s in zip(rv, rs))


# ---------------------------------------------------------------------------------------------------------------------
# KAT 1 -- worked by hand (every number below can be checked with pencil and paper against nmf.m):
#   V = [1 2; 3 4], K = 1, W_init = [3; 4], H_init = [1 2], 'euclidean', one iteration.
#   nmf.m:133   W = [3;4]/5
#   nmf.m:139   V_hat = W*H = [3/5 6/5; 4/5 8/5]
#   nmf.m:149   V*H' = [5; 11];  H*V_hat'*W = [3 4]*[3/5;4/5] = 5            -> neg = [5;11] + 5*W    = [8; 15]
#   nmf.m:150   V_hat*H' = [3; 4]; H*V'*W = [5 11]*[3/5;4/5] = 59/5          -> pos = [3;4] + 59/5*W  = (84/25)*[3; 4]
#   nmf.m:168   W = W .* neg ./ pos = [10/21; 25/28]
#   nmf.m:169   sum(W.^2) = 7225/7056 = (85/84)^2                            -> W = [8/17; 15/17]
#   nmf.m:173   V_hat = [8/17 16/17; 15/17 30/17]
#   nmf.m:180   W'*V = [53/17 76/17];  nmf.m:181  W'*V_hat = [1 2]           -> H = [53/17 76/17]        (nmf.m:199)
#   nmf.m:203   V_hat = [424 608; 795 1140]/289;  V - V_hat = [-135 -30; 72 16]/289
#   nmf.m:208   cost(1) = 0.5*(135^2+30^2+72^2+16^2)/289^2 = 0.5*24565/83521 = 5/34
# ---------------------------------------------------------------------------------------------------------------------
KAT1 = dict(V=[[1, 2], [3, 4]], W0=[[3], [4]], H0=[[1, 2]], div="euclidean",
            W=[[Fr(8, 17)], [Fr(15, 17)]], H=[[Fr(53, 17), Fr(76, 17)]], cost=Fr(5, 34))


def pin_kat1(impl, tol, cost_tol):
    k = KAT1
    W, H, c = impl.nmf(np.array(k["V"], dtype=np.float64), 1,
                       dict(divergence=k["div"], W_init=np.array(k["W0"], dtype=np.float64), H_init=np.array(k["H0"], dtype=np.float64), maxiter=1))
    assert len(c) == 1
    assert rel(W, tofloat(k["W"])) <= tol and rel(H, tofloat(k["H"])) <= tol, (W, H)
    assert abs(c[0] - float(k["cost"])) <= cost_tol * float(k["cost"]), c


def pin_kat1_selfcheck():
    """the exact helpers reproduce the pencil-and-paper numbers (so they can be trusted on the larger cases)"""
    k = KAT1
    V, H0 = fmat(k["V"]), fmat(k["H0"])
    W0 = [[Fr(3, 5)], [Fr(4, 5)]]
    U, q = nmf_w_step_exact(V, W0, H0, "euclidean")
    assert U == [[Fr(10, 21)], [Fr(25, 28)]] and q == [Fr(7225, 7056)]
    Wn = [[Fr(8, 17)], [Fr(15, 17)]]
    Hn = nmf_h_step_exact(V, Wn, H0, "euclidean")
    assert Hn == k["H"]
    assert Fr(1, 2) * sum(((v - s) ** 2 for rv, rs in zip(V, mm(Wn, Hn)) for v, s in zip(rv, rs)), Fr(0)) == k["cost"]


# ---------------------------------------------------------------------------------------------------------------------
# KAT 2 -- W step alone (H_fixed), K = 2, m = 3, n = 4, all three divergences, with and without W_sparsity.
# Integer data; W_init columns are Pythagorean so line 133 keeps them rational.  The expected W is U / sqrt(q) with U, q exact.
# ---------------------------------------------------------------------------------------------------------------------
KAT2_V = [[2, 1, 3, 5], [4, 6, 1, 2], [3, 2, 7, 1]]
KAT2_W0 = [[2, 1], [1, 2], [2, 2]]            # column norms 3 and 3
KAT2_H0 = [[1, 3, 2, 1], [2, 1, 1, 4]]


def pin_w_step(impl, tol):
    V, H0 = fmat(KAT2_V), fmat(KAT2_H0)
    Wn = [[Fr(x, 3) for x in r] for r in KAT2_W0]                     # nmf.m:133
    for div in ("euclidean", "kl", "is"):
        for lam in (Fr(0), Fr(1, 4)):
            U, q = nmf_w_step_exact(V, Wn, H0, div, lam)
            want = tofloat(U) / np.sqrt(np.array([float(x) for x in q]))[None, :]
            cfg = dict(divergence=div, W_init=np.array(KAT2_W0, dtype=np.float64), H_init=np.array(KAT2_H0, dtype=np.float64),
                       H_fixed=True, W_sparsity=float(lam), maxiter=1)
            W, H, c = impl.nmf(np.array(KAT2_V, dtype=np.float64), 2, cfg)
            assert rel(W, want) <= tol, (div, lam, rel(W, want))
            assert rel(H, np.array(KAT2_H0, dtype=np.float64)) <= 1e-7      # fixed: untouched (fp32 round trip at most)


# ---------------------------------------------------------------------------------------------------------------------
# KAT 3 -- H step alone (W_fixed): W stays at its exactly-normalised init, so H and the cost are exact rationals (+ logs).
# ---------------------------------------------------------------------------------------------------------------------
def pin_h_step(impl, tol, cost_tol):
    V, H0 = fmat(KAT2_V), fmat(KAT2_H0)
    Wn = [[Fr(x, 3) for x in r] for r in KAT2_W0]
    for div in ("euclidean", "kl", "is"):
        for lam in (Fr(0), Fr(1, 8)):
            Hn = nmf_h_step_exact(V, Wn, H0, div, lam)
            cost = cost_exact_float(V, mm(Wn, Hn), div) + float(lam) * float(sum(sum(r) for r in Hn))   # nmf.m:217 (W_sparsity = 0)
            cfg = dict(divergence=div, W_init=np.array(KAT2_W0, dtype=np.float64), H_init=np.array(KAT2_H0, dtype=np.float64),
                       W_fixed=True, H_sparsity=float(lam), maxiter=1)
            W, H, c = impl.nmf(np.array(KAT2_V, dtype=np.float64), 2, cfg)
            assert rel(H, tofloat(Hn)) <= tol, (div, lam, rel(H, tofloat(Hn)))
            assert rel(W, tofloat(Wn)) <= max(tol * 1e-2, 1e-15)
            assert abs(c[0] - cost) <= cost_tol * abs(cost), (div, lam, c[0], cost)


# ---------------------------------------------------------------------------------------------------------------------
# KAT 4 -- two sources (cell arguments): V_hat is NOT refreshed between the sources of one W step (nmf.m:145-173), source 2
# fixed.  Concatenated form: per-column lambda and masks.  Exact W step on source 1 only.
# ---------------------------------------------------------------------------------------------------------------------
def pin_two_sources(impl, tol):
    V = fmat(KAT2_V)
    W1, W2 = [[Fr(2, 3)], [Fr(1, 3)], [Fr(2, 3)]], [[Fr(1, 3)], [Fr(2, 3)], [Fr(2, 3)]]
    H1, H2 = fmat([KAT2_H0[0]]), fmat([KAT2_H0[1]])
    Vh = add(mm(W1, H1), mm(W2, H2))
    # nmf.m:152-153 for source 1 with the JOINT V_hat
    R = ew(lambda v, s: v / s, V, Vh)
    neg = add(mm(R, mt(H1)), mm(W1, ddiag(mm(mm(H1, ones(4, 3)), W1))))
    pos = add(mm(ones(3, 4), mt(H1)), mm(W1, ddiag(mm(mm(H1, mt(R)), W1))))
    U = ew(lambda w, q: w * q, W1, ew(lambda a, b: a / b, neg, pos))
    q = sum((U[i][0] ** 2 for i in range(3)), Fr(0))
    want1 = tofloat(U) / math.sqrt(float(q))
    cfg = dict(divergence="kl", W_init=[np.array([[2.], [1.], [2.]]), np.array([[1.], [2.], [2.]])],
               H_init=[np.array([KAT2_H0[0]], dtype=np.float64), np.array([KAT2_H0[1]], dtype=np.float64)],
               W_fixed=[False, True], H_fixed=[True, True], maxiter=1)
    W, H, c = impl.nmf(np.array(KAT2_V, dtype=np.float64), [1, 1], cfg)
    assert isinstance(W, list) and isinstance(H, list) and len(W) == 2          # cells in, cells out (nmf.m:228-234)
    assert rel(W[0], want1) <= tol and rel(W[1], tofloat(W2)) <= max(tol * 1e-2, 1e-15)


# ---------------------------------------------------------------------------------------------------------------------
# Fixed points: V = W*H exactly representable  =>  V_hat == V  =>  every numerator equals its denominator in
# nmf.m:149-156,180-187 (euclidean: A = V = V_hat = B; KL: A = V./V_hat = 1 = B; IS: A = V./V_hat.^2 = 1./V_hat = B), so one
# iteration is the identity whenever W has unit-L2 columns (nmf.m:169) and lambda = 0; the cost is 0.
# A swapped numerator/denominator, a dropped diag term, a wrong transpose or a missing normalisation all break it -- except
# mutations that are symmetric in (V, V_hat); the KATs above catch those.
# ---------------------------------------------------------------------------------------------------------------------
def _planted(m, n, K, seed):
    rs = np.random.RandomState(seed)
    W = rs.randint(1, 9, size=(m, K)).astype(np.float64)
    H = rs.randint(1, 9, size=(K, n)).astype(np.float64) / 8.0
    W = W / np.sqrt((W ** 2).sum(0))
    return W, H, W @ H


def _fp_cost_ok(div, c, V, cost_rel):
    """|cost| at a fixed point: euclidean is quadratic in the rounding of V_hat, KL / IS are sums of terms LINEAR in it
    (log(1 + d) + 1/(1 + d) - 1 is O(d^2) only in exact arithmetic), so they get sqrt(cost_rel)"""
    if div == "euclidean":
        return np.all(np.abs(c) <= cost_rel * float((V ** 2).sum()))
    return np.all(np.abs(c) <= np.sqrt(cost_rel) * (float(V.sum()) if div == "kl" else float(V.size)))


def pin_nmf_fixed_point(impl, tol, cost_rel, shapes=((6, 9, 2), (17, 23, 4)), iters=3, extra_cfg=None, divs=("euclidean", "kl", "is")):
    for (m, n, K) in shapes:
        W, H, V = _planted(m, n, K, 10 * m + K)
        for div in divs:
            cfg = dict(divergence=div, W_init=W, H_init=H, maxiter=iters, tolerance=1e-300)
            cfg.update(extra_cfg or {})
            Wo, Ho, c = impl.nmf(V, K, cfg)
            assert rel(Wo, W) <= tol and rel(Ho, H) <= tol, (div, m, n, K, rel(Wo, W), rel(Ho, H))
            assert _fp_cost_ok(div, c, V, cost_rel), (div, c)


def pin_cnmf_fixed_point(impl, tol, cost_rel, shapes=((6, 12, 2, 3), (9, 20, 3, 2)), iters=2):
    """cnmf.m:187-232.  Same argument with V = sum_t W_t*rshift_{t-1}(H) and slab norms exactly T (cnmf.m:161-165, 196-199).
    KL is NOT a fixed point in the last T-1 columns of H: cnmf.m:220-221 leaves V_pos unshifted for 'kl', so there
    gpos(k,j) = sum_t colsum(W_t)_k while gneg(k,j) = sum_{t <= n-j+1} colsum(W_t)_k (the left shift runs out of columns);
    H(k,j) is multiplied by their ratio, a closed form in W alone.  After that V_hat ~= V, so only one iteration is pinned for KL."""
    for (m, n, K, T) in shapes:
        rs = np.random.RandomState(7 * m + T)
        W = rs.randint(1, 9, size=(m, K, T)).astype(np.float64)
        W = W / (np.sqrt((W ** 2).sum(axis=(0, 2))) / T)[None, :, None]
        H = rs.randint(1, 9, size=(K, n)).astype(np.float64) / 8.0
        V = np.zeros((m, n))
        for t in range(T):
            V[:, t:] += W[:, :, t] @ H[:, : n - t]
        for div in ("euclidean", "is", "kl"):
            it = 1 if div == "kl" else iters
            Wo, Ho, c = impl.cnmf(V, K, T, dict(divergence=div, W_init=W, H_init=H, maxiter=it, tolerance=1e-300))
            want_H = H.copy()
            if div == "kl":
                cs = W.sum(axis=0)                                   # K x T column sums of every slice
                for j in range(n - T + 1, n):                        # 0-based column j: slices t (0-based) with j + t <= n - 1 survive
                    want_H[:, j] *= cs[:, : n - j].sum(axis=1) / cs.sum(axis=1)
            assert rel(Wo, W) <= tol and rel(Ho, want_H) <= tol, (div, m, n, K, T, rel(Wo, W), rel(Ho, want_H))
            if div != "kl":
                assert _fp_cost_ok(div, c, V, cost_rel), (div, c)


# ---------------------------------------------------------------------------------------------------------------------
# KAT 5 -- cnmf (T = 2) W step alone and H step alone in exact rationals, literally cnmf.m:187-199 and cnmf.m:209-232 with
# (alpha, beta) = (1,1) / (1,0) / (1,-1) for euclidean / kl / is (cnmf.m:137-147).  The fixed points cannot see the shift
# direction of H in the W step (numerator == denominator for ANY H_shifted when V_hat == V); this does.
# Slabs of W_init have Frobenius norms 3 and 6, so the init normalisation (cnmf.m:161-165: norm/T) stays rational.
# ---------------------------------------------------------------------------------------------------------------------
CNMF_V = [[2, 1, 3, 5, 2], [4, 6, 1, 2, 3], [3, 2, 7, 1, 1]]
CNMF_W0 = [[[1, 1], [1, 2]], [[1, 2], [1, 2]], [[1, 1], [1, 5]]]     # W0[i][k][t]; slab k=0: (1,1,1,1,2,1) -> 3, k=1: (1,1,1,2,2,5) -> 6
CNMF_H0 = [[1, 3, 2, 1, 2], [2, 1, 1, 4, 1]]


def _cnmf_setup():
    T, m, K, n = 2, 3, 2, 5
    V = fmat(CNMF_V)
    wn = [Fr(3, T), Fr(6, T)]                                                        # cnmf.m:161  norm(.,'fro') / context_len
    W = [[[Fr(CNMF_W0[i][k][t]) / wn[k] for i in range(m)] for k in range(K)] for t in range(T)]   # W[t][k][i]
    Wt = [mt(W[t]) for t in range(T)]                                                # m x K slices
    H = [[Fr(CNMF_H0[k][j]) * wn[k] for j in range(n)] for k in range(K)]            # cnmf.m:163
    return T, m, K, n, V, Wt, H


def _rshift(H, t, n):            # [zeros(K, t-1) H(:, 1:n-t+1)], t 1-based          cnmf.m:188
    return [[Fr(0)] * (t - 1) + r[: n - t + 1] for r in H]


def _lshift(X, t, n):            # [X(:, t:n) zeros(m, t-1)]                          cnmf.m:219
    return [r[t - 1:] + [Fr(0)] * (t - 1) for r in X]


def _cnmf_vhat(Wt, H, T, n):     # RFD.m:36-38
    Vh = None
    for t in range(1, T + 1):
        P = mm(Wt[t - 1], _rshift(H, t, n))
        Vh = P if Vh is None else add(Vh, P)
    return Vh


def _cnmf_maps(V, Vh, div):      # V.^alpha .* V_hat.^(beta-1), V_hat.^(alpha+beta-1)   cnmf.m:191-192, 213-214
    if div == "euclidean":
        return V, Vh
    if div == "kl":
        return ew(lambda v, s: v / s, V, Vh), ew(lambda s: Fr(1), Vh)
    return ew(lambda v, s: v / (s * s), V, Vh), ew(lambda s: 1 / s, Vh)


def pin_cnmf_kat(impl, tol, cost_tol):
    T, m, K, n, V, Wt, H = _cnmf_setup()
    W0 = np.array(CNMF_W0, dtype=np.float64)
    H0 = np.array(CNMF_H0, dtype=np.float64)
    Vf = np.array(CNMF_V, dtype=np.float64)
    for div in ("euclidean", "kl", "is"):
        for lam in (Fr(0), Fr(1, 4)):
            # ---- W step (H fixed): cnmf.m:187-199
            Vh = _cnmf_vhat(Wt, H, T, n)
            Vn, Vp = _cnmf_maps(V, Vh, div)
            U = []
            for t in range(1, T + 1):
                Hs, W_t = _rshift(H, t, n), Wt[t - 1]
                gneg = add(mm(Vn, mt(Hs)), mm(W_t, ddiag(mm(mm(Hs, mt(Vp)), W_t))))   # cnmf.m:191
                gpos = add(mm(Vp, mt(Hs)), mm(W_t, ddiag(mm(mm(Hs, mt(Vn)), W_t))))   # cnmf.m:192
                U.append(ew(lambda w, q: w * q, W_t, ew(lambda a, b: a / fmax_eps(b + lam), gneg, gpos)))   # cnmf.m:193
            q = [sum((U[t][i][k] ** 2 for t in range(T) for i in range(m)), Fr(0)) for k in range(K)]
            want = np.stack([tofloat(U[t]) for t in range(T)], axis=2) / (np.sqrt(np.array([float(x) for x in q])) / T)[None, :, None]   # cnmf.m:196-199
            Wo, Ho, c = impl.cnmf(Vf, K, T, dict(divergence=div, W_init=W0, H_init=H0, H_fixed=True, W_sparsity=float(lam), maxiter=1))
            assert rel(Wo, want) <= tol, ("cnmf W step", div, lam, rel(Wo, want))
            assert rel(Ho, tofloat(H)) <= max(tol * 1e-2, 1e-15)                      # H only carries the init rescale (cnmf.m:163)
            # ---- H step (W fixed): cnmf.m:209-232
            gneg = gpos = None
            for t in range(1, T + 1):
                Vn_s = _lshift(Vn, t, n)
                Vp_s = Vp if div == "kl" else _lshift(Vp, t, n)                       # cnmf.m:220-224
                a, b = mm(mt(Wt[t - 1]), Vn_s), mm(mt(Wt[t - 1]), Vp_s)
                gneg, gpos = (a, b) if gneg is None else (add(gneg, a), add(gpos, b))
            Hn = ew(lambda h, r: h * r, H, ew(lambda a, b: a / fmax_eps(b + lam), gneg, gpos))   # cnmf.m:231
            cost = cost_exact_float(V, _cnmf_vhat(Wt, Hn, T, n), div) + float(lam) * float(sum(sum(r) for r in Hn))
            Wo, Ho, c = impl.cnmf(Vf, K, T, dict(divergence=div, W_init=W0, H_init=H0, W_fixed=True, H_sparsity=float(lam), maxiter=1))
            assert rel(Ho, tofloat(Hn)) <= tol, ("cnmf H step", div, lam, rel(Ho, tofloat(Hn)))
            assert abs(c[0] - cost) <= cost_tol * abs(cost), (div, lam, c[0], cost)


# ---------------------------------------------------------------------------------------------------------------------
# projfunc closed forms (projfunc.m:22-53).  With nn = 1 and no negative entry arising, the result is the point of the
# circle {sum v = k1, sum v^2 = k2} closest to s: v = c + r*(s - mean(s))/||s - mean(s)||, c = k1/N, r^2 = k2 - k1^2/N.
# N = 2: v = (k1 +- sqrt(2*k2 - k1^2))/2, larger entry where s is larger.
# One forced zero (N = 3, s = (1, 1/2, 0), k1 = 1.2, k2 = 1): the first pass gives v_3 < 0, it is pinned to 0
# (projfunc.m:49-53) and the remaining two solve the N = 2 problem: v = ((6 + sqrt(14))/10, (6 - sqrt(14))/10, 0), 2 iterations.
# ---------------------------------------------------------------------------------------------------------------------
def pin_projfunc(impl, tol):
    v, it = impl.projfunc(np.array([3.0, 1.0]), 1.0, 0.625, 1)                     # sqrt(2*0.625 - 1) = 1/2
    assert rel(v.ravel(), [0.75, 0.25]) <= tol and it == 1
    v, it = impl.projfunc(np.array([0.2, 0.9]), 1.4, 1.0, 1)                       # sqrt(2 - 1.96) = 0.2
    assert rel(v.ravel(), [0.6, 0.8]) <= tol and it == 1
    s = np.array([4.0, 1.0, 2.5, 0.5])
    k1, k2 = 2.0, 1.1
    d = s - s.mean()
    want = k1 / 4 + math.sqrt(k2 - k1 * k1 / 4) * d / np.linalg.norm(d)
    assert want.min() > 0
    v, it = impl.projfunc(s, k1, k2, 1)
    assert rel(v.ravel(), want) <= tol and it == 1
    v, it = impl.projfunc(np.array([1.0, 0.5, 0.0]), 1.2, 1.0, 1)
    r14 = math.sqrt(14.0)
    assert rel(v.ravel(), [(6 + r14) / 10, (6 - r14) / 10, 0.0]) <= tol and it == 2 and v.ravel()[2] == 0.0
    # post-conditions (projfunc.m:3-7) on a long random vector
    rs = np.random.RandomState(3)
    s = rs.rand(1000)
    k1 = math.sqrt(1000) - (math.sqrt(1000) - 1) * 0.7
    v, it = impl.projfunc(s, k1, 1.0, 1)
    v = v.ravel()
    assert abs(v.sum() - k1) <= 10 * tol * k1 and abs((v ** 2).sum() - 1.0) <= 10 * tol and v.min() >= 0
    assert np.array_equal(np.argsort(-s, kind="stable")[:50], np.argsort(-v, kind="stable")[:50])   # order of the large entries is preserved


ALL_NMF_PINS = ("kat1", "w_step", "h_step", "two_sources", "nmf_fixed_point")


def run_nmf_pins(impl, tol, cost_tol, fp_tol=None, fp_cost_rel=1e-25):
    pin_kat1(impl, tol, cost_tol)
    pin_w_step(impl, tol)
    pin_h_step(impl, tol, cost_tol)
    pin_two_sources(impl, tol)
    pin_nmf_fixed_point(impl, fp_tol if fp_tol is not None else tol, fp_cost_rel)
