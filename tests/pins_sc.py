"""Pins for the nmfsc / cnmfsc / lnmf / constrainednmf loops that depend on NEITHER restatement under oracle/.

tests/pins.py covers nmf, cnmf and projfunc; this file covers the four loops whose control flow is discrete (line-search
accept / halve / x1.2, early `return`, `<=` stop rule, label bookkeeping) -- exactly where a shared misreading of the MATLAB
would hide.  Every expected value comes from the MATLAB lines themselves:

  * rational steps in `fractions`, square roots / logs taken once at the end (lnmf, constrainednmf, the MU branches);
  * the line-search loops of nmfsc.m:141-245 / cnmfsc.m:155-277 in 50-digit `decimal` arithmetic, with the Hoyer projection NOT
    restated from projfunc.m but replaced by its closed form, which is valid for every vector these KATs ever project:
        N = 4, sparseness 1/4  =>  k1 = sqrt(4) - (sqrt(4) - 1)/4 = 7/4, k2 = 1   (nmfsc.m:89-93,102-106)
        v = k1/N + r * d/||d||,  d = s - mean(s),  r^2 = k2 - k1^2/N = 15/64       (projfunc.m:22-38: v0 = s + (k1 - sum s)/N,
                                                                                    w = v0 - midpoint = d, the '+' root of :37)
        min(v) >= 7/16 - (sqrt(15)/8) * sqrt(3/4) = 0.0182 > 0  for EVERY s, because a zero-mean unit 4-vector has no entry
        below -sqrt(3/4): projfunc.m:40-44 always returns in its first pass (usediters = 1), lines 46-53 never run.
    That is why all projected dimensions below have length 4.
  * a hand-derived case per loop (numbers written out in the comments, asserted by `selfcheck()`), so the transcriptions can be
    trusted on the other cases.

tests/test_oracle_pins.py runs these on the float64 oracle and on MUTATED copies of it (every mutant must be rejected);
tests/test_gpu_pins.py runs them on the HIP path.  Parity stays "unpinned by the reference" (it ships no vectors).
"""
from decimal import Decimal as D, getcontext
from fractions import Fraction as Fr
import math

import numpy as np

from pins import rel, fmat, mt, mm, ew, add, ones, tofloat, cost_exact_float

getcontext().prec = 50
EPS = 2.0 ** -52
K1 = D(7) / D(4)            # L1 target for N = 4, sparseness 1/4
TRIES_TO_UNDERFLOW = 665    # stepsize 2^-(t-1) at try t; after rejecting try t it is 2^-t, and 2^-665 = 6.5e-201 is the first < 1e-200


# ---------------------------------------------------------------------------------------------------------------------
# decimal matrix helpers (lists of rows)
# ---------------------------------------------------------------------------------------------------------------------
def dmat(a):
    return [[D(x) if not isinstance(x, Fr) else D(x.numerator) / D(x.denominator) for x in row] for row in a]


def dT(A):
    return [list(r) for r in zip(*A)]


def dmm(A, B):
    Bt = dT(B)
    return [[sum((x * y for x, y in zip(r, c)), D(0)) for c in Bt] for r in A]


def dew(f, A, B=None):
    if B is None:
        return [[f(x) for x in r] for r in A]
    return [[f(x, y) for x, y in zip(ra, rb)] for ra, rb in zip(A, B)]


def dfloat(A):
    return np.array([[float(x) for x in r] for r in A], dtype=np.float64)


def half_sq(V, Vh):           # 0.5 * sum(sum((V - V_hat).^2))
    return sum(((v - s) ** 2 for rv, rs in zip(V, Vh) for v, s in zip(rv, rs)), D(0)) / 2


def proj4(s):
    """closed form of projfunc(s, 7/4, 1, 1) for a 4-vector (see the module docstring)"""
    assert len(s) == 4
    mean = sum(s, D(0)) / 4
    d = [x - mean for x in s]
    nd = sum((x * x for x in d), D(0)).sqrt()
    assert nd > D("1e-30"), "direction undefined"
    r = (D(15) / D(64)).sqrt()
    v = [K1 / 4 + r * x / nd for x in d]
    assert min(v) > 0
    return v


def proj_rows(H):
    return [proj4(r) for r in H]


def proj_cols(W):
    return dT([proj4(c) for c in dT(W)])


def rshift(H, t):             # [zeros(K, t-1) H(:, 1:n-t+1)]   cnmfsc.m:221
    n = len(H[0])
    return [[D(0)] * (t - 1) + r[: n - t + 1] for r in H]


def lshift(X, t):             # [X(:, t:n) zeros(m, t-1)]       cnmfsc.m:161
    return [r[t - 1:] + [D(0)] * (t - 1) for r in X]


def dmax_eps(x):              # max(x, eps): these KATs keep every such denominator far above eps
    assert x > D("1e-6")
    return x


# ---------------------------------------------------------------------------------------------------------------------
# nmfsc.m:57-245 in 50-digit decimals.  sW / sH in {0, 1/4}; the projected dimension must be 4.
# ---------------------------------------------------------------------------------------------------------------------
def nmfsc_hp(V, W0, H0, sW, sH, maxiter, tol=D("1e-3"), W_fixed=False, H_fixed=False):
    V = dmat(V)
    vmax = max(max(r) for r in V)
    V = dew(lambda x: x / vmax, V)                                    # nmfsc.m:62
    W, H = dmat(W0), dmat(H0)
    K = len(H)
    if sW > 0:
        W = proj_cols(W)                                              # nmfsc.m:93-96
    if sH > 0:
        H = proj_rows(H)                                              # nmfsc.m:106-109
    stepW = stepH = D(1)                                              # nmfsc.m:133-134
    Vh = dmm(W, H)                                                    # nmfsc.m:138
    cost = [half_sq(V, Vh)]                                           # nmfsc.m:139
    triesH, triesW, margins = [], [], []
    out = lambda early: dict(W=W, H=H, cost=cost, triesH=triesH, triesW=triesW, stepH=stepH, stepW=stepW, early=early, margins=margins)
    for it in range(1, maxiter + 1):                                  # nmfsc.m:141
        if not H_fixed:
            neg, pos = dmm(dT(W), V), dmm(dT(W), Vh)                  # nmfsc.m:144-145
            if sH > 0:
                dH = dew(lambda p, q: p - q, pos, neg)                # nmfsc.m:148
                begobj = cost[it - 1]                                 # nmfsc.m:149
                tries = 0
                while True:
                    tries += 1
                    Hnew = proj_rows(dew(lambda h, g: h - stepH * g, H, dH))   # nmfsc.m:154-157
                    Vh = dmm(W, Hnew)                                 # nmfsc.m:160
                    newobj = half_sq(V, Vh)                           # nmfsc.m:161
                    margins.append(abs(newobj - begobj) / begobj)
                    if newobj <= begobj:                              # nmfsc.m:164
                        break
                    stepH = stepH / 2                                 # nmfsc.m:169
                    if stepH < D("1e-200"):                           # nmfsc.m:170-174: cost = cost(1:iter); return
                        triesH.append(tries)
                        return out(True)
                triesH.append(tries)
                stepH = D("1.2") * stepH                              # nmfsc.m:178
                H = Hnew                                              # nmfsc.m:179
            else:
                H = dew(lambda h, q: h * q, H, dew(lambda a, b: a / dmax_eps(b), neg, pos))   # nmfsc.m:182
                norms = [sum((x * x for x in r), D(0)).sqrt() for r in H]                     # nmfsc.m:185
                H = [[x / norms[k] for x in H[k]] for k in range(K)]                          # nmfsc.m:186
                W = [[W[i][k] * norms[k] for k in range(K)] for i in range(len(W))]          # nmfsc.m:187
        if not W_fixed:
            Vh = dmm(W, H)                                            # nmfsc.m:193
            neg, pos = dmm(V, dT(H)), dmm(Vh, dT(H))                  # nmfsc.m:194-195
            if sW > 0:
                begobj = half_sq(V, Vh)                               # nmfsc.m:197  (recomputed, NOT cost(iter))
                dW = dew(lambda p, q: p - q, pos, neg)                # nmfsc.m:200
                tries = 0
                while True:
                    tries += 1
                    Wnew = proj_cols(dew(lambda w, g: w - stepW * g, W, dW))   # nmfsc.m:205-208
                    Vh = dmm(Wnew, H)                                 # nmfsc.m:211
                    newobj = half_sq(V, Vh)
                    margins.append(abs(newobj - begobj) / begobj)
                    if newobj <= begobj:                              # nmfsc.m:215
                        break
                    stepW = stepW / 2                                 # nmfsc.m:220
                    if stepW < D("1e-200"):                           # nmfsc.m:221-225
                        triesW.append(tries)
                        return out(True)
                triesW.append(tries)
                stepW = D("1.2") * stepW                              # nmfsc.m:228
                W = Wnew                                              # nmfsc.m:229
            else:
                W = dew(lambda w, q: w * q, W, dew(lambda a, b: a / dmax_eps(b), neg, pos))   # nmfsc.m:232 (no normalisation here)
        Vh = dmm(W, H)                                                # nmfsc.m:237
        cost.append(half_sq(V, Vh))                                   # nmfsc.m:238
        if it > 1 and cost[it] < cost[it - 1] and cost[it - 1] - cost[it] < tol:   # nmfsc.m:241-244
            break
    return out(False)


# ---------------------------------------------------------------------------------------------------------------------
# cnmfsc.m:67-277 in 50-digit decimals (W given as a list of T slices, each m x K)
# ---------------------------------------------------------------------------------------------------------------------
def rfd(Wl, H):               # ReconstructFromDecomposition.m:30-38
    Vh = None
    for t in range(1, len(Wl) + 1):
        P = dmm(Wl[t - 1], rshift(H, t))
        Vh = P if Vh is None else dew(lambda a, b: a + b, Vh, P)
    return Vh


def cnmfsc_hp(V, W0l, H0, sW, sH, maxiter, tol=D("1e-3"), W_fixed=False, H_fixed=False, eps=D(2) ** -52):
    V = dmat(V)
    vmax = max(max(r) for r in V)
    V = dew(lambda x: x / vmax, V)                                    # cnmfsc.m:72
    T = len(W0l)
    W0 = [dmat(w) for w in W0l]                                       # cnmfsc.m:93
    W = [[list(r) for r in w] for w in W0]                            # cnmfsc.m:94
    H = dmat(H0)
    K, m = len(H), len(W0[0])
    if sW > 0:
        W = [proj_cols(w) for w in W]                                 # cnmfsc.m:105-109: W, NOT W0
    if sH > 0:
        H = proj_rows(H)                                              # cnmfsc.m:121-123
    stepW, stepH = [D(1)] * T, D(1)                                   # cnmfsc.m:147-148
    Vh = rfd(W, H)                                                    # cnmfsc.m:152
    cost = [half_sq(V, Vh)]
    triesH, triesW, margins = [], [], []
    out = lambda early: dict(W=W, H=H, cost=cost, triesH=triesH, triesW=triesW, stepH=stepH, stepW=stepW, early=early, margins=margins)
    for it in range(1, maxiter + 1):
        if not H_fixed:
            neg = pos = None
            for t in range(1, T + 1):                                 # cnmfsc.m:160-165: on W0
                a, b = dmm(dT(W0[t - 1]), lshift(V, t)), dmm(dT(W0[t - 1]), lshift(Vh, t))
                neg, pos = (a, b) if neg is None else (dew(lambda x, y: x + y, neg, a), dew(lambda x, y: x + y, pos, b))
            if sH > 0:
                dH = dew(lambda p, q: p - q, pos, neg)                # cnmfsc.m:168
                begobj = cost[it - 1]
                tries = 0
                while True:
                    tries += 1
                    Hnew = proj_rows(dew(lambda h, g: h - stepH * g, H, dH))
                    Vh = rfd(W0, Hnew)                                # cnmfsc.m:180
                    newobj = half_sq(V, Vh)
                    margins.append(abs(newobj - begobj) / begobj)
                    if newobj <= begobj:
                        break
                    stepH = stepH / 2
                    if stepH < D("1e-200"):
                        triesH.append(tries)
                        return out(True)
                triesH.append(tries)
                stepH = D("1.2") * stepH
                H = Hnew
            else:
                H = dew(lambda h, q: h * q, H, dew(lambda a, b: a / (b + eps), neg, pos))     # cnmfsc.m:202: pos + eps, NOT max(pos, eps)
                norms = [sum((x * x for x in r), D(0)).sqrt() for r in H]                     # cnmfsc.m:205
                H = [[x / norms[k] for x in H[k]] for k in range(K)]
                W0 = [[[w[i][k] * norms[k] for k in range(K)] for i in range(m)] for w in W0]   # cnmfsc.m:207-209
        if not W_fixed:
            Vh = rfd(W0, H)                                           # cnmfsc.m:215
            if sW > 0:
                for t in range(1, T + 1):
                    begobj = half_sq(V, Vh)                           # cnmfsc.m:218: against whatever V_hat the previous t left
                    Hs = rshift(H, t)
                    dW = dew(lambda p, q: p - q, dmm(Vh, dT(Hs)), dmm(V, dT(Hs)))   # cnmfsc.m:222-224
                    tries = 0
                    while True:
                        tries += 1
                        Wnew = proj_cols(dew(lambda w, g: w - stepW[t - 1] * g, W0[t - 1], dW))   # cnmfsc.m:229-233
                        Vh = dmm(Wnew, H)                             # cnmfsc.m:235: a 2-D Wnew takes RFD.m:30-31, plain Wnew*H, no shift
                        newobj = half_sq(V, Vh)
                        margins.append(abs(newobj - begobj) / begobj)
                        if newobj <= begobj:
                            break
                        stepW[t - 1] = stepW[t - 1] / 2
                        if stepW[t - 1] < D("1e-200"):                # cnmfsc.m:245-249
                            triesW.append(tries)
                            return out(True)
                    triesW.append(tries)
                    stepW[t - 1] = D("1.2") * stepW[t - 1]
                    W[t - 1] = Wnew
            else:
                for t in range(1, T + 1):                             # cnmfsc.m:257-263
                    Hs = rshift(H, t)
                    neg, pos = dmm(V, dT(Hs)), dmm(Vh, dT(Hs))
                    W[t - 1] = dew(lambda w, q: w * q, W0[t - 1], dew(lambda a, b: a / dmax_eps(b), neg, pos))
                    dWt = dew(lambda a, b: a - b, W[t - 1], W0[t - 1])
                    Vh = dew(lambda a, b: max(a + b, D(0)), Vh, dmm(dWt, Hs))   # cnmfsc.m:262
        W0 = [[list(r) for r in w] for w in W]                        # cnmfsc.m:266
        Vh = rfd(W0, H)                                               # cnmfsc.m:269
        cost.append(half_sq(V, Vh))
        if it > 1 and cost[it] < cost[it - 1] and cost[it - 1] - cost[it] < tol:
            break
    return out(False)


def _w3(Wl):                  # list of T (m x K) slices -> m x K x T float array
    return np.stack([dfloat(w) for w in Wl], axis=2)


# ---------------------------------------------------------------------------------------------------------------------
# KAT S1 -- nmfsc, sparse-W branch gives up by step-size underflow in iteration 1 (nmfsc.m:220-225), worked by hand.
#   V = 2*v*[1 1 1 1] with v = [1; 1/2; 1/2; 1] after nmfsc.m:62 (max(V(:)) = 2), K = 1, W_init = [3;1;1;3], H_init = [2 2 2 2],
#   W_sparsity = 1/4, H_sparsity unset.   ||v||^2 = 5/2, sum(v) = 3.
#   nmfsc.m:93-96   W = projfunc(W_init, 7/4, 1, 1) =: w.  d = [1 -1 -1 1], ||d|| = 2:  w = 7/16 + sqrt(15)/16 * [1 -1 -1 1]
#                   s := w'*v = (7/16)*3 + (sqrt(15)/16)*(1 - 1/2 - 1/2 + 1) = (21 + sqrt(15))/16 = 1.55456...
#   nmfsc.m:139     V_hat = 2*w*1'; per column ||v - 2w||^2 = 5/2 - 4s + 4       -> cost(1) = 0.5*4*(13/2 - 4s) = 13 - 8s = (5 - sqrt(15))/2
#   nmfsc.m:144-145 W'*V = s*1', W'*V_hat = 2*1'                                  -> H = 2*s/2 = s each                (:182)
#   nmfsc.m:185-187 norms = 2s                                                    -> H = [1 1 1 1]/2,  W = 2s*w
#   nmfsc.m:197     V_hat = s*w*1' (the projection of v on w)                     -> begobj = 0.5*4*(5/2 - s^2) = 5 - 2s^2 = 0.1667
#   nmfsc.m:200     dW = V_hat*H' - V*H' = 2s*w - 2v, so W - mu*dW = 2s(1-mu)*w + 2mu*v is never a constant vector (projfunc defined)
#   nmfsc.m:205-212 ANY Wnew out of projfunc has unit norm: per column ||v - Wnew/2||^2 = 5/2 - Wnew'*v + 1/4 >= 11/4 - ||v||
#                   -> newobj >= 11/2 - 2*sqrt(5/2) = 2.338 > begobj for every step size: 665 rejected tries, then
#                   `cost = cost(1:iter); return` with iter = 1
#   returned: W = 2s*w (line 187's W, not Wnew), H = [1 1 1 1]/2, cost = [(5 - sqrt(15))/2], stepsizeW = 2^-665; no H line search.
# ---------------------------------------------------------------------------------------------------------------------
S1 = dict(V=[[2] * 4, [1] * 4, [1] * 4, [2] * 4], W0=[[3], [1], [1], [3]], H0=[[2, 2, 2, 2]], sW=0.25)


def _s1_expected():
    r15 = math.sqrt(15.0)
    w = np.array([7 / 16 + r15 / 16 * x for x in (1, -1, -1, 1)])
    s = (21 + r15) / 16
    return 2 * s * w[:, None], np.full((1, 4), 0.5), np.array([(5 - r15) / 2])


def pin_nmfsc_underflow(impl, tol, cost_tol, extra_cfg=None):
    k = S1
    cfg = dict(W_init=np.array(k["W0"], dtype=np.float64), H_init=np.array(k["H0"], dtype=np.float64), W_sparsity=k["sW"], maxiter=3)
    cfg.update(extra_cfg or {})
    info = {}
    W, H, c = impl.nmfsc(3.0 * np.array(k["V"], dtype=np.float64), 1, cfg, info=info)     # 3*V: nmfsc.m:62 divides by max(V(:))
    We, He, ce = _s1_expected()
    assert len(c) == 1, c                                                                  # trimmed to cost(1:iter), iter = 1
    assert abs(c[0] - ce[0]) <= cost_tol * ce[0], c
    assert info["triesW"] == [TRIES_TO_UNDERFLOW] and info["triesH"] == [], info
    assert rel(W, We) <= tol and rel(H, He) <= tol, (W, H)
    assert 0 < info["stepsizeW"] < 1e-200


# ---------------------------------------------------------------------------------------------------------------------
# KAT S2 -- nmfsc MU branches alone (no sparsity): nmfsc.m:182-187 (H update, row-norm rescale of H AND W) then :232 (W update
# WITHOUT any normalisation or diag terms), in exact rationals; the only irrational step is the row norm.
# ---------------------------------------------------------------------------------------------------------------------
S2_V = [[2, 1, 3, 4], [4, 3, 1, 2], [3, 2, 4, 1]]
S2_W0 = [[2, 1], [1, 2], [2, 2]]
S2_H0 = [[1, 3, 2, 1], [2, 1, 1, 4]]


def _nmfsc_mu_exact(V, W, H):
    """one iteration of nmfsc.m:141-238 with W_sparsity = H_sparsity = 0; returns float W, H, cost(2)"""
    Vh = mm(W, H)
    neg, pos = mm(mt(W), V), mm(mt(W), Vh)                            # nmfsc.m:144-145
    H = ew(lambda h, q: h * q, H, ew(lambda a, b: a / b, neg, pos))   # nmfsc.m:182
    n2 = [sum((x * x for x in r), Fr(0)) for r in H]                  # squared row norms, exact
    Hf = tofloat(H) / np.sqrt(np.array([float(x) for x in n2]))[:, None]          # nmfsc.m:186
    Wf = tofloat(W) * np.sqrt(np.array([float(x) for x in n2]))[None, :]          # nmfsc.m:187
    # W*H is unchanged by the rescale, so nmfsc.m:193-195 can stay rational: with H = D^-1*Hr, W = Wr*D,
    #   V*H' = (V*Hr')*D^-1 and V_hat*H' = (V_hat*Hr')*D^-1: the ratio of line 232 does not see D at all
    Vh = mm(W, H)
    q = ew(lambda a, b: a / b, mm(V, mt(H)), mm(Vh, mt(H)))           # nmfsc.m:194-195,232
    Wf = Wf * tofloat(q)
    Wr = ew(lambda w, x: w * x, W, q)
    cost = float(Fr(1, 2) * sum(((v - s) ** 2 for rv, rs in zip(V, mm(Wr, H)) for v, s in zip(rv, rs)), Fr(0)))   # nmfsc.m:238
    return Wf, Hf, cost


def pin_nmfsc_mu(impl, tol, cost_tol, extra_cfg=None):
    V = [[Fr(x, 4) for x in r] for r in S2_V]                         # nmfsc.m:62: max(V(:)) = 4
    W, H = fmat(S2_W0), fmat(S2_H0)
    c1 = float(Fr(1, 2) * sum(((v - s) ** 2 for rv, rs in zip(V, mm(W, H)) for v, s in zip(rv, rs)), Fr(0)))       # nmfsc.m:139
    We, He, c2 = _nmfsc_mu_exact(V, W, H)
    cfg = dict(W_init=np.array(S2_W0, dtype=np.float64), H_init=np.array(S2_H0, dtype=np.float64), maxiter=1)
    cfg.update(extra_cfg or {})
    Wo, Ho, c = impl.nmfsc(np.array(S2_V, dtype=np.float64), 2, cfg)
    assert len(c) == 2 and abs(c[0] - c1) <= cost_tol * c1 and abs(c[1] - c2) <= cost_tol * c2, (c, c1, c2)
    assert rel(Wo, We) <= tol and rel(Ho, He) <= tol, (rel(Wo, We), rel(Ho, He))
    assert abs(np.sqrt((np.asarray(Ho) ** 2).sum(1)) - 1).max() <= 10 * tol     # rows of H leave line 186 with unit norm


# ---------------------------------------------------------------------------------------------------------------------
# KAT S3 / S4 -- the line searches.  Inputs are small integers found by search so that the FIRST try is rejected and the SECOND
# accepted (margins >= 1e-3 relative on both comparisons, asserted in selfcheck): tries = [2], stepsize = 0.5 * 1.2 = 0.6.
#   S3: sparse H (n = 4), W by the MU rule of line 232.   S4: sparse W (m = 4), H by the MU rule + rescale of lines 182-187 (K = 2).
#   S4B: sparse W with H_fixed (K = 1).
#   S5: both line searches, two iterations, K = 2 (m = n = 4): tries and step sizes carry over between iterations.
# Expected values: nmfsc_hp above (50 digits).
# ---------------------------------------------------------------------------------------------------------------------
S3 = dict(V=[[4, 2, 1, 3], [3, 4, 3, 3]], W0=[[1], [1]], H0=[[4, 4, 3, 3]], sW=0, sH=0.25, K=1)
S4 = dict(V=[[4, 2, 1], [1, 3, 1], [4, 2, 1], [3, 2, 1]], W0=[[2, 4], [4, 2], [2, 5], [5, 4]], H0=[[1, 2, 2], [1, 1, 3]], sW=0.25, sH=0, K=2)
S4B = dict(V=[[4, 3], [3, 2], [4, 2], [1, 3]], W0=[[5], [2], [1], [1]], H0=[[1, 1]], sW=0.25, sH=0, K=1, H_fixed=True)
S5 = dict(V=[[4, 3, 4, 1], [2, 1, 3, 2], [1, 3, 4, 4], [2, 4, 2, 3]], W0=[[4, 2], [5, 2], [3, 1], [5, 4]], H0=[[5, 3, 4, 2], [5, 3, 4, 1]],
          sW=0.25, sH=0.25, K=2)
SC_CASES = dict(S3=S3, S4=S4, S4B=S4B, S5=S5)


def _run_hp(k, maxiter):
    return nmfsc_hp(k["V"], k["W0"], k["H0"], D(str(k["sW"])), D(str(k["sH"])), maxiter, tol=D("1e-300"), H_fixed=k.get("H_fixed", False))


def pin_nmfsc_linesearch(impl, tol, cost_tol, extra_cfg=None, cases=("S3", "S4", "S4B", "S5")):
    for name in cases:
        k = SC_CASES[name]
        iters = 2 if name == "S5" else 1
        e = _run_hp(k, iters)
        cfg = dict(W_init=np.array(k["W0"], dtype=np.float64), H_init=np.array(k["H0"], dtype=np.float64), maxiter=iters, tolerance=1e-300)
        if k["sW"]:
            cfg["W_sparsity"] = k["sW"]
        if k["sH"]:
            cfg["H_sparsity"] = k["sH"]
        if k.get("H_fixed", False):
            cfg["H_fixed"] = True
        cfg.update(extra_cfg or {})
        info = {}
        W, H, c = impl.nmfsc(np.array(k["V"], dtype=np.float64), k["K"], cfg, info=info)
        assert info["triesH"] == e["triesH"] and info["triesW"] == e["triesW"], (name, info, e["triesH"], e["triesW"])
        ce = np.array([float(x) for x in e["cost"]])
        assert len(c) == len(ce) and np.all(np.abs(c - ce) <= cost_tol * np.abs(ce)), (name, c, ce)
        assert rel(W, dfloat(e["W"])) <= tol and rel(H, dfloat(e["H"])) <= tol, (name, rel(W, dfloat(e["W"])), rel(H, dfloat(e["H"])))
        if k["sH"]:
            assert abs(info["stepsizeH"] - float(e["stepH"])) <= 1e-12 * float(e["stepH"]), (name, info["stepsizeH"], e["stepH"])
        if k["sW"]:
            assert abs(info["stepsizeW"] - float(e["stepW"])) <= 1e-12 * float(e["stepW"]), (name, info["stepsizeW"], e["stepW"])


# ---------------------------------------------------------------------------------------------------------------------
# KAT L1 -- lnmf.m:59-81, one iteration in exact rationals (sqrt at the end of line 76, logs in line 81).
#   L1-normalised columns (line 59 and again line 70), NO diag terms, plain ratio, H <- sqrt(H .* (W'*(V./V_hat))).
# KAT L2 -- lnmf.m:84-86: `<=` in BOTH comparisons and a bare `break` (cost keeps its maxiter entries).  With W_fixed and H_fixed
#   nothing changes, so cost(2) == cost(1) =: c exactly: the loop stops at iter 2 and cost = [c c 0 0 0].  nmf's strict rule
#   (nmf.m:221) would run to maxiter ([c c c c c]); a trimming implementation would return two entries.
# ---------------------------------------------------------------------------------------------------------------------
L_V = [[2, 1, 3, 5], [4, 6, 1, 2], [3, 2, 7, 1]]
L_W0 = [[2, 1], [1, 2], [1, 5]]            # column sums 4 and 8
L_H0 = [[1, 3, 2, 1], [2, 1, 1, 4]]


def _lnmf_exact(V, W0, H):
    cs = [sum(W0[i][k] for i in range(len(W0))) for k in range(len(W0[0]))]
    W = [[W0[i][k] / cs[k] for k in range(len(cs))] for i in range(len(W0))]       # lnmf.m:59
    Vh = mm(W, H)                                                                  # lnmf.m:62
    m, n = len(V), len(V[0])
    num = mm(ew(lambda v, s: v / s, V, Vh), mt(H))
    den = mm(ones(m, n), mt(H))
    W = ew(lambda w, q: w * q, W, ew(lambda a, b: a / b, num, den))                # lnmf.m:69
    cs = [sum(W[i][k] for i in range(m)) for k in range(len(cs))]
    W = [[W[i][k] / cs[k] for k in range(len(cs))] for i in range(m)]              # lnmf.m:70
    Vh = mm(W, H)                                                                  # lnmf.m:71
    H2 = ew(lambda h, g: h * g, H, mm(mt(W), ew(lambda v, s: v / s, V, Vh)))       # the radicand of lnmf.m:76, exact
    Hf = np.sqrt(tofloat(H2))
    # V_hat = W*H with the irrational H: float64 from here (lnmf.m:77,81)
    Wf, Vf = tofloat(W), tofloat(V)
    Vhf = Wf @ Hf
    cost = float(np.sum(Vf * np.log(Vf / Vhf) - Vf + Vhf))
    return W, Wf, Hf, cost


def pin_lnmf(impl, tol, cost_tol):
    V, W0, H0 = fmat(L_V), fmat(L_W0), fmat(L_H0)
    W, Wf, Hf, cost = _lnmf_exact(V, W0, H0)
    Vn, W0n, H0n = (np.array(a, dtype=np.float64) for a in (L_V, L_W0, L_H0))
    Wo, Ho, c = impl.lnmf(Vn, 2, dict(W_init=W0n, H_init=H0n, maxiter=1))
    assert len(c) == 1 and abs(c[0] - cost) <= cost_tol * abs(cost), (c, cost)
    assert rel(Wo, Wf) <= tol and rel(Ho, Hf) <= tol, (rel(Wo, Wf), rel(Ho, Hf))
    assert abs(np.asarray(Wo).sum(0) - 1).max() <= 10 * tol                        # L1-normalised columns (lnmf.m:70)
    # L2: nothing moves -> identical costs -> `<=` stops at iteration 2, cost vector NOT trimmed
    cs = [Fr(4), Fr(8)]
    Wn = [[W0[i][k] / cs[k] for k in range(2)] for i in range(3)]
    c0 = cost_exact_float(V, mm(Wn, H0), "kl")
    Wo, Ho, c = impl.lnmf(Vn, 2, dict(W_init=W0n, H_init=H0n, W_fixed=True, H_fixed=True, maxiter=5, tolerance=1e-9))
    assert len(c) == 5, c
    assert abs(c[0] - c0) <= cost_tol * c0 and c[1] == c[0] and np.all(c[2:] == 0), (c, c0)
    assert rel(Wo, tofloat(Wn)) <= max(tol * 1e-2, 1e-15) and rel(Ho, H0n) <= 1e-7


# ---------------------------------------------------------------------------------------------------------------------
# KAT C1 -- constrainednmf.m:147-177,213-267: label bookkeeping + the Z update, exact rationals.
#   labels = [7 -1 7 3]: one unlabelled sample and classes {3, 7}, class 7 with TWO members.
#   :151-154  unique -> [-1 3 7]; processed = [2 -1 2 1];  :163 stable sort -> sorted_idx = [2 4 1 3] (1-based), V = V(:, sorted_idx)
#   :166-170  A = [1 0 0 0; 0 1 0 0; 0 0 1 1]  (unlabelled first, then class 1 = label 3, class 2 = label 7)
#   :174-177  Z is K x 3, H = Z*A: the two members of class 7 share one column of Z
#   :215-216  (euclidean) Z <- Z .* ((W'*V*A') ./ max(W'*V_hat*A' + Z_sparsity, eps)); kl: (W'*(V./V_hat)*A') ./ (W'*ones*A' + ...)
#   :251      cost += Z_sparsity * sum(|Z|) -- on Z, not on H
#   :263-267  A's columns go back to the original sample order, H = Z*A
# W is fixed (its update is nmf.m's W step, pinned in tests/pins.py); W_init columns are Pythagorean so line 145 stays rational.
# ---------------------------------------------------------------------------------------------------------------------
C_V = [[2, 1, 3, 5], [4, 6, 1, 2], [3, 2, 7, 1]]
C_LABELS = [7, -1, 7, 3]
C_W0 = [[2, 1], [1, 2], [2, 2]]            # column norms 3, 3
C_Z0 = [[1, 2, 3], [2, 1, 1]]
C_SORTED = [1, 3, 0, 2]                    # 0-based sorted_idx
C_A = [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 1]]


def _constrained_exact(div, lam):
    V = fmat([[C_V[i][j] for j in C_SORTED] for i in range(3)])        # constrainednmf.m:164
    W = [[Fr(x, 3) for x in r] for r in C_W0]                          # constrainednmf.m:145
    A, Z = fmat(C_A), fmat(C_Z0)
    H = mm(Z, A)                                                       # :177
    Vh = mm(W, H)                                                      # :179, :210
    if div == "euclidean":
        neg, pos = mm(mm(mt(W), V), mt(A)), mm(mm(mt(W), Vh), mt(A))   # :215-216
    else:
        neg, pos = mm(mm(mt(W), ew(lambda v, s: v / s, V, Vh)), mt(A)), mm(mm(mt(W), ones(3, 4)), mt(A))   # :218-219
    Z = ew(lambda z, q: z * q, Z, ew(lambda a, b: a / (b + lam), neg, pos))   # :235
    H = mm(Z, A)                                                       # :237
    cost = cost_exact_float(V, mm(W, H), "kl" if div == "kl" else "euclidean") + float(lam) * float(sum(sum(r) for r in Z))   # :241-251
    Horig = [[None] * 4 for _ in range(2)]
    Aorig = [[None] * 4 for _ in range(3)]
    for s in range(4):                                                 # :263-266: A(:, sorted_idx(samp)) = A_temp(:, samp)
        for k in range(2):
            Horig[k][C_SORTED[s]] = H[k][s]
        for c in range(3):
            Aorig[c][C_SORTED[s]] = A[c][s]
    return tofloat(W), tofloat(Horig), tofloat(Z), tofloat(Aorig), cost


def pin_constrainednmf(impl, tol, cost_tol):
    for div in ("euclidean", "kl"):
        for lam in (Fr(0), Fr(1, 4)):
            We, He, Ze, Ae, ce = _constrained_exact(div, lam)
            cfg = dict(divergence=div, W_init=np.array(C_W0, dtype=np.float64), Z_init=np.array(C_Z0, dtype=np.float64), W_fixed=True,
                       Z_sparsity=float(lam), maxiter=1)
            W, H, Z, A, c = impl.constrainednmf(np.array(C_V, dtype=np.float64), np.array(C_LABELS), 2, cfg)
            assert np.array_equal(np.asarray(A), Ae), A
            assert rel(Z, Ze) <= tol and rel(H, He) <= tol, (div, lam, rel(Z, Ze), rel(H, He))
            assert rel(W, We) <= max(tol * 1e-2, 1e-15)
            assert len(c) == 1 and abs(c[0] - ce) <= cost_tol * abs(ce), (div, lam, c, ce)
            assert np.array_equal(np.asarray(H)[:, 0], np.asarray(H)[:, 2])       # the two members of class 7 share their encoding


# ---------------------------------------------------------------------------------------------------------------------
# KAT X1 -- cnmfsc MU H step divides by (positive_grad + eps), NOT max(positive_grad, eps)  (cnmfsc.m:202 vs nmfsc.m:182).
#   Only visible when positive_grad is of the order of eps: W_init = 2^-27 * integers makes W0'*V_hat ~ 2^-54 * O(100) ~ eps * O(10),
#   where `pos + eps` and `max(pos, eps)` differ by 5-30 % per element.  H_fixed = false, W_fixed = true, no sparsity, T = 2.
#   The row norms of line 205 then rescale H and W0 (lines 206-209); W itself (returned) is rescaled through W0 = W ... no:
#   with W_fixed the returned W is the UNscaled W (line 266 copies W over W0, dropping the rescale) -- also pinned here.
# KAT X2 -- cnmfsc sparse-W line search compares against the SHIFT-LESS product Wnew*H (cnmfsc.m:235: a 2-D Wnew takes RFD.m:30-31)
#   and, for t > 1, against the V_hat the previous slice's last try left behind (line 218).  Expected tries per slice, W, cost and
#   the early return (if any) from cnmfsc_hp; m = 4, W_sparsity = 1/4, T = 2.
# KAT X3 -- cnmfsc W by the MU rule with the incremental V_hat = max(V_hat + (W_t - W0_t)*H_shifted, 0) of line 262, H sparse (n = 4).
# ---------------------------------------------------------------------------------------------------------------------
X1 = dict(V=[[4, 1, 2, 1, 3], [1, 3, 1, 2, 2], [2, 2, 4, 1, 1]], W0=[[[3, 1], [1, 2], [2, 1]], [[1, 2], [2, 1], [1, 3]]],
          H0=[[1, 2, 1, 3, 2], [3, 1, 2, 1, 1]], scale=2.0 ** -27, K=2)
X2 = dict(V=[[4, 2, 1, 3, 1], [2, 1, 1, 2, 4], [1, 1, 4, 1, 2], [3, 2, 1, 4, 1]], W0=[[[5, 1], [3, 2], [3, 5], [4, 1]], [[3, 2], [2, 1], [5, 1], [2, 2]]],
          H0=[[2, 1, 1, 2, 3], [3, 2, 1, 2, 1]], K=2)                      # tries per (iteration, t): [1 5 1 1], no early return
X2B = dict(V=[[4, 1, 1], [1, 4, 2], [1, 3, 1], [1, 1, 4]], W0=[[[3], [5], [3], [1]], [[1], [5], [5], [5]]], H0=[[3, 1, 1]], K=1)   # [1 665]: returns in iteration 1
X4 = dict(V=X1["V"], W0=X1["W0"], H0=X1["H0"], K=2)     # no sparsity: MU H step, W0 rescaled by the row norms (cnmfsc.m:205-209), MU W step from that W0
X3 = dict(V=[[4, 2, 2, 1], [4, 4, 4, 2]], W0=[[[2, 3], [2, 2]], [[5, 5], [3, 2]]], H0=[[3, 4, 3, 2], [3, 1, 3, 4]], K=2)           # H tries [1 2]


def _cnmfsc_call(impl, k, cfg, scale=1.0):
    W0 = np.stack([np.array(w, dtype=np.float64) for w in k["W0"]], axis=2) * scale
    info = {}
    c = dict(W_init=W0, H_init=np.array(k["H0"], dtype=np.float64))
    c.update(cfg)
    W, H, cost = impl.cnmfsc(np.array(k["V"], dtype=np.float64), k["K"], len(k["W0"]), c, info=info)
    return W, H, cost, info


def pin_cnmfsc(impl, tol, cost_tol):
    # X1: pos + eps
    k = X1
    sc = D(2) ** -27
    W0l = [dew(lambda x: x * sc, dmat(w)) for w in k["W0"]]
    e = cnmfsc_hp(k["V"], W0l, k["H0"], D(0), D(0), 1, W_fixed=True)
    W, H, c, info = _cnmfsc_call(impl, k, dict(W_fixed=True, maxiter=1), scale=k["scale"])
    ce = np.array([float(x) for x in e["cost"]])
    assert rel(H, dfloat(e["H"])) <= tol, ("X1 H", rel(H, dfloat(e["H"])))
    assert rel(W, _w3(e["W"])) <= tol, ("X1 W", rel(W, _w3(e["W"])))
    assert len(c) == 2 and np.all(np.abs(c - ce) <= cost_tol * np.abs(ce)), (c, ce)
    # X2: shift-less line search; X2B: it gives up in the second slice of iteration 1 -> `cost = cost(1:iter); return`
    for k, want_tries, want_early in ((X2, [1, 5, 1, 1], False), (X2B, [1, TRIES_TO_UNDERFLOW], True)):
        e = cnmfsc_hp(k["V"], k["W0"], k["H0"], D("0.25"), D(0), 2, tol=D("1e-300"), H_fixed=True)
        assert e["triesW"] == want_tries and e["early"] == want_early and min(e["margins"]) > D("0.01")
        W, H, c, info = _cnmfsc_call(impl, k, dict(W_sparsity=0.25, H_fixed=True, maxiter=2, tolerance=1e-300))
        ce = np.array([float(x) for x in e["cost"]])
        assert info["triesW"] == e["triesW"], (info, e["triesW"])
        assert bool(info["converged_early"]) == e["early"]
        assert len(c) == len(ce) and np.all(np.abs(c - ce) <= cost_tol * np.abs(ce)), (c, ce)     # X2B: ONE entry (trimmed to cost(1:1))
        assert rel(W, _w3(e["W"])) <= tol and rel(H, dfloat(e["H"])) <= tol, ("X2", rel(W, _w3(e["W"])), rel(H, dfloat(e["H"])))
    # X4: both MU branches, two iterations
    k = X4
    e = cnmfsc_hp(k["V"], k["W0"], k["H0"], D(0), D(0), 2, tol=D("1e-300"))
    W, H, c, info = _cnmfsc_call(impl, k, dict(maxiter=2, tolerance=1e-300))
    ce = np.array([float(x) for x in e["cost"]])
    assert len(c) == 3 and np.all(np.abs(c - ce) <= cost_tol * np.abs(ce)), (c, ce)
    assert rel(W, _w3(e["W"])) <= tol and rel(H, dfloat(e["H"])) <= tol, ("X4", rel(W, _w3(e["W"])), rel(H, dfloat(e["H"])))
    # X3: sparse H line search on W0 + the incremental V_hat of the MU W step
    k = X3
    e = cnmfsc_hp(k["V"], k["W0"], k["H0"], D(0), D("0.25"), 2, tol=D("1e-300"))
    assert e["triesH"] == [1, 2] and min(e["margins"]) > D("0.01")
    W, H, c, info = _cnmfsc_call(impl, k, dict(H_sparsity=0.25, maxiter=2, tolerance=1e-300))
    ce = np.array([float(x) for x in e["cost"]])
    assert info["triesH"] == e["triesH"], (info, e["triesH"])
    assert len(c) == len(ce) and np.all(np.abs(c - ce) <= cost_tol * np.abs(ce)), (c, ce)
    assert rel(W, _w3(e["W"])) <= tol and rel(H, dfloat(e["H"])) <= tol, ("X3", rel(W, _w3(e["W"])), rel(H, dfloat(e["H"])))


# ---------------------------------------------------------------------------------------------------------------------
# self-check: the transcriptions reproduce the hand derivation of S1 and the designed branch pattern of S3-S5 with margins
# ---------------------------------------------------------------------------------------------------------------------
def selfcheck():
    e = nmfsc_hp(S1["V"], S1["W0"], S1["H0"], D("0.25"), D(0), 3)
    We, He, ce = _s1_expected()
    assert e["early"] and len(e["cost"]) == 1 and abs(float(e["cost"][0]) - ce[0]) < 1e-15 and e["triesW"] == [TRIES_TO_UNDERFLOW] and e["triesH"] == []
    assert rel(dfloat(e["W"]), We) < 1e-15 and rel(dfloat(e["H"]), He) < 1e-15
    assert abs(float(e["stepW"]) / 2.0 ** -665 - 1) < 1e-12
    for k in (S3, S4, S4B):
        e = _run_hp(k, 1)
        assert min(e["margins"]) > D("0.01")
        t = e["triesH"] if k["sH"] else e["triesW"]
        assert t == [2], (k, t)
        st = e["stepH"] if k["sH"] else e["stepW"]
        assert abs(float(st) - 0.6) < 1e-15
    e = _run_hp(S5, 2)
    assert e["triesH"] == [1, 2] and e["triesW"] == [1, 2] and not e["early"] and min(e["margins"]) > D("0.01")
    # L2 / C1 consistency of the hand-written bookkeeping
    assert [C_LABELS[i] for i in C_SORTED] == [-1, 3, 7, 7]
