"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- parity unpinned by the reference.

Float64 NumPy restatement of the reference's multiplicative-update hot path, written
line-against-line with the MATLAB sources under /root/reference (cited per function).
It deliberately keeps the reference's *literal* operation list (eight m*n*K GEMMs per
nmf iteration, GEMMs against ones(m,n), diagonal-extraction GEMM chains, `.^` inside
cnmf's t loop) so that it doubles as the "reference CPU path" timed by bench.py.

MATLAB semantics reproduced here:
  * eps = 2**-52; max(x, eps) ignores NaN  -> np.fmax
  * x.^0 == 1 (also 0.^0, NaN.^0)          -> np.power does the same for 0**0; NaN**0 == 1 too
  * cell2mat on 1xS cell -> hstack (W, also 3-D along dim 2); on Sx1 cell -> vstack (H)
  * rand(m,K,1) is a matrix  -> cnmf with T == 1 uses the W*H branch of RFD
  * real(sqrt(negative)) == 0 in projfunc
  * the stop rule is evaluated from iter 2 on, strict decrease AND decrease < tolerance
Arrays follow MATLAB shapes: V (m,n), W (m,K) or (m,K,T), H (K,n).  Values only -- the
memory order of the ndarray is irrelevant here.
"""
from __future__ import annotations

import numpy as np

EPS = 2.0 ** -52


# --------------------------------------------------------------------------------------
# ReconstructFromDecomposition.m:23-39
# --------------------------------------------------------------------------------------
def reconstruct_from_decomposition(W, H):
    if isinstance(W, (list, tuple)):                      # RFD.m:23-25  cell2mat(1xS) -> dim 2
        W = np.concatenate([np.asarray(w, dtype=np.float64) for w in W], axis=1)
    if isinstance(H, (list, tuple)):                      # RFD.m:26-28  cell2mat(Sx1) -> dim 1
        H = np.concatenate([np.asarray(h, dtype=np.float64) for h in H], axis=0)
    W = np.asarray(W, dtype=np.float64)
    H = np.asarray(H, dtype=np.float64)
    if W.ndim == 2:                                       # RFD.m:30-31
        return W @ H
    m, K, T = W.shape                                     # RFD.m:32-38
    n = H.shape[1]
    V_hat = np.zeros((m, n))
    for t in range(1, T + 1):
        Hs = np.concatenate([np.zeros((K, t - 1)), H[:, : n - t + 1]], axis=1)
        V_hat = V_hat + np.ascontiguousarray(W[:, :, t - 1]) @ Hs   # contiguous copy (what MATLAB's W(:,:,t) is): a strided slice would bypass BLAS
    return V_hat


# --------------------------------------------------------------------------------------
# projfunc.m:13-65 (P. Hoyer 2004)
# --------------------------------------------------------------------------------------
def projfunc(s, k1, k2, nn=True):
    s = np.asarray(s, dtype=np.float64).reshape(-1).copy()
    N = s.size                                            # projfunc.m:13
    if not nn:                                            # projfunc.m:16-19
        isneg = s < 0
        s = np.abs(s)
    v = s + (k1 - s.sum()) / N                            # projfunc.m:22
    zerocoeff = np.zeros(0, dtype=np.int64)               # projfunc.m:25
    j = 0
    while True:
        midpoint = np.ones(N) * k1 / (N - zerocoeff.size)  # projfunc.m:31
        midpoint[zerocoeff] = 0.0                         # projfunc.m:32
        w = v - midpoint                                  # projfunc.m:33
        a = np.sum(w ** 2)                                # projfunc.m:34
        b = 2.0 * np.dot(w, v)                            # projfunc.m:35
        c = np.sum(v ** 2) - k2                           # projfunc.m:36
        disc = b * b - 4.0 * a * c
        sq = np.sqrt(disc) if disc >= 0 else 0.0          # real(sqrt(neg)) == 0, projfunc.m:37
        with np.errstate(divide="ignore", invalid="ignore"):
            alphap = (-b + sq) / (2.0 * a)
        v = alphap * w + v                                # projfunc.m:38
        if np.all(v >= 0):                                # projfunc.m:40-44
            usediters = j + 1
            break
        j += 1                                            # projfunc.m:46
        zerocoeff = np.flatnonzero(v <= 0)                # projfunc.m:49
        v[zerocoeff] = 0.0                                # projfunc.m:50
        tempsum = v.sum()                                 # projfunc.m:51
        v = v + (k1 - tempsum) / (N - zerocoeff.size)     # projfunc.m:52
        v[zerocoeff] = 0.0                                # projfunc.m:53
    if not nn:                                            # projfunc.m:58-60
        v = (-2.0 * isneg + 1.0) * v
    return v, usediters


# --------------------------------------------------------------------------------------
# helpers shared by the nmf / cnmf local ValidateParameters (nmf.m:238-413, cnmf.m:271-449)
# --------------------------------------------------------------------------------------
def _is_cell(x):
    return isinstance(x, (list, tuple))


def _per_source(config, name, S, default, clamp):
    """nmf.m:312-401 / cnmf.m:348-437: scalar or 1-cell broadcast, S-cell kept, else error."""
    val = config.get(name, None)
    if val is None or (_is_cell(val) and len(val) == 0):
        return [default] * S
    if _is_cell(val) and len(val) > 1 and len(val) != S:
        kind = "sparsity levels" if name.endswith("sparsity") else "update switches"
        raise ValueError("Requested %d sources. Given %d %s." % (S, len(val), kind))
    if (not _is_cell(val)) or len(val) == 1:
        t = val[0] if _is_cell(val) else val
        t = clamp(t)
        return [t] * S
    return [clamp(t) for t in val]


def _validate_common(V, Ks, config, rng):
    cfg = dict(config) if config is not None else {}
    S = len(Ks)
    cfg.setdefault("divergence", "euclidean")             # nmf.m:250-252
    is_ab = cfg["divergence"] in ("ab_divergence", "ab")
    if ("alpha" not in cfg) or not is_ab:                 # nmf.m:255-259
        cfg["alpha"] = 1.0
    if ("beta" not in cfg) or not is_ab:                  # nmf.m:262-266
        cfg["beta"] = 1.0
    clampnn = lambda x: max(float(x), 0.0)
    cfg["W_sparsity"] = _per_source(cfg, "W_sparsity", S, 0.0, clampnn)
    cfg["H_sparsity"] = _per_source(cfg, "H_sparsity", S, 0.0, clampnn)
    cfg["W_fixed"] = _per_source(cfg, "W_fixed", S, False, bool)
    cfg["H_fixed"] = _per_source(cfg, "H_fixed", S, False, bool)
    if ("maxiter" not in cfg) or cfg["maxiter"] <= 0:     # nmf.m:404-406
        cfg["maxiter"] = 100
    if ("tolerance" not in cfg) or cfg["tolerance"] <= 0:  # nmf.m:409-411
        cfg["tolerance"] = 1e-3
    return cfg


def _init_H(cfg, Ks, n, rng):
    S = len(Ks)
    Hi = cfg.get("H_init", None)
    if Hi is None or (hasattr(Hi, "__len__") and len(Hi) == 0):   # nmf.m:269-278
        is_cell = S != 1
        H = [np.fmax(rng.rand(K, n), EPS) for K in Ks]
    elif _is_cell(Hi) and len(Hi) != S:                   # nmf.m:279-280
        raise ValueError("Requested %d sources. Given %d initial encoding matrices." % (S, len(Hi)))
    elif not _is_cell(Hi):                                # nmf.m:281-283
        is_cell = False
        H = [np.array(Hi, dtype=np.float64)]
    else:                                                 # nmf.m:284-287
        is_cell = True
        H = [np.array(h, dtype=np.float64) for h in Hi]
    return H, is_cell


# --------------------------------------------------------------------------------------
# nmf.m:108-236
# --------------------------------------------------------------------------------------
def _nmf_validate(V, Ks, config, rng):
    m, n = V.shape
    S = len(Ks)
    cfg = _validate_common(V, Ks, config, rng)
    H, is_H_cell = _init_H(cfg, Ks, n, rng)
    Wi = cfg.get("W_init", None)
    if Wi is None or (hasattr(Wi, "__len__") and len(Wi) == 0):   # nmf.m:290-300
        is_W_cell = S != 1
        W = []
        for K in Ks:
            w = np.fmax(rng.rand(m, K), EPS)
            w = w * (1.0 / np.sqrt(np.sum(w ** 2, axis=0)))[None, :]
            W.append(w)
    elif _is_cell(Wi) and len(Wi) != S:                   # nmf.m:301-302
        raise ValueError("Requested %d sources. Given %d initial basis matrices." % (S, len(Wi)))
    elif not _is_cell(Wi):                                # nmf.m:303-305
        is_W_cell = False
        W = [np.array(Wi, dtype=np.float64)]
    else:                                                 # nmf.m:306-309
        is_W_cell = True
        W = [np.array(w, dtype=np.float64) for w in Wi]
    return cfg, W, H, is_W_cell, is_H_cell


def _col_normalize(Ws):
    return Ws * (1.0 / np.sqrt(np.sum(Ws ** 2, axis=0)))[None, :]   # W * diag(1 ./ sqrt(sum(W.^2,1)))


def _ddiag(M):
    """diag(diag(M)) as a vector (X * diag(d) == X * d[None, :])."""
    return np.diag(M).copy()


def _cost(div, V, V_hat, alpha, beta):
    with np.errstate(divide="ignore", invalid="ignore"):
        if div == "euclidean":                            # nmf.m:207-208
            return 0.5 * np.sum((V - V_hat) ** 2)
        if div in ("kl_divergence", "kl"):                # nmf.m:209-210
            return np.sum(V * np.log(V / V_hat) - V + V_hat)
        if div in ("is_divergence", "is"):                # nmf.m:211-212
            return np.sum(np.log(V_hat / V) + (V / V_hat) - 1.0)
        if div in ("ab_divergence", "ab"):                # nmf.m:213-214
            a, b = np.float64(alpha), np.float64(beta)     # IEEE division like MATLAB: alpha*beta == 0 gives -Inf, not an exception
            return (np.float64(-1.0) / (a * b)) * np.sum(V ** a * V_hat ** b - (a * V ** (a + b) + b * V_hat ** (a + b) + b) / (a + b))
    return None


def nmf(V, num_basis_elems, config=None, rng=None, trace=None):
    """nmf.m:1.  Returns (W, H, cost).  W/H are lists iff the corresponding init was a cell or S>1.

    `trace`, if a list, receives (W_all, H_all) copies after every iteration (test aid only).
    """
    V = np.asarray(V, dtype=np.float64)
    m, n = V.shape                                        # nmf.m:113
    Ks = list(num_basis_elems) if _is_cell(num_basis_elems) else [num_basis_elems]  # nmf.m:114-117
    Ks = [int(k) for k in Ks]
    S = len(Ks)
    rng = rng if rng is not None else np.random.RandomState(0)
    cfg, W, H, is_W_cell, is_H_cell = _nmf_validate(V, Ks, config, rng)   # nmf.m:118
    div, alpha, beta = cfg["divergence"], float(cfg["alpha"]), float(cfg["beta"])
    if div in ("ab_divergence", "ab") and alpha == 0 and beta == 0:       # nmf.m:120-122
        raise ValueError("alpha = 0 and beta = 0 is not supported at this time.")
    use_dual = alpha == 0                                 # nmf.m:124-128
    W = [_col_normalize(w) for w in W]                    # nmf.m:130-134
    W_all = np.concatenate(W, axis=1)                     # nmf.m:136
    H_all = np.concatenate(H, axis=0)                     # nmf.m:137
    V_hat = reconstruct_from_decomposition(W_all, H_all)  # nmf.m:139
    maxiter = int(cfg["maxiter"])
    cost = np.zeros(maxiter)                              # nmf.m:141
    ones_mn = np.ones((m, n))
    ones_nm = np.ones((n, m))
    n_run = maxiter
    with np.errstate(divide="ignore", invalid="ignore"):
        for it in range(1, maxiter + 1):                  # nmf.m:143
            for s in range(S):                            # nmf.m:145
                if cfg["W_fixed"][s]:
                    continue
                Ws, Hs = W[s], H[s]
                if div == "euclidean":                    # nmf.m:148-150
                    neg = V @ Hs.T + Ws * _ddiag(Hs @ V_hat.T @ Ws)[None, :]
                    pos = V_hat @ Hs.T + Ws * _ddiag(Hs @ V.T @ Ws)[None, :]
                elif div in ("kl_divergence", "kl"):      # nmf.m:151-153
                    neg = (V / V_hat) @ Hs.T + Ws * _ddiag(Hs @ ones_nm @ Ws)[None, :]
                    pos = ones_mn @ Hs.T + Ws * _ddiag(Hs @ (V.T / V_hat.T) @ Ws)[None, :]
                elif div in ("is_divergence", "is"):      # nmf.m:154-156
                    neg = (V / V_hat ** 2) @ Hs.T + Ws * _ddiag(Hs @ (ones_nm / V_hat.T) @ Ws)[None, :]
                    pos = (ones_mn / V_hat) @ Hs.T + Ws * _ddiag(Hs @ (V.T / V_hat.T ** 2) @ Ws)[None, :]
                elif div in ("ab_divergence", "ab"):      # nmf.m:157-164
                    if use_dual:
                        neg = ((V ** (alpha - 1) * V_hat ** beta) @ Hs.T + Ws * _ddiag(Hs @ V.T ** (alpha + beta - 1) @ Ws)[None, :]) ** (1 / beta)
                        pos = (V ** (alpha + beta - 1) @ Hs.T + Ws * _ddiag(Hs @ (V ** (alpha - 1) * V_hat ** beta).T @ Ws)[None, :]) ** (1 / beta)
                    else:
                        neg = ((V ** alpha * V_hat ** (beta - 1)) @ Hs.T + Ws * _ddiag(Hs @ V_hat.T ** (alpha + beta - 1) @ Ws)[None, :]) ** (1 / alpha)
                        pos = (V_hat ** (alpha + beta - 1) @ Hs.T + Ws * _ddiag(Hs @ (V ** alpha * V_hat ** (beta - 1)).T @ Ws)[None, :]) ** (1 / alpha)
                else:                                     # nmf.m:165-166
                    raise ValueError("No update equations defined for cost function with divergence type " + str(div))
                Ws = Ws * (neg / np.fmax(pos + cfg["W_sparsity"][s], EPS))   # nmf.m:168
                W[s] = _col_normalize(Ws)                 # nmf.m:169
            W_all = np.concatenate(W, axis=1)             # nmf.m:172
            V_hat = reconstruct_from_decomposition(W_all, H_all)   # nmf.m:173
            for s in range(S):                            # nmf.m:176
                if cfg["H_fixed"][s]:
                    continue
                Ws, Hs = W[s], H[s]
                if div == "euclidean":                    # nmf.m:179-181
                    neg = Ws.T @ V
                    pos = Ws.T @ V_hat
                elif div in ("kl_divergence", "kl"):      # nmf.m:182-184
                    neg = Ws.T @ (V / V_hat)
                    pos = Ws.T @ ones_mn
                elif div in ("is_divergence", "is"):      # nmf.m:185-187
                    neg = Ws.T @ (V / V_hat ** 2)
                    pos = Ws.T @ (ones_mn / V_hat)
                elif div in ("ab_divergence", "ab"):      # nmf.m:188-195
                    if use_dual:
                        neg = (Ws.T @ (V ** (alpha - 1) * V_hat ** beta)) ** (1 / beta)
                        pos = (Ws.T @ V ** (alpha + beta - 1)) ** (1 / beta)
                    else:
                        neg = (Ws.T @ (V ** alpha * V_hat ** (beta - 1))) ** (1 / alpha)
                        pos = (Ws.T @ V_hat ** (alpha + beta - 1)) ** (1 / alpha)
                else:
                    raise ValueError("No update equations defined for cost function with divergence type " + str(div))
                H[s] = Hs * (neg / np.fmax(pos + cfg["H_sparsity"][s], EPS))   # nmf.m:199
            H_all = np.concatenate(H, axis=0)             # nmf.m:202
            V_hat = reconstruct_from_decomposition(W_all, H_all)   # nmf.m:203
            c = _cost(div, V, V_hat, alpha, beta)         # nmf.m:206-215
            for s in range(S):                            # nmf.m:216-218
                c = c + cfg["W_sparsity"][s] * np.sum(np.abs(W[s])) + cfg["H_sparsity"][s] * np.sum(np.abs(H[s]))
            cost[it - 1] = c
            if trace is not None:
                trace.append((W_all.copy(), H_all.copy()))
            if it > 1 and cost[it - 1] < cost[it - 2] and cost[it - 2] - cost[it - 1] < cfg["tolerance"]:   # nmf.m:221-224
                n_run = it
                break
    cost = cost[:n_run]
    Wout = W if is_W_cell else W[0]                       # nmf.m:228-234
    Hout = H if is_H_cell else H[0]
    return Wout, Hout, cost


# --------------------------------------------------------------------------------------
# cnmf.m:121-269
# --------------------------------------------------------------------------------------
def _slab_norms(Ws, T):
    """norm(squeeze(W(:,k,:)),'fro') / T for every k (cnmf.m:162, 197)."""
    if Ws.ndim == 2:
        return np.sqrt(np.sum(Ws ** 2, axis=0)) / T
    return np.sqrt(np.sum(Ws ** 2, axis=(0, 2))) / T


def _cnmf_validate(V, Ks, T, config, rng):
    m, n = V.shape
    S = len(Ks)
    cfg = _validate_common(V, Ks, config, rng)
    H, is_H_cell = _init_H(cfg, Ks, n, rng)               # cnmf.m:302-320
    Wi = cfg.get("W_init", None)
    if Wi is None or (hasattr(Wi, "__len__") and len(Wi) == 0):   # cnmf.m:323-336
        is_W_cell = S != 1
        W = []
        for K in Ks:
            w = rng.rand(m, K, T)
            w = w / _slab_norms(w, T)[None, :, None]
            W.append(w)
    elif _is_cell(Wi) and len(Wi) != S:                   # cnmf.m:337-338
        raise ValueError("Requested %d sources. Given %d initial basis matrices." % (S, len(Wi)))
    elif not _is_cell(Wi):                                # cnmf.m:339-341
        is_W_cell = False
        W = [np.array(Wi, dtype=np.float64)]
    else:                                                 # cnmf.m:342-345
        is_W_cell = True
        W = [np.array(w, dtype=np.float64) for w in Wi]
    # MATLAB arrays carry trailing singleton dims implicitly: always work on (m,K,T)
    W = [w.reshape(w.shape[0], w.shape[1], -1) for w in W]
    for w in W:
        if w.shape[2] != T:
            raise ValueError("W_init context length %d != context_len %d" % (w.shape[2], T))
    return cfg, W, H, is_W_cell, is_H_cell


def _pw(x, p):
    """MATLAB x.^p for real scalar p; x.^0 == 1 everywhere (also NaN, 0)."""
    if p == 0:
        return np.ones_like(x)
    if p == 1:
        return x
    return np.power(x, p)


def _rshift(Hs, t, n):
    """[zeros(K,t-1) H(:,1:n-t+1)]  (cnmf.m:188, RFD.m:37)."""
    K = Hs.shape[0]
    return np.concatenate([np.zeros((K, t - 1)), Hs[:, : n - t + 1]], axis=1)


def _lshift(X, t, n):
    """[X(:,t:n) zeros(m,t-1)]  (cnmf.m:219-223)."""
    m = X.shape[0]
    return np.concatenate([X[:, t - 1: n], np.zeros((m, t - 1))], axis=1)


def cnmf(V, num_basis_elems, context_len, config=None, rng=None):
    """cnmf.m:1.  W is (m,K,T) (a (m,K,1) tensor is returned as (m,K) like MATLAB does)."""
    V = np.asarray(V, dtype=np.float64)
    m, n = V.shape                                        # cnmf.m:126
    Ks = list(num_basis_elems) if _is_cell(num_basis_elems) else [num_basis_elems]
    Ks = [int(k) for k in Ks]
    S = len(Ks)
    T = int(context_len)
    rng = rng if rng is not None else np.random.RandomState(0)
    cfg, W, H, is_W_cell, is_H_cell = _cnmf_validate(V, Ks, T, config, rng)   # cnmf.m:131
    div = cfg["divergence"]
    alpha, beta = float(cfg["alpha"]), float(cfg["beta"])
    if div in ("ab_divergence", "ab") and alpha == 0 and beta == 0:           # cnmf.m:133-135
        raise ValueError("alpha = 0 and beta = 0 is not supported at this time.")
    if div in ("euclidean", "frobenius"):                 # cnmf.m:137-147
        alpha, beta = 1.0, 1.0
    elif div in ("kl_divergence", "kl"):
        alpha, beta = 1.0, 0.0
    elif div in ("is_divergence", "is"):
        alpha, beta = 1.0, -1.0
    use_dual = alpha == 0                                 # cnmf.m:149-153
    is_kl = div in ("kl_divergence", "kl")
    for s in range(S):                                    # cnmf.m:157-166
        w_norm = _slab_norms(W[s], T)
        W[s] = W[s] / w_norm[None, :, None]
        H[s] = w_norm[:, None] * H[s]
    W_all = np.concatenate(W, axis=1)                     # cnmf.m:168
    H_all = np.concatenate(H, axis=0)                     # cnmf.m:169
    rfd = lambda Wa, Ha: reconstruct_from_decomposition(Wa[:, :, 0] if T == 1 else Wa, Ha)
    V_hat = rfd(W_all, H_all)                             # cnmf.m:171
    maxiter = int(cfg["maxiter"])
    cost = np.zeros(maxiter)                              # cnmf.m:173
    n_run = maxiter
    with np.errstate(divide="ignore", invalid="ignore"):
        for it in range(1, maxiter + 1):                  # cnmf.m:175
            for s in range(S):                            # cnmf.m:177
                if cfg["W_fixed"][s]:
                    continue
                lam = cfg["W_sparsity"][s]
                for t in range(1, T + 1):
                    Hsh = _rshift(H[s], t, n)             # cnmf.m:181/188
                    Wt = np.ascontiguousarray(W[s][:, :, t - 1])   # MATLAB's W(:,:,t) is a contiguous copy; a strided view bypasses BLAS
                    if use_dual:                          # cnmf.m:180-185
                        Vn = _pw(V, alpha - 1) * _pw(V_hat, beta)
                        Vp = _pw(V, alpha + beta - 1)
                        ex = 1.0 / beta
                        gneg = _pw(Vn @ Hsh.T + Wt * _ddiag(Hsh @ _pw(V.T, alpha + beta - 1) @ Wt)[None, :], ex)
                        gpos = _pw(Vp @ Hsh.T + Wt * _ddiag(Hsh @ Vn.T @ Wt)[None, :], ex)
                    else:                                 # cnmf.m:187-194
                        Vn = _pw(V, alpha) * _pw(V_hat, beta - 1)
                        Vp = _pw(V_hat, alpha + beta - 1)
                        ex = 1.0 / alpha
                        gneg = _pw(Vn @ Hsh.T + Wt * _ddiag(Hsh @ _pw(V_hat.T, alpha + beta - 1) @ Wt)[None, :], ex)
                        gpos = _pw(Vp @ Hsh.T + Wt * _ddiag(Hsh @ Vn.T @ Wt)[None, :], ex)
                    W[s][:, :, t - 1] = Wt * (gneg / np.fmax(gpos + lam, EPS))   # cnmf.m:184/193
                w_norm = _slab_norms(W[s], T)             # cnmf.m:196-199 (H is NOT rescaled here)
                W[s] = W[s] / w_norm[None, :, None]
            W_all = np.concatenate(W, axis=1)             # cnmf.m:202
            H_all = np.concatenate(H, axis=0)             # cnmf.m:203
            V_hat = rfd(W_all, H_all)                     # cnmf.m:204
            for s in range(S):                            # cnmf.m:207
                if cfg["H_fixed"][s]:
                    continue
                if use_dual:                              # cnmf.m:209-214
                    V_neg = _pw(V, alpha - 1) * _pw(V_hat, beta)
                    V_pos = _pw(V, alpha + beta - 1)
                else:
                    V_neg = _pw(V, alpha) * _pw(V_hat, beta - 1)
                    V_pos = _pw(V_hat, alpha + beta - 1)
                gneg = np.zeros((Ks[s], n))               # cnmf.m:215-216
                gpos = np.zeros((Ks[s], n))
                for t in range(1, T + 1):                 # cnmf.m:217-226
                    Vn_sh = _lshift(V_neg, t, n)
                    Vp_sh = V_pos if is_kl else _lshift(V_pos, t, n)   # cnmf.m:220-224 (KL: unshifted)
                    Wt = np.ascontiguousarray(W[s][:, :, t - 1])   # MATLAB's W(:,:,t) is a contiguous copy; a strided view bypasses BLAS
                    gneg = gneg + Wt.T @ Vn_sh
                    gpos = gpos + Wt.T @ Vp_sh
                ex = (1.0 / beta) if use_dual else (1.0 / alpha)       # cnmf.m:227-231
                H[s] = H[s] * (_pw(gneg, ex) / np.fmax(_pw(gpos, ex) + cfg["H_sparsity"][s], EPS))
            H_all = np.concatenate(H, axis=0)             # cnmf.m:235
            V_hat = rfd(W_all, H_all)                     # cnmf.m:236
            if div == "frobenius":                        # cnmf.m:239-248 has no such case: cost stays 0
                c = 0.0
            else:
                c = _cost(div, V, V_hat, alpha, beta)
                if c is None:                             # unknown string: no `otherwise` in cnmf.m:239-248
                    c = 0.0
            for s in range(S):                            # cnmf.m:249-251
                c = c + cfg["W_sparsity"][s] * np.sum(np.abs(W[s])) + cfg["H_sparsity"][s] * np.sum(np.abs(H[s]))
            cost[it - 1] = c
            if it > 1 and cost[it - 1] < cost[it - 2] and cost[it - 2] - cost[it - 1] < cfg["tolerance"]:   # cnmf.m:254-257
                n_run = it
                break
    cost = cost[:n_run]
    if T == 1:
        W = [w[:, :, 0] for w in W]
    Wout = W if is_W_cell else W[0]                       # cnmf.m:261-267
    Hout = H if is_H_cell else H[0]
    return Wout, Hout, cost


# --------------------------------------------------------------------------------------
# nmfsc.m:57-245
# --------------------------------------------------------------------------------------
def nmfsc(V, num_basis_elems, config=None, rng=None, info=None):
    """nmfsc.m:1.  `info`, if a dict, receives the line-search try counts and final step sizes."""
    V = np.asarray(V, dtype=np.float64)
    if V.min() < 0:                                       # nmfsc.m:57-59
        raise ValueError("Negative values in data!")
    V = V / V.max()                                       # nmfsc.m:62
    m, n = V.shape                                        # nmfsc.m:65
    K = int(num_basis_elems)
    cfg = dict(config) if config is not None else {}
    rng = rng if rng is not None else np.random.RandomState(0)
    if cfg.get("W_init", None) is None or np.size(cfg["W_init"]) == 0:        # nmfsc.m:73-75
        cfg["W_init"] = rng.rand(m, K)
    if cfg.get("H_init", None) is None or np.size(cfg["H_init"]) == 0:        # nmfsc.m:78-81
        h = rng.rand(K, n)
        cfg["H_init"] = (1.0 / np.sqrt(np.sum(h ** 2, axis=1)))[:, None] * h
    W = np.array(cfg["W_init"], dtype=np.float64)         # nmfsc.m:83-84
    H = np.array(cfg["H_init"], dtype=np.float64)
    L1a = L1s = None
    sW = cfg.get("W_sparsity", None)
    if sW is None or np.size(sW) == 0:                    # nmfsc.m:87-97
        sW = 0.0
    elif sW > 0:
        sW = min(float(sW), 1.0)
        L1a = np.sqrt(m) - (np.sqrt(m) - 1) * sW
        for k in range(K):
            W[:, k] = projfunc(W[:, k], L1a, 1.0, True)[0]
    sH = cfg.get("H_sparsity", None)
    if sH is None or np.size(sH) == 0:                    # nmfsc.m:100-110
        sH = 0.0
    elif sH > 0:
        sH = min(float(sH), 1.0)
        L1s = np.sqrt(n) - (np.sqrt(n) - 1) * sH
        for k in range(K):
            H[k, :] = projfunc(H[k, :], L1s, 1.0, True)[0]
    W_fixed = bool(cfg.get("W_fixed", False) or False)    # nmfsc.m:113-120
    H_fixed = bool(cfg.get("H_fixed", False) or False)
    maxiter = cfg.get("maxiter", None)
    if maxiter is None or maxiter <= 0:                   # nmfsc.m:123-125
        maxiter = 100
    maxiter = int(maxiter)
    tol = cfg.get("tolerance", None)
    if tol is None or tol <= 0:                           # nmfsc.m:128-130
        tol = 1e-3
    stepsizeW = 1.0                                       # nmfsc.m:133-134
    stepsizeH = 1.0
    cost = np.zeros(maxiter + 1)                          # nmfsc.m:137
    V_hat = reconstruct_from_decomposition(W, H)          # nmfsc.m:138
    cost[0] = 0.5 * np.sum((V - V_hat) ** 2)              # nmfsc.m:139
    triesH, triesW = [], []
    n_cost = maxiter + 1

    def _finish(ncost, early=False):
        if info is not None:
            info.update(triesH=triesH, triesW=triesW, stepsizeH=stepsizeH, stepsizeW=stepsizeW, converged_early=early)
        return W, H, cost[:ncost]

    for it in range(1, maxiter + 1):                      # nmfsc.m:141
        if not H_fixed:                                   # nmfsc.m:143
            neg = W.T @ V                                 # nmfsc.m:144
            pos = W.T @ V_hat                             # nmfsc.m:145
            if sH > 0:                                    # nmfsc.m:146
                dH = pos - neg                            # nmfsc.m:148
                begobj = cost[it - 1]                     # nmfsc.m:149
                tries = 0
                while True:                               # nmfsc.m:152
                    tries += 1
                    Hnew = H - stepsizeH * dH             # nmfsc.m:154
                    for k in range(K):                    # nmfsc.m:155-157
                        Hnew[k, :] = projfunc(Hnew[k, :], L1s, 1.0, True)[0]
                    V_hat = reconstruct_from_decomposition(W, Hnew)   # nmfsc.m:160
                    newobj = 0.5 * np.sum((V - V_hat) ** 2)           # nmfsc.m:161
                    if newobj <= begobj:                  # nmfsc.m:164-166
                        break
                    stepsizeH = stepsizeH / 2             # nmfsc.m:169
                    if stepsizeH < 1e-200:                # nmfsc.m:170-174
                        triesH.append(tries)
                        return _finish(it, True)
                triesH.append(tries)
                stepsizeH = 1.2 * stepsizeH               # nmfsc.m:178
                H = Hnew                                  # nmfsc.m:179
            else:
                H = H * (neg / np.fmax(pos, EPS))         # nmfsc.m:182
                norms = np.sqrt(np.sum(H ** 2, axis=1))   # nmfsc.m:185
                H = (1.0 / norms)[:, None] * H            # nmfsc.m:186
                W = W * norms[None, :]                    # nmfsc.m:187
        if not W_fixed:                                   # nmfsc.m:192
            V_hat = reconstruct_from_decomposition(W, H)  # nmfsc.m:193
            neg = V @ H.T                                 # nmfsc.m:194
            pos = V_hat @ H.T                             # nmfsc.m:195
            if sW > 0:                                    # nmfsc.m:196
                begobj = 0.5 * np.sum((V - V_hat) ** 2)   # nmfsc.m:197
                dW = pos - neg                            # nmfsc.m:200
                tries = 0
                while True:                               # nmfsc.m:203
                    tries += 1
                    Wnew = W - stepsizeW * dW             # nmfsc.m:205
                    for k in range(K):                    # nmfsc.m:206-208
                        Wnew[:, k] = projfunc(Wnew[:, k], L1a, 1.0, True)[0]
                    V_hat = reconstruct_from_decomposition(Wnew, H)   # nmfsc.m:211
                    newobj = 0.5 * np.sum((V - V_hat) ** 2)           # nmfsc.m:212
                    if newobj <= begobj:                  # nmfsc.m:215-217
                        break
                    stepsizeW = stepsizeW / 2             # nmfsc.m:220
                    if stepsizeW < 1e-200:                # nmfsc.m:221-225
                        triesW.append(tries)
                        return _finish(it, True)
                triesW.append(tries)
                stepsizeW = 1.2 * stepsizeW               # nmfsc.m:228
                W = Wnew                                  # nmfsc.m:229
            else:
                with np.errstate(divide="ignore", invalid="ignore"):
                    W = W * (neg / np.fmax(pos, EPS))     # nmfsc.m:232
        V_hat = reconstruct_from_decomposition(W, H)      # nmfsc.m:237
        cost[it] = 0.5 * np.sum((V - V_hat) ** 2)         # nmfsc.m:238
        if it > 1 and cost[it] < cost[it - 1] and cost[it - 1] - cost[it] < tol:   # nmfsc.m:241-244
            n_cost = it + 1
            break
    return _finish(n_cost)


# --------------------------------------------------------------------------------------
# cnmfsc.m:67-277.  Mirrors the reference's quirks:
#   * the initial projection changes W but NOT W0 (cnmfsc.m:94-112); the H step runs on W0
#   * the MU H-step divides by (positive_grad + eps), not max(., eps) (cnmfsc.m:206)
#   * in the sparse W branch the line search evaluates ReconstructFromDecomposition(Wnew, H) with the 2-D slice Wnew,
#     i.e. the plain product Wnew*H without shifts (cnmfsc.m:238), and V_hat stays that product for the next t
# --------------------------------------------------------------------------------------
def cnmfsc(V, num_basis_elems, context_len, config=None, rng=None, info=None):
    V = np.asarray(V, dtype=np.float64)
    if V.min() < 0:                                       # cnmfsc.m:67-69
        raise ValueError("Negative values in data!")
    V = V / V.max()                                       # cnmfsc.m:72
    m, n = V.shape
    K, T = int(num_basis_elems), int(context_len)
    cfg = dict(config) if config is not None else {}
    rng = rng if rng is not None else np.random.RandomState(0)
    if cfg.get("W_init", None) is None or np.size(cfg["W_init"]) == 0:        # cnmfsc.m:83-85
        cfg["W_init"] = rng.rand(m, K, T)
    if cfg.get("H_init", None) is None or np.size(cfg["H_init"]) == 0:        # cnmfsc.m:88-91
        h = rng.rand(K, n)
        cfg["H_init"] = (1.0 / np.sqrt(np.sum(h ** 2, axis=1)))[:, None] * h
    W0 = np.array(cfg["W_init"], dtype=np.float64).reshape(m, K, T)           # cnmfsc.m:93
    W = W0.copy()                                                             # cnmfsc.m:94
    H = np.array(cfg["H_init"], dtype=np.float64)
    rfd3 = lambda Wx, Hx: reconstruct_from_decomposition(Wx[:, :, 0] if Wx.shape[2] == 1 else Wx, Hx)
    L1a = L1s = None
    sW = cfg.get("W_sparsity", None)
    if sW is None or np.size(sW) == 0:                    # cnmfsc.m:98-111
        sW = 0.0
    elif sW > 0:
        sW = min(float(sW), 1.0)
        L1a = np.sqrt(m) - (np.sqrt(m) - 1) * sW
        for t in range(T):
            for k in range(K):
                W[:, k, t] = projfunc(W[:, k, t], L1a, 1.0, True)[0]
    sH = cfg.get("H_sparsity", None)
    if sH is None or np.size(sH) == 0:                    # cnmfsc.m:114-124
        sH = 0.0
    elif sH > 0:
        sH = min(float(sH), 1.0)
        L1s = np.sqrt(n) - (np.sqrt(n) - 1) * sH
        for k in range(K):
            H[k, :] = projfunc(H[k, :], L1s, 1.0, True)[0]
    W_fixed = bool(cfg.get("W_fixed", False) or False)
    H_fixed = bool(cfg.get("H_fixed", False) or False)
    maxiter = cfg.get("maxiter", None)
    maxiter = 100 if (maxiter is None or maxiter <= 0) else int(maxiter)      # cnmfsc.m:137-139
    tol = cfg.get("tolerance", None)
    tol = 1e-3 if (tol is None or tol <= 0) else float(tol)                   # cnmfsc.m:142-144
    stepsizeW = np.ones(T)                                # cnmfsc.m:147
    stepsizeH = 1.0
    cost = np.zeros(maxiter + 1)
    V_hat = rfd3(W, H)                                    # cnmfsc.m:152
    cost[0] = 0.5 * np.sum((V - V_hat) ** 2)
    triesH, triesW = [], []

    def _finish(ncost, early=False):
        if info is not None:
            info.update(triesH=triesH, triesW=triesW, stepsizeH=stepsizeH, stepsizeW=stepsizeW.copy(), converged_early=early)
        Wout = W[:, :, 0] if T == 1 else W
        return Wout, H, cost[:ncost]

    n_cost = maxiter + 1
    for it in range(1, maxiter + 1):                      # cnmfsc.m:155
        if not H_fixed:                                   # cnmfsc.m:157
            neg = np.zeros((K, n))
            pos = np.zeros((K, n))
            for t in range(1, T + 1):                     # cnmfsc.m:160-165
                neg = neg + np.ascontiguousarray(W0[:, :, t - 1]).T @ _lshift(V, t, n)
                pos = pos + np.ascontiguousarray(W0[:, :, t - 1]).T @ _lshift(V_hat, t, n)
            if sH > 0:
                dH = pos - neg                            # cnmfsc.m:168
                begobj = cost[it - 1]
                tries = 0
                while True:
                    tries += 1
                    Hnew = H - stepsizeH * dH             # cnmfsc.m:174
                    for k in range(K):
                        Hnew[k, :] = projfunc(Hnew[k, :], L1s, 1.0, True)[0]
                    V_hat = rfd3(W0, Hnew)                # cnmfsc.m:180
                    newobj = 0.5 * np.sum((V - V_hat) ** 2)
                    if newobj <= begobj:                  # cnmfsc.m:184
                        break
                    stepsizeH = stepsizeH / 2
                    if stepsizeH < 1e-200:                # cnmfsc.m:190-194
                        triesH.append(tries)
                        return _finish(it, True)
                triesH.append(tries)
                stepsizeH = 1.2 * stepsizeH               # cnmfsc.m:198
                H = Hnew
            else:
                with np.errstate(divide="ignore", invalid="ignore"):
                    H = H * (neg / (pos + EPS))           # cnmfsc.m:202
                norms = np.sqrt(np.sum(H ** 2, axis=1))   # cnmfsc.m:205
                H = (1.0 / norms)[:, None] * H
                for t in range(T):                        # cnmfsc.m:207-209
                    W0[:, :, t] = W0[:, :, t] * norms[None, :]
        if not W_fixed:                                   # cnmfsc.m:214
            V_hat = rfd3(W0, H)                           # cnmfsc.m:215
            if sW > 0:
                for t in range(1, T + 1):
                    begobj = 0.5 * np.sum((V - V_hat) ** 2)               # cnmfsc.m:218
                    Hsh = _rshift(H, t, n)                # cnmfsc.m:221
                    neg = V @ Hsh.T
                    pos = V_hat @ Hsh.T
                    dW = pos - neg                        # cnmfsc.m:224
                    tries = 0
                    while True:
                        tries += 1
                        Wnew = W0[:, :, t - 1] - stepsizeW[t - 1] * dW    # cnmfsc.m:229
                        for k in range(K):
                            Wnew[:, k] = projfunc(Wnew[:, k], L1a, 1.0, True)[0]
                        V_hat = reconstruct_from_decomposition(Wnew, H)   # cnmfsc.m:235: 2-D slice => plain Wnew*H
                        newobj = 0.5 * np.sum((V - V_hat) ** 2)
                        if newobj <= begobj:
                            break
                        stepsizeW[t - 1] = stepsizeW[t - 1] / 2
                        if stepsizeW[t - 1] < 1e-200:     # cnmfsc.m:245-249
                            triesW.append(tries)
                            return _finish(it, True)
                    triesW.append(tries)
                    stepsizeW[t - 1] = 1.2 * stepsizeW[t - 1]             # cnmfsc.m:252
                    W[:, :, t - 1] = Wnew
            else:
                for t in range(1, T + 1):                 # cnmfsc.m:257-263
                    Hsh = _rshift(H, t, n)
                    neg = V @ Hsh.T
                    pos = V_hat @ Hsh.T
                    with np.errstate(divide="ignore", invalid="ignore"):
                        W[:, :, t - 1] = W0[:, :, t - 1] * (neg / np.fmax(pos, EPS))
                    V_hat = np.fmax(V_hat + np.ascontiguousarray(W[:, :, t - 1] - W0[:, :, t - 1]) @ Hsh, 0.0)
        W0 = W.copy()                                     # cnmfsc.m:266 (value semantics)
        V_hat = rfd3(W0, H)                               # cnmfsc.m:269
        cost[it] = 0.5 * np.sum((V - V_hat) ** 2)
        if it > 1 and cost[it] < cost[it - 1] and cost[it - 1] - cost[it] < tol:   # cnmfsc.m:273-276
            n_cost = it + 1
            break
    return _finish(n_cost)


# --------------------------------------------------------------------------------------
# lnmf.m:49-92.  Note the reference's details: L1 column normalisation, no diagonal
# terms, H <- sqrt(H .* (W'*(V./V_hat))), stop rule with <= (lnmf.m:87) and a cost vector that is NOT trimmed on break.
# --------------------------------------------------------------------------------------
def lnmf(V, num_basis_elems, config=None, rng=None):
    V = np.asarray(V, dtype=np.float64)
    m, n = V.shape
    K = int(num_basis_elems)
    cfg = dict(config) if config is not None else {}
    rng = rng if rng is not None else np.random.RandomState(0)
    if cfg.get("H_init", None) is None or np.size(cfg["H_init"]) == 0:       # lnmf.m:104-106
        cfg["H_init"] = np.fmax(rng.rand(K, n), EPS)
    if cfg.get("W_init", None) is None or np.size(cfg["W_init"]) == 0:       # lnmf.m:108-111
        w = np.fmax(rng.rand(m, K), EPS)
        cfg["W_init"] = w * (1.0 / np.sum(w, axis=0))[None, :]
    W_fixed = bool(cfg.get("W_fixed", False) or False)
    H_fixed = bool(cfg.get("H_fixed", False) or False)
    maxiter = cfg.get("maxiter", None)
    maxiter = 100 if (maxiter is None or maxiter <= 0) else int(maxiter)     # lnmf.m:121-123
    tol = cfg.get("tolerance", None)
    tol = 1e-3 if (tol is None or tol <= 0) else float(tol)                  # lnmf.m:125-127
    W = np.array(cfg["W_init"], dtype=np.float64)
    W = W * (1.0 / np.sum(W, axis=0))[None, :]            # lnmf.m:59
    H = np.array(cfg["H_init"], dtype=np.float64)
    V_hat = W @ H                                         # lnmf.m:62
    cost = np.zeros(maxiter)
    ones_mn = np.ones((m, n))
    with np.errstate(divide="ignore", invalid="ignore"):
        for it in range(1, maxiter + 1):
            if not W_fixed:                               # lnmf.m:68-72
                W = W * (((V / V_hat) @ H.T) / np.fmax(ones_mn @ H.T, EPS))
                W = W * (1.0 / np.sum(W, axis=0))[None, :]
                V_hat = W @ H
            if not H_fixed:                               # lnmf.m:75-78
                H = np.sqrt(H * (W.T @ (V / V_hat)))
                V_hat = W @ H
            cost[it - 1] = np.sum(V * np.log(V / V_hat) - V + V_hat)          # lnmf.m:81
            if it > 1 and cost[it - 1] <= cost[it - 2] and cost[it - 2] - cost[it - 1] <= tol:   # lnmf.m:84-86
                break
    return W, H, cost                                     # cost is NOT trimmed (lnmf.m:85-86 only breaks)


# --------------------------------------------------------------------------------------
# constrainednmf.m:92-267   
# --------------------------------------------------------------------------------------
def constrainednmf(V, labels, num_basis_elems, config=None, rng=None):
    """constrainednmf.m:1.  Returns (W, H, Z, A, cost).  The reference draws Z with rand() inside the function
    (constrainednmf.m:174); config['Z_init'] overrides that draw so seeded comparisons are possible (test aid)."""
    V = np.asarray(V, dtype=np.float64)
    m, n = V.shape                                        # constrainednmf.m:96
    labels = np.asarray(labels).reshape(-1)
    if labels.size != n:                                  # constrainednmf.m:98
        raise ValueError("Length of the label vector not equal to number of samples. Length of label vector = %d; number of samples = %d"
                         % (labels.size, n))
    K = int(num_basis_elems)
    cfg = dict(config) if config is not None else {}
    rng = rng if rng is not None else np.random.RandomState(0)

    def empty(key):
        return cfg.get(key, None) is None or np.size(cfg[key]) == 0

    if empty("W_init"):                                   # constrainednmf.m:100-102
        cfg["W_init"] = rng.rand(m, K)
    lamW = 0.0 if empty("W_sparsity") else float(cfg["W_sparsity"])        # 103-105
    lamZ = 0.0 if empty("Z_sparsity") else float(cfg["Z_sparsity"])        # 106-108
    W_fixed = False if empty("W_fixed") else bool(cfg["W_fixed"])          # 109-111
    Z_fixed = False if empty("Z_fixed") else bool(cfg["Z_fixed"])          # 112-114
    div = cfg.get("divergence", "euclidean")              # 115-117
    is_ab = div in ("ab_divergence", "ab")
    alpha = float(cfg["alpha"]) if ("alpha" in cfg and is_ab) else 1.0     # 118-122
    beta = float(cfg["beta"]) if ("beta" in cfg and is_ab) else 1.0        # 123-127
    use_dual = alpha == 0                                 # 128-132
    maxiter = cfg.get("maxiter", None)
    maxiter = 100 if (maxiter is None or maxiter <= 0) else int(maxiter)   # 133-135
    tol = cfg.get("tolerance", None)
    tol = 1e-3 if (tol is None or tol <= 0) else float(tol)                # 136-138
    if is_ab and alpha == 0 and beta == 0:                # 140-142
        raise ValueError("alpha = 0 and beta = 0 is not supported at this time.")

    W = _col_normalize(np.array(cfg["W_init"], dtype=np.float64))          # 144-145

    num_labeled = int(np.count_nonzero(labels > -1))      # 149
    uniq, inv = np.unique(labels, return_inverse=True)    # 151 / 156 (MATLAB's third output is 1-based)
    processed = inv.reshape(-1).astype(np.int64) + 1
    if num_labeled < n:                                   # 150-154
        processed = processed - 1
        processed[processed == 0] = -1
        num_classes = len(uniq) - 1
    else:                                                 # 155-158
        num_classes = len(uniq)
    sorted_idx = np.argsort(processed, kind="stable")     # 163 (MATLAB sort is stable)
    sorted_labels = processed[sorted_idx]
    V = V[:, sorted_idx]                                  # 164
    n_u = n - num_labeled
    Cm = np.zeros((num_classes, num_labeled))             # 166-169
    for samp in range(n_u, n):
        Cm[sorted_labels[samp] - 1, samp - n_u] = 1.0
    A = np.block([[np.eye(n_u), np.zeros((n_u, num_labeled))],
                  [np.zeros((num_classes, n_u)), Cm]])    # 170
    nz = n + num_classes - num_labeled
    Z = rng.rand(K, nz) if empty("Z_init") else np.array(cfg["Z_init"], dtype=np.float64)   # 174
    H = Z @ A                                             # 177
    V_hat = reconstruct_from_decomposition(W, H)          # 179
    cost = np.zeros(maxiter)                              # 181
    ones_mn, ones_nm = np.ones((m, n)), np.ones((n, m))
    n_run = maxiter
    with np.errstate(divide="ignore", invalid="ignore"):
        for it in range(1, maxiter + 1):                  # 183
            if not W_fixed:                               # 185-209: the W step of nmf.m
                if div == "euclidean":
                    neg = V @ H.T + W * _ddiag(H @ V_hat.T @ W)[None, :]
                    pos = V_hat @ H.T + W * _ddiag(H @ V.T @ W)[None, :]
                elif div in ("kl_divergence", "kl"):
                    neg = (V / V_hat) @ H.T + W * _ddiag(H @ ones_nm @ W)[None, :]
                    pos = ones_mn @ H.T + W * _ddiag(H @ (V.T / V_hat.T) @ W)[None, :]
                elif div in ("is_divergence", "is"):
                    neg = (V / V_hat ** 2) @ H.T + W * _ddiag(H @ (ones_nm / V_hat.T) @ W)[None, :]
                    pos = (ones_mn / V_hat) @ H.T + W * _ddiag(H @ (V.T / V_hat.T ** 2) @ W)[None, :]
                elif is_ab:
                    if use_dual:
                        neg = ((V ** (alpha - 1) * V_hat ** beta) @ H.T + W * _ddiag(H @ V.T ** (alpha + beta - 1) @ W)[None, :]) ** (1 / beta)
                        pos = (V ** (alpha + beta - 1) @ H.T + W * _ddiag(H @ (V ** (alpha - 1) * V_hat ** beta).T @ W)[None, :]) ** (1 / beta)
                    else:
                        neg = ((V ** alpha * V_hat ** (beta - 1)) @ H.T + W * _ddiag(H @ V_hat.T ** (alpha + beta - 1) @ W)[None, :]) ** (1 / alpha)
                        pos = (V_hat ** (alpha + beta - 1) @ H.T + W * _ddiag(H @ (V ** alpha * V_hat ** (beta - 1)).T @ W)[None, :]) ** (1 / alpha)
                else:                                     # 204-205
                    raise ValueError("No update equations defined for cost function with divergence type " + str(div))
                W = W * (neg / np.fmax(pos + lamW, EPS))  # 207
                W = _col_normalize(W)                     # 208
            V_hat = reconstruct_from_decomposition(W, H)  # 210
            if not Z_fixed:                               # 213-236
                if div == "euclidean":
                    neg = W.T @ V @ A.T
                    pos = W.T @ V_hat @ A.T
                elif div in ("kl_divergence", "kl"):
                    neg = W.T @ (V / V_hat) @ A.T
                    pos = W.T @ ones_mn @ A.T
                elif div in ("is_divergence", "is"):
                    neg = W.T @ (V / V_hat ** 2) @ A.T
                    pos = W.T @ (ones_mn / (W @ H)) @ A.T
                elif is_ab:
                    if use_dual:
                        neg = (W.T @ (V ** (alpha - 1) * V_hat ** beta) @ A.T) ** (1 / beta)
                        pos = (W.T @ V ** (alpha + beta - 1) @ A.T) ** (1 / beta)
                    else:
                        # constrainednmf.m:229 reads  W' * V.^alpha .* V_hat.^(beta-1) * A'  -- by MATLAB precedence that is
                        # ((W'*V.^alpha) .* V_hat.^(beta-1)) * A', a K x n times m x n element-wise product: a dimension error
                        raise ValueError("Matrix dimensions must agree.")
                else:
                    raise ValueError("No update equations defined for cost function with divergence type " + str(div))
                Z = Z * (neg / np.fmax(pos + lamZ, EPS))  # 235
            H = Z @ A                                     # 237
            V_hat = reconstruct_from_decomposition(W, H)  # 238
            c = _cost(div, V, V_hat, alpha, beta)         # 241-250
            cost[it - 1] = c + lamW * np.sum(np.abs(W)) + lamZ * np.sum(np.abs(Z))   # 251
            if it > 1 and cost[it - 1] < cost[it - 2] and cost[it - 2] - cost[it - 1] < tol:   # 254-257
                n_run = it
                break
    cost = cost[:n_run]
    A_temp = A.copy()                                     # 263-266
    for samp in range(n):
        A[:, sorted_idx[samp]] = A_temp[:, samp]
    H = Z @ A                                             # 267
    return W, H, Z, A, cost


# --------------------------------------------------------------------------------------
# SortDictionary.m:25-49
# --------------------------------------------------------------------------------------
def sort_dictionary(W, H=None):
    W = np.asarray(W, dtype=np.float64)
    K = W.shape[1]                                        # SortDictionary.m:31
    W_sum = np.cumsum(W, axis=0)                          # 33
    cog = np.zeros(K, dtype=np.int64)
    for j in range(K):                                    # 35-42
        hit = np.nonzero(W_sum[:, j] <= W_sum[-1, j] / 2)[0]
        cog[j] = 1 if hit.size == 0 else hit[-1] + 1
    order = np.argsort(cog, kind="stable")                # 43
    W_sorted = W[:, order]                                # 44
    H_sorted = None if H is None else np.asarray(H, dtype=np.float64)[order, :]   # 45-47
    return W_sorted, H_sorted
