"""Host-side mirror of the reference call surface for the hot path, above the C ABI.

    W, H, cost = nmf(V, num_basis_elems, config)                  nmf.m:1
    W, H, cost = cnmf(V, num_basis_elems, context_len, config)    cnmf.m:1
    W, H, cost = nmfsc(V, num_basis_elems, config)                nmfsc.m:1
    V_hat      = ReconstructFromDecomposition(W, H)               ReconstructFromDecomposition.m:1
    v, iters   = projfunc(s, k1, k2, nn)                          projfunc.m:1
    W, H, cost = lnmf(V, num_basis_elems, config)                 lnmf.m:1            
    W, H, cost = cnmfsc(V, num_basis_elems, context_len, config)  cnmfsc.m:1          (f1)
    W, H, Z, A, cost = constrainednmf(V, labels, num_basis_elems, config)  constrainednmf.m:1   (f4)
    W_sorted, H_sorted = SortDictionary(W, H)                     SortDictionary.m:1  (f4)

Same argument meaning, defaults and error behaviour as the MATLAB functions (a MATLAB cell array is
a Python list, a struct a dict; errors are ValueError carrying the reference's message).  This file
does only what the reference's local `ValidateParameters` does (nmf.m:238-413, cnmf.m:271-449) plus
packing for the C ABI; every numeric step runs in libnmfx on the MI355X.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

EPS = 2.0 ** -52

_DIV_NMF = {"euclidean": _lib.DIV_EUCLIDEAN, "kl_divergence": _lib.DIV_KL, "kl": _lib.DIV_KL,
            "is_divergence": _lib.DIV_IS, "is": _lib.DIV_IS, "ab_divergence": _lib.DIV_AB, "ab": _lib.DIV_AB}


def _is_cell(x):
    return isinstance(x, (list, tuple))


def _isempty(x):
    if x is None:
        return True
    if _is_cell(x):
        return len(x) == 0
    return np.size(x) == 0


def _rng(config):
    r = config.get("rng", None) if config else None
    if r is not None:
        return r
    seed = config.get("seed", None) if config else None
    return np.random.RandomState(seed)


def _per_source(cfg, name, S, default, conv, what):
    """nmf.m:312-401: missing/empty -> default; scalar or 1-cell -> broadcast; S-cell kept; else error."""
    val = cfg.get(name, None)
    if _isempty(val):
        return [default] * S
    if _is_cell(val) and len(val) > 1 and len(val) != S:
        raise ValueError("Requested %d sources. Given %d %s." % (S, len(val), what))
    if not _is_cell(val) or len(val) == 1:
        t = conv(val[0] if _is_cell(val) else val)
        return [t] * S
    return [conv(t) for t in val]


def _validate(V, Ks, T, config, cnmf_mode):
    """The local ValidateParameters of nmf.m:238-413 (cnmf_mode False) / cnmf.m:271-449 (True)."""
    cfg = dict(config) if config else {}
    m, n = V.shape
    S = len(Ks)
    rng = _rng(cfg)
    if "divergence" not in cfg:                                   # nmf.m:250-252
        cfg["divergence"] = "euclidean"
    is_ab = cfg["divergence"] in ("ab_divergence", "ab")
    if "alpha" not in cfg or not is_ab:                           # nmf.m:255-259
        cfg["alpha"] = 1.0
    if "beta" not in cfg or not is_ab:                            # nmf.m:262-266
        cfg["beta"] = 1.0
    Hi = cfg.get("H_init", None)                                  # nmf.m:269-287
    if _isempty(Hi):
        is_H_cell = S != 1
        H = [np.fmax(rng.rand(K, n), EPS) for K in Ks]
    elif _is_cell(Hi) and len(Hi) != S:
        raise ValueError("Requested %d sources. Given %d initial encoding matrices." % (S, len(Hi)))
    elif not _is_cell(Hi):
        is_H_cell = False
        H = [np.asarray(Hi, dtype=np.float64)]
    else:
        is_H_cell = True
        H = [np.asarray(h, dtype=np.float64) for h in Hi]
    Wi = cfg.get("W_init", None)                                  # nmf.m:290-309 / cnmf.m:323-345
    if _isempty(Wi):
        is_W_cell = S != 1
        W = []
        for K in Ks:
            if cnmf_mode:
                w = rng.rand(m, K, T)
                w = w / (np.sqrt(np.sum(w ** 2, axis=(0, 2))) / T)[None, :, None]
            else:
                w = np.fmax(rng.rand(m, K), EPS)
                w = w * (1.0 / np.sqrt(np.sum(w ** 2, axis=0)))[None, :]
            W.append(w)
    elif _is_cell(Wi) and len(Wi) != S:
        raise ValueError("Requested %d sources. Given %d initial basis matrices." % (S, len(Wi)))
    elif not _is_cell(Wi):
        is_W_cell = False
        W = [np.asarray(Wi, dtype=np.float64)]
    else:
        is_W_cell = True
        W = [np.asarray(w, dtype=np.float64) for w in Wi]
    nonneg = lambda x: max(float(x), 0.0)
    cfg["W_sparsity"] = _per_source(cfg, "W_sparsity", S, 0.0, nonneg, "sparsity levels")     # nmf.m:312-334
    cfg["H_sparsity"] = _per_source(cfg, "H_sparsity", S, 0.0, nonneg, "sparsity levels")     # nmf.m:337-359
    cfg["W_fixed"] = _per_source(cfg, "W_fixed", S, False, bool, "update switches")           # nmf.m:362-380
    cfg["H_fixed"] = _per_source(cfg, "H_fixed", S, False, bool, "update switches")           # nmf.m:383-401
    if "maxiter" not in cfg or cfg["maxiter"] is None or cfg["maxiter"] <= 0:                 # nmf.m:404-406
        cfg["maxiter"] = 100
    if "tolerance" not in cfg or cfg["tolerance"] is None or cfg["tolerance"] <= 0:           # nmf.m:409-411
        cfg["tolerance"] = 1e-3
    return cfg, W, H, is_W_cell, is_H_cell


def _fptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _as_data(V):
    """V as the library takes it: float32 or float64 (anything else is widened to float64, like MATLAB's double), column-major.
    A column-major array of either type is passed as is -- no copy of an 8 GiB matrix just to hand it over."""
    V = np.asarray(V)
    if V.dtype != np.float32 and V.dtype != np.float64:
        V = V.astype(np.float64)
    return V


def _multi_backend(v):
    """config.nmfx_multi_backend: a name, or the number the MATLAB wrappers pass (0 auto | 1 peer | 2 rccl); anything else is an error, not a silent auto"""
    names = {None: 0, "auto": 0, "peer": 1, "rccl": 2}
    if isinstance(v, str) or v is None:
        if v not in names:
            raise ValueError("nmfx_multi_backend must be 'auto', 'peer' or 'rccl' (or 0, 1, 2); got %r" % (v,))
        return names[v]
    if isinstance(v, (int, np.integer, float)) and not isinstance(v, bool) and float(v) in (0.0, 1.0, 2.0):
        return int(v)
    raise ValueError("nmfx_multi_backend must be 'auto', 'peer' or 'rccl' (or 0, 1, 2); got %r" % (v,))


def _gpu_ids(cfg):
    gpus = cfg.get("nmfx_gpus", None)
    if gpus is None:
        return None
    return np.asarray(list(range(int(gpus))) if np.isscalar(gpus) else list(gpus), dtype=np.int32)


def _f_order(a, dtype):
    return a if (a.dtype == dtype and a.flags.f_contiguous) else np.asfortranarray(a, dtype=dtype)


def _run_mu(fn, V, Ks, T, cfg, W, H, divergence, device):
    m, n = V.shape
    S = len(Ks)
    K = int(sum(Ks))
    for s in range(S):
        if W[s].shape[0] != m or W[s].shape[1] != Ks[s]:
            raise ValueError("W_init{%d} must be %d-by-%d" % (s + 1, m, Ks[s]))
        if H[s].shape != (Ks[s], n):
            raise ValueError("H_init{%d} must be %d-by-%d" % (s + 1, Ks[s], n))
    W3 = [w.reshape(w.shape[0], w.shape[1], -1) for w in W]
    for w in W3:
        if w.shape[2] != T:
            raise ValueError("W_init has context length %d, expected %d" % (w.shape[2], T))
    # the three arrays travel in V's own precision (float32 data stays float32: half the host traffic, same device arithmetic)
    dt = V.dtype
    W_all = _f_order(W3[0] if S == 1 else np.concatenate(W3, axis=1), dt)   # cell2mat(1xS) -> along dim 2 (nmf.m:136)
    H_all = _f_order(H[0] if S == 1 else np.concatenate(H, axis=0), dt)     # cell2mat(Sx1) -> along dim 1 (nmf.m:137)
    Vf = _f_order(V, dt)
    maxiter = int(cfg["maxiter"])
    Wout = np.zeros((m, K, T), order="F", dtype=dt)
    Hout = np.zeros((K, n), order="F", dtype=dt)
    cost = np.zeros(maxiter)
    Ks_a = np.asarray(Ks, dtype=np.int32)
    lw = np.asarray(cfg["W_sparsity"], dtype=np.float64)
    lh = np.asarray(cfg["H_sparsity"], dtype=np.float64)
    fw = np.asarray(cfg["W_fixed"], dtype=np.uint8)
    fh = np.asarray(cfg["H_fixed"], dtype=np.uint8)
    p = _lib.Problem()
    p.m, p.n, p.K_total, p.T, p.dtype = m, n, K, T, (_lib.F32 if dt == np.float32 else _lib.F64)
    p.V, p.W_init, p.H_init = _fptr(Vf), _fptr(W_all), _fptr(H_all)
    p.divergence, p.alpha, p.beta = divergence, float(cfg["alpha"]), float(cfg["beta"])
    p.num_sources, p.K_s = S, _fptr(Ks_a)
    p.W_sparsity, p.H_sparsity, p.W_fixed, p.H_fixed = _fptr(lw), _fptr(lh), _fptr(fw), _fptr(fh)
    p.maxiter = maxiter
    p.tolerance = -1.0 if cfg.get("nmfx_disable_stop", False) else float(cfg["tolerance"])
    p.device = int(device)
    p.path = int(cfg.get("nmfx_path", 0))       # extension: 0 auto, 1 generic kernels only, 2 require the fused kernels
    # extension: nmfx_gpus = N or a list of device ordinals -> V / H column-sharded over N GPUs of this process (nmf, cnmf, lnmf, nmfsc)
    ids = _gpu_ids(cfg)
    if ids is not None:
        p.n_gpus, p.device_ids = int(ids.size), _fptr(ids)
    # extension: the exchange of the packed W-step sums between those GPUs -- "rccl" (ncclAllReduce), "peer" (reduce-scatter + all-gather over peer mappings), default auto
    p.multi_backend = _multi_backend(cfg.get("nmfx_multi_backend", None))
    r = _lib.Result()
    r.W, r.H, r.cost = _fptr(Wout), _fptr(Hout), _fptr(cost)
    _lib.check(fn(C.byref(p), C.byref(r)))
    cost = cost[: r.cost_len].copy()
    Wl, Hl, k0 = [], [], 0
    for s in range(S):
        Wl.append(np.array(Wout[:, k0:k0 + Ks[s], :]))
        Hl.append(np.array(Hout[k0:k0 + Ks[s], :]))
        k0 += Ks[s]
    return Wl, Hl, cost


def nmf(V, num_basis_elems, config=None, device=0):
    """[W, H, cost] = nmf(V, num_basis_elems, config)  -- nmf.m:1."""
    V = _as_data(V)
    if V.ndim != 2:
        raise ValueError("V must be a matrix")
    Ks = [int(k) for k in (num_basis_elems if _is_cell(num_basis_elems) else [num_basis_elems])]   # nmf.m:114-117
    cfg, W, H, is_W_cell, is_H_cell = _validate(V, Ks, 1, config, False)                           # nmf.m:118
    div = cfg["divergence"]
    if div in ("ab_divergence", "ab") and cfg["alpha"] == 0 and cfg["beta"] == 0:                   # nmf.m:120-122
        raise ValueError("alpha = 0 and beta = 0 is not supported at this time.")
    if div not in _DIV_NMF:                                                                        # nmf.m:165-166
        raise ValueError("No update equations defined for cost function with divergence type " + str(div))
    Wl, Hl, cost = _run_mu(_lib.load().nmfx_nmf, V, Ks, 1, cfg, W, H, _DIV_NMF[div], device)
    Wl = [w[:, :, 0] for w in Wl]
    return (Wl if is_W_cell else Wl[0]), (Hl if is_H_cell else Hl[0]), cost                        # nmf.m:228-234


def cnmf(V, num_basis_elems, context_len, config=None, device=0):
    """[W, H, cost] = cnmf(V, num_basis_elems, context_len, config)  -- cnmf.m:1."""
    V = _as_data(V)
    if V.ndim != 2:
        raise ValueError("V must be a matrix")
    T = int(context_len)
    Ks = [int(k) for k in (num_basis_elems if _is_cell(num_basis_elems) else [num_basis_elems])]
    cfg, W, H, is_W_cell, is_H_cell = _validate(V, Ks, T, config, True)                            # cnmf.m:131
    div = cfg["divergence"]
    if div in ("ab_divergence", "ab") and cfg["alpha"] == 0 and cfg["beta"] == 0:                   # cnmf.m:133-135
        raise ValueError("alpha = 0 and beta = 0 is not supported at this time.")
    # cnmf.m:137-147 has no `otherwise`: 'frobenius' and any unrecognised string run the (1,1) updates; the cost
    # switch (cnmf.m:239-248) has no case for them either, so their cost vector holds only the L1 terms.
    code = _DIV_NMF.get(div, _lib.DIV_EUCLIDEAN_NOCOST)
    Wl, Hl, cost = _run_mu(_lib.load().nmfx_cnmf, V, Ks, T, cfg, W, H, code, device)
    if T == 1:                                             # rand(m,K,1) is a matrix in MATLAB
        Wl = [w[:, :, 0] for w in Wl]
    return (Wl if is_W_cell else Wl[0]), (Hl if is_H_cell else Hl[0]), cost                        # cnmf.m:261-267


def lnmf(V, num_basis_elems, config=None, device=0):
    """[W, H, cost] = lnmf(V, num_basis_elems, config)  -- lnmf.m:1.  `cost` has maxiter entries, zero
    after an early stop (the reference breaks without trimming, lnmf.m:84-86)."""
    V = np.asarray(V, dtype=np.float64)
    if V.ndim != 2:
        raise ValueError("V must be a matrix")
    m, n = V.shape
    K = int(num_basis_elems)
    cfg = dict(config) if config else {}
    rng = _rng(cfg)
    if _isempty(cfg.get("H_init", None)):                          # lnmf.m:104-106
        cfg["H_init"] = np.fmax(rng.rand(K, n), EPS)
    if _isempty(cfg.get("W_init", None)):                          # lnmf.m:108-111
        w = np.fmax(rng.rand(m, K), EPS)
        cfg["W_init"] = w * (1.0 / np.sum(w, axis=0))[None, :]
    cfg["W_fixed"] = [False if _isempty(cfg.get("W_fixed", None)) else bool(cfg["W_fixed"])]
    cfg["H_fixed"] = [False if _isempty(cfg.get("H_fixed", None)) else bool(cfg["H_fixed"])]
    cfg["W_sparsity"], cfg["H_sparsity"] = [0.0], [0.0]
    if cfg.get("maxiter", None) is None or cfg["maxiter"] <= 0:      # lnmf.m:121-123
        cfg["maxiter"] = 100
    if cfg.get("tolerance", None) is None or cfg["tolerance"] <= 0:  # lnmf.m:125-127
        cfg["tolerance"] = 1e-3
    cfg["alpha"] = cfg["beta"] = 1.0
    W0 = np.asarray(cfg["W_init"], dtype=np.float64)
    H0 = np.asarray(cfg["H_init"], dtype=np.float64)
    Wl, Hl, cost = _run_mu(_lib.load().nmfx_lnmf, V, [K], 1, cfg, [W0], [H0], _lib.DIV_KL, device)
    return Wl[0][:, :, 0], Hl[0], cost


def _label_segments(labels, n):
    """constrainednmf.m:147-170 -- host bookkeeping only: processed labels, the stable sort that makes equal labels contiguous
    (unlabelled samples, label -1, first) and, instead of the dense 0/1 matrix A, the column ranges of its non-zeros."""
    labels = np.asarray(labels).reshape(-1)
    if labels.size != n:                                            # constrainednmf.m:98
        raise ValueError("Length of the label vector not equal to number of samples. Length of label vector = %d; number of samples = %d"
                         % (labels.size, n))
    num_labeled = int(np.count_nonzero(labels > -1))               # constrainednmf.m:149
    uniq, processed = np.unique(labels, return_inverse=True)      # constrainednmf.m:151/156 (1-based in MATLAB)
    processed = processed.reshape(-1).astype(np.int64) + 1
    if num_labeled < n:
        processed -= 1                                              # constrainednmf.m:152-154
        processed[processed == 0] = -1
        num_classes = len(uniq) - 1
    else:
        num_classes = len(uniq)
    sorted_idx = np.argsort(processed, kind="stable")              # constrainednmf.m:163
    sorted_labels = processed[sorted_idx]
    n_u = n - num_labeled
    seg = list(range(n_u + 1))
    for c in range(1, num_classes + 1):                            # rows of C, constrainednmf.m:166-169
        seg.append(seg[-1] + int(np.count_nonzero(sorted_labels[n_u:] == c)))
    return sorted_idx, np.asarray(seg, dtype=np.int64), n_u, num_classes


def constrainednmf(V, labels, num_basis_elems, config=None, device=0):
    """[W, H, Z, A, cost] = constrainednmf(V, labels, num_basis_elems, config)  -- constrainednmf.m:1.

    The reference draws Z with rand() inside the function (constrainednmf.m:174); `config['Z_init']` (extension) supplies it
    for reproducible runs.  A is returned dense like the reference's; config['nmfx_sparse_A'] = True returns scipy CSR instead."""
    V = np.asarray(V, dtype=np.float64)
    if V.ndim != 2:
        raise ValueError("V must be a matrix")
    m, n = V.shape
    K = int(num_basis_elems)
    cfg = dict(config) if config else {}
    sorted_idx, seg, n_u, num_classes = _label_segments(labels, n)
    rng = _rng(cfg)
    if _isempty(cfg.get("W_init", None)):                          # constrainednmf.m:100-102
        cfg["W_init"] = rng.rand(m, K)
    for key in ("W_sparsity", "Z_sparsity"):                      # constrainednmf.m:103-108
        cfg[key] = 0.0 if _isempty(cfg.get(key, None)) else float(cfg[key])
    for key in ("W_fixed", "Z_fixed"):                            # constrainednmf.m:109-114
        cfg[key] = False if _isempty(cfg.get(key, None)) else bool(cfg[key])
    if "divergence" not in cfg:                                    # constrainednmf.m:115-117
        cfg["divergence"] = "euclidean"
    div = cfg["divergence"]
    is_ab = div in ("ab_divergence", "ab")
    cfg["alpha"] = float(cfg["alpha"]) if ("alpha" in cfg and is_ab) else 1.0    # constrainednmf.m:118-127
    cfg["beta"] = float(cfg["beta"]) if ("beta" in cfg and is_ab) else 1.0
    if cfg.get("maxiter", None) is None or cfg["maxiter"] <= 0:     # constrainednmf.m:133-135
        cfg["maxiter"] = 100
    if cfg.get("tolerance", None) is None or cfg["tolerance"] <= 0:  # constrainednmf.m:136-138
        cfg["tolerance"] = 1e-3
    if is_ab and cfg["alpha"] == 0 and cfg["beta"] == 0:             # constrainednmf.m:140-142
        raise ValueError("alpha = 0 and beta = 0 is not supported at this time.")
    if div not in _DIV_NMF:                                        # constrainednmf.m:204-205
        raise ValueError("No update equations defined for cost function with divergence type " + str(div))
    nz = n_u + num_classes
    Z0 = cfg.get("Z_init", None)
    Z0 = rng.rand(K, nz) if _isempty(Z0) else np.asarray(Z0, dtype=np.float64)   # constrainednmf.m:174
    if Z0.shape != (K, nz):
        raise ValueError("Z_init must be %d-by-%d" % (K, nz))
    W0 = np.asarray(cfg["W_init"], dtype=np.float64)
    if W0.shape != (m, K):
        raise ValueError("W_init must be %d-by-%d" % (m, K))
    Vs = np.asfortranarray(V[:, sorted_idx])                       # constrainednmf.m:164
    W0 = np.asfortranarray(W0)
    Z0 = np.asfortranarray(Z0)
    maxiter = int(cfg["maxiter"])
    Wout, Hout, Zout, cost = np.zeros((m, K), order="F"), np.zeros((K, n), order="F"), np.zeros((K, nz), order="F"), np.zeros(maxiter)
    one = np.asarray([K], dtype=np.int32)
    lw, lz = np.asarray([cfg["W_sparsity"]]), np.asarray([cfg["Z_sparsity"]])
    fw, fz = np.asarray([cfg["W_fixed"]], dtype=np.uint8), np.asarray([cfg["Z_fixed"]], dtype=np.uint8)
    p = _lib.Problem()
    p.m, p.n, p.K_total, p.T, p.dtype = m, n, K, 1, _lib.F64
    p.V, p.W_init, p.H_init = _fptr(Vs), _fptr(W0), None
    p.divergence, p.alpha, p.beta = _DIV_NMF[div], cfg["alpha"], cfg["beta"]
    p.num_sources, p.K_s = 1, _fptr(one)
    p.W_sparsity, p.H_sparsity, p.W_fixed, p.H_fixed = _fptr(lw), _fptr(lz), _fptr(fw), _fptr(fz)
    p.maxiter = maxiter
    p.tolerance = -1.0 if cfg.get("nmfx_disable_stop", False) else float(cfg["tolerance"])
    p.device, p.path = int(device), int(cfg.get("nmfx_path", 0))
    r = _lib.Result()
    r.W, r.H, r.cost = _fptr(Wout), _fptr(Hout), _fptr(cost)
    _lib.check(_lib.load().nmfx_constrainednmf(C.byref(p), _fptr(seg), nz, _fptr(Z0), C.byref(r), _fptr(Zout)))
    # constrainednmf.m:259-267: A (and with it H = Z*A) goes back to the original sample order
    H = np.empty((K, n))
    H[:, sorted_idx] = Hout
    zcol = np.repeat(np.arange(nz), np.diff(seg))                  # Z column of every SORTED sample
    rows, cols = zcol, sorted_idx
    if cfg.get("nmfx_sparse_A", False):
        import scipy.sparse as sp
        A = sp.csr_matrix((np.ones(n), (rows, cols)), shape=(nz, n))
    else:
        A = np.zeros((nz, n))
        A[rows, cols] = 1.0
    return np.array(Wout), H, np.array(Zout), A, cost[: r.cost_len].copy()


def SortDictionary(W, H=None, device=0):
    """[W_sorted, H_sorted] = SortDictionary(W, H)  -- SortDictionary.m:1.  H_sorted is None when H is not given."""
    W = np.asfortranarray(W, dtype=np.float64)
    if W.ndim != 2:
        raise ValueError("SortDictionary does not work for CNMF bases")   # SortDictionary.m:3
    m, K = W.shape
    Ws = np.zeros((m, K), order="F")
    Hf = Hs = None
    n = 0
    if H is not None:
        Hf = np.asfortranarray(H, dtype=np.float64)
        if Hf.ndim != 2 or Hf.shape[0] != K:
            raise ValueError("H must have %d rows" % K)
        n = Hf.shape[1]
        Hs = np.zeros((K, n), order="F")
    order = np.zeros(K, dtype=np.int32)
    _lib.check(_lib.load().nmfx_sort_dictionary(m, K, n, _lib.F64, _fptr(W), _fptr(Hf) if Hf is not None else None, _fptr(Ws),
                                                _fptr(Hs) if Hs is not None else None, _fptr(order), int(device)))
    return Ws, Hs


def nmfsc(V, num_basis_elems, config=None, device=0, info=None):
    """[W, H, cost] = nmfsc(V, num_basis_elems, config)  -- nmfsc.m:1.

    `info` (dict, optional) receives the line-search try counts and final step sizes (test aid).
    """
    V = np.asarray(V, dtype=np.float64)
    if V.ndim != 2:
        raise ValueError("V must be a matrix")
    if V.min() < 0:                                                # nmfsc.m:57-59
        raise ValueError("Negative values in data!")
    m, n = V.shape
    K = int(num_basis_elems)
    cfg = dict(config) if config else {}
    rng = _rng(cfg)
    if _isempty(cfg.get("W_init", None)):                          # nmfsc.m:73-75
        cfg["W_init"] = rng.rand(m, K)
    if _isempty(cfg.get("H_init", None)):                          # nmfsc.m:78-81
        h = rng.rand(K, n)
        cfg["H_init"] = (1.0 / np.sqrt(np.sum(h ** 2, axis=1)))[:, None] * h
    W0 = np.asfortranarray(cfg["W_init"], dtype=np.float64)
    H0 = np.asfortranarray(cfg["H_init"], dtype=np.float64)
    if W0.shape != (m, K) or H0.shape != (K, n):
        raise ValueError("W_init must be %d-by-%d and H_init %d-by-%d" % (m, K, K, n))
    sW = 0.0 if _isempty(cfg.get("W_sparsity", None)) else float(cfg["W_sparsity"])   # nmfsc.m:87-92
    sH = 0.0 if _isempty(cfg.get("H_sparsity", None)) else float(cfg["H_sparsity"])   # nmfsc.m:100-105
    fixW = False if _isempty(cfg.get("W_fixed", None)) else bool(cfg["W_fixed"])      # nmfsc.m:113-115
    fixH = False if _isempty(cfg.get("H_fixed", None)) else bool(cfg["H_fixed"])      # nmfsc.m:118-120
    maxiter = cfg.get("maxiter", None)
    maxiter = 100 if (maxiter is None or maxiter <= 0) else int(maxiter)              # nmfsc.m:123-125
    tol = cfg.get("tolerance", None)
    tol = 1e-3 if (tol is None or tol <= 0) else float(tol)                           # nmfsc.m:128-130
    Vf = np.asfortranarray(V)
    Wout = np.zeros((m, K), order="F")
    Hout = np.zeros((K, n), order="F")
    cost = np.zeros(maxiter + 1)
    tH = np.zeros(maxiter, dtype=np.int32)
    tW = np.zeros(maxiter, dtype=np.int32)
    fw = np.asarray([fixW], dtype=np.uint8)
    fh = np.asarray([fixH], dtype=np.uint8)
    p = _lib.Problem()
    p.m, p.n, p.K_total, p.T, p.dtype = m, n, K, 1, _lib.F64
    p.V, p.W_init, p.H_init = _fptr(Vf), _fptr(W0), _fptr(H0)
    p.num_sources = 1
    p.W_fixed, p.H_fixed = _fptr(fw), _fptr(fh)
    p.maxiter, p.tolerance, p.device = maxiter, (-1.0 if cfg.get("nmfx_disable_stop", False) else tol), int(device)
    p.sc_W_sparsity, p.sc_H_sparsity = sW, sH
    p.path = int(cfg.get("nmfx_path", 0))
    ids = _gpu_ids(cfg)                                            # extension: column shards over N GPUs of this process
    if ids is not None:
        p.n_gpus, p.device_ids = int(ids.size), _fptr(ids)
    r = _lib.Result()
    r.W, r.H, r.cost, r.tries_H, r.tries_W = _fptr(Wout), _fptr(Hout), _fptr(cost), _fptr(tH), _fptr(tW)
    _lib.check(_lib.load().nmfx_nmfsc(C.byref(p), C.byref(r)))
    if r.converged_early:
        print("Algorithm converged")                               # nmfsc.m:171 display(...)
    if info is not None:
        info.update(triesH=[int(t) for t in tH if t > 0], triesW=[int(t) for t in tW if t > 0],
                    stepsizeH=r.stepsize_H, stepsizeW=r.stepsize_W, converged_early=bool(r.converged_early))
    return np.array(Wout), np.array(Hout), cost[: r.cost_len].copy()


def cnmfsc(V, num_basis_elems, context_len, config=None, device=0, info=None):
    """[W, H, cost] = cnmfsc(V, num_basis_elems, context_len, config)  -- cnmfsc.m:1."""
    V = np.asarray(V, dtype=np.float64)
    if V.ndim != 2:
        raise ValueError("V must be a matrix")
    if V.min() < 0:                                                # cnmfsc.m:67-69
        raise ValueError("Negative values in data!")
    m, n = V.shape
    K, T = int(num_basis_elems), int(context_len)
    cfg = dict(config) if config else {}
    rng = _rng(cfg)
    if _isempty(cfg.get("W_init", None)):                          # cnmfsc.m:83-85
        cfg["W_init"] = rng.rand(m, K, T)
    if _isempty(cfg.get("H_init", None)):                          # cnmfsc.m:88-91
        h = rng.rand(K, n)
        cfg["H_init"] = (1.0 / np.sqrt(np.sum(h ** 2, axis=1)))[:, None] * h
    W0 = np.asfortranarray(np.asarray(cfg["W_init"], dtype=np.float64).reshape(m, K, -1))
    H0 = np.asfortranarray(cfg["H_init"], dtype=np.float64)
    if W0.shape != (m, K, T) or H0.shape != (K, n):
        raise ValueError("W_init must be %d-by-%d-by-%d and H_init %d-by-%d" % (m, K, T, K, n))
    sW = 0.0 if _isempty(cfg.get("W_sparsity", None)) else float(cfg["W_sparsity"])
    sH = 0.0 if _isempty(cfg.get("H_sparsity", None)) else float(cfg["H_sparsity"])
    fixW = False if _isempty(cfg.get("W_fixed", None)) else bool(cfg["W_fixed"])
    fixH = False if _isempty(cfg.get("H_fixed", None)) else bool(cfg["H_fixed"])
    maxiter = cfg.get("maxiter", None)
    maxiter = 100 if (maxiter is None or maxiter <= 0) else int(maxiter)              # cnmfsc.m:137-139
    tol = cfg.get("tolerance", None)
    tol = 1e-3 if (tol is None or tol <= 0) else float(tol)                           # cnmfsc.m:142-144
    Vf = np.asfortranarray(V)
    Wout = np.zeros((m, K, T), order="F")
    Hout = np.zeros((K, n), order="F")
    cost = np.zeros(maxiter + 1)
    tH = np.zeros(maxiter, dtype=np.int32)
    tW = np.zeros(maxiter * T, dtype=np.int32)
    fw = np.asarray([fixW], dtype=np.uint8)
    fh = np.asarray([fixH], dtype=np.uint8)
    p = _lib.Problem()
    p.m, p.n, p.K_total, p.T, p.dtype = m, n, K, T, _lib.F64
    p.V, p.W_init, p.H_init = _fptr(Vf), _fptr(W0), _fptr(H0)
    p.num_sources = 1
    p.W_fixed, p.H_fixed = _fptr(fw), _fptr(fh)
    p.maxiter, p.tolerance, p.device = maxiter, (-1.0 if cfg.get("nmfx_disable_stop", False) else tol), int(device)
    p.sc_W_sparsity, p.sc_H_sparsity = sW, sH
    p.path = int(cfg.get("nmfx_path", 0))
    ids = _gpu_ids(cfg)
    if ids is not None:
        p.n_gpus, p.device_ids = int(ids.size), _fptr(ids)
    r = _lib.Result()
    r.W, r.H, r.cost, r.tries_H, r.tries_W = _fptr(Wout), _fptr(Hout), _fptr(cost), _fptr(tH), _fptr(tW)
    _lib.check(_lib.load().nmfx_cnmfsc(C.byref(p), C.byref(r)))
    if r.converged_early:
        print("Algorithm converged")                               # cnmfsc.m:191 display(...)
    if info is not None:
        info.update(triesH=[int(t) for t in tH if t > 0], triesW=[int(t) for t in tW if t > 0], stepsizeH=r.stepsize_H,
                    converged_early=bool(r.converged_early))
    Wr = np.array(Wout)
    return (Wr[:, :, 0] if T == 1 else Wr), np.array(Hout), cost[: r.cost_len].copy()


def ReconstructFromDecomposition(W, H, device=0):
    """V_hat = ReconstructFromDecomposition(W, H)  -- ReconstructFromDecomposition.m:1."""
    if _is_cell(W):                                                # RFD.m:23-25
        W = np.concatenate([np.asarray(w, dtype=np.float64).reshape(np.shape(w)[0], np.shape(w)[1], -1) for w in W], axis=1)
    if _is_cell(H):                                                # RFD.m:26-28
        H = np.concatenate([np.asarray(h, dtype=np.float64) for h in H], axis=0)
    W = np.asarray(W, dtype=np.float64)
    H = np.asfortranarray(H, dtype=np.float64)
    if W.ndim == 2:
        W = W.reshape(W.shape[0], W.shape[1], 1)
    m, K, T = W.shape
    if H.shape[0] != K:
        raise ValueError("Inner matrix dimensions must agree.")
    n = H.shape[1]
    Wf = np.asfortranarray(W)
    out = np.zeros((m, n), order="F")
    _lib.check(_lib.load().nmfx_reconstruct(m, n, K, T, _lib.F64, _fptr(Wf), _fptr(H), _fptr(out), int(device)))
    return np.array(out)


reconstruct_from_decomposition = ReconstructFromDecomposition


def projfunc(s, k1, k2, nn=True, device=0):
    """[v, usediters] = projfunc(s, k1, k2, nn)  -- projfunc.m:1 (one vector)."""
    s = np.ascontiguousarray(np.asarray(s, dtype=np.float64).reshape(-1))
    v = np.zeros_like(s)
    it = np.zeros(1, dtype=np.int32)
    _lib.check(_lib.load().nmfx_projfunc(s.size, 1, _lib.F64, _fptr(s), float(k1), float(k2), int(bool(nn)), _fptr(v), _fptr(it), int(device)))
    return v, int(it[0])
