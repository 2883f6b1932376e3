"""TEST INFRASTRUCTURE ONLY -- ctypes binding of oracle/liboracle.so (plain-C float64 restatement).

Takes/returns MATLAB-shaped NumPy arrays; converts to column-major buffers internally.
Multi-source problems are passed in concatenated form (per-column / per-row lambda and fixed masks),
which is exactly the form the HIP engine runs.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

DIV = {"euclidean": 0, "kl_divergence": 1, "kl": 1, "is_divergence": 2, "is": 2, "frobenius": 4}


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
    return _LIB


def _f(a):
    return np.asfortranarray(np.array(a, dtype=np.float64))


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _vec(x, K, dtype):
    a = np.ascontiguousarray(np.broadcast_to(np.asarray(x, dtype=dtype), (K,)))
    return a


def reconstruct(W, H):
    W = _f(W)
    H = _f(H)
    m, K = W.shape[0], W.shape[1]
    T = 1 if W.ndim == 2 else W.shape[2]
    n = H.shape[1]
    out = np.zeros((m, n), order="F")
    lib().oracle_reconstruct(m, n, K, T, _p(W), _p(H), _p(out))
    return out


def _run(fn, V, W, H, T, div, lamW, lamH, fixW, fixH, maxiter, tol):
    V, W, H = _f(V), _f(W), _f(H)
    m, n = V.shape
    K = H.shape[0]
    lamW, lamH = _vec(lamW, K, np.float64), _vec(lamH, K, np.float64)
    fixW, fixH = _vec(fixW, K, np.uint8), _vec(fixH, K, np.uint8)
    cost = np.zeros(maxiter)
    iters = C.c_int(0)
    args = [m, n, K] + ([T] if T is not None else []) + [
        _p(V), _p(W), _p(H), DIV[div], _p(lamW), _p(lamH), _p(fixW), _p(fixH), int(maxiter), C.c_double(tol), _p(cost), C.byref(iters)]
    rc = fn(*args)
    assert rc == 0
    return W, H, cost[: iters.value]


def nmf(V, W_init, H_init, div="euclidean", lamW=0.0, lamH=0.0, fixW=0, fixH=0, maxiter=100, tol=1e-3):
    return _run(lib().oracle_nmf, V, W_init, H_init, None, div, lamW, lamH, fixW, fixH, maxiter, tol)


def cnmf(V, W_init, H_init, div="euclidean", lamW=0.0, lamH=0.0, fixW=0, fixH=0, maxiter=100, tol=1e-3):
    W = _f(W_init)
    if W.ndim == 2:
        W = W.reshape(W.shape[0], W.shape[1], 1, order="F")
    T = W.shape[2]
    W, H, cost = _run(lib().oracle_cnmf, V, W, H_init, T, div, lamW, lamH, fixW, fixH, maxiter, tol)
    return W, H, cost


def projfunc(s, k1, k2, nn=True):
    s = np.ascontiguousarray(np.asarray(s, dtype=np.float64).reshape(-1))
    v = np.zeros_like(s)
    it = C.c_int(0)
    lib().oracle_projfunc(s.size, _p(s), 1, C.c_double(k1), C.c_double(k2), int(bool(nn)), _p(v), 1, C.byref(it))
    return v, it.value


def nmfsc(V, W_init, H_init, sW=0.0, sH=0.0, fixW=False, fixH=False, maxiter=100, tol=1e-3):
    V, W, H = _f(V), _f(W_init), _f(H_init)
    m, n = V.shape
    K = H.shape[0]
    cost = np.zeros(maxiter + 1)
    ncost = C.c_int(0)
    tH = np.zeros(maxiter, dtype=np.int32)
    tW = np.zeros(maxiter, dtype=np.int32)
    steps = np.zeros(2)
    rc = lib().oracle_nmfsc(m, n, K, _p(V), _p(W), _p(H), C.c_double(sW), C.c_double(sH), int(fixW), int(fixH),
                            int(maxiter), C.c_double(tol), _p(cost), C.byref(ncost), _p(tH), _p(tW), _p(steps))
    if rc == 1:
        raise ValueError("Negative values in data!")
    info = dict(triesH=[int(t) for t in tH if t > 0], triesW=[int(t) for t in tW if t > 0],
                stepsizeH=steps[0], stepsizeW=steps[1])
    return W, H, cost[: ncost.value], info


def lnmf(V, W_init, H_init, fixW=False, fixH=False, maxiter=100, tol=1e-3):
    V, W, H = _f(V), _f(W_init), _f(H_init)
    m, n = V.shape
    K = H.shape[0]
    cost = np.zeros(maxiter)
    it = C.c_int(0)
    lib().oracle_lnmf(m, n, K, _p(V), _p(W), _p(H), int(fixW), int(fixH), int(maxiter), C.c_double(tol), _p(cost), C.byref(it))
    return W, H, cost


def cnmfsc(V, W_init, H_init, sW=0.0, sH=0.0, fixW=False, fixH=False, maxiter=100, tol=1e-3):
    V, H = _f(V), _f(H_init)
    W = _f(W_init)
    if W.ndim == 2:
        W = W.reshape(W.shape[0], W.shape[1], 1, order="F")
    m, n = V.shape
    K, T = W.shape[1], W.shape[2]
    cost = np.zeros(maxiter + 1)
    ncost = C.c_int(0)
    tH = np.zeros(maxiter, dtype=np.int32)
    tW = np.zeros(maxiter * T, dtype=np.int32)
    rc = lib().oracle_cnmfsc(m, n, K, T, _p(V), _p(W), _p(H), C.c_double(sW), C.c_double(sH), int(fixW), int(fixH), int(maxiter), C.c_double(tol),
                             _p(cost), C.byref(ncost), _p(tH), _p(tW))
    if rc == 1:
        raise ValueError("Negative values in data!")
    return W, H, cost[: ncost.value], dict(triesH=[int(t) for t in tH if t > 0], triesW=[int(t) for t in tW if t > 0])


def constrainednmf_sorted(V_sorted, W_init, Z_init, seg, div="euclidean", lamW=0.0, lamZ=0.0, fixW=False, fixZ=False, maxiter=100, tol=1e-3):
    """constrainednmf.m:183-258 on label-sorted samples; `seg` = column ranges of the rows of A (see nmf_oracle.c).  Returns W, H (sorted), Z, cost."""
    V, W, Z = _f(V_sorted), _f(W_init), _f(Z_init)
    m, n = V.shape
    K, nz = Z.shape
    H = np.zeros((K, n), order="F")
    seg = np.ascontiguousarray(seg, dtype=np.int64)
    cost = np.zeros(maxiter)
    it = C.c_int(0)
    lib().oracle_constrainednmf(m, n, K, _p(V), _p(W), _p(Z), _p(H), seg.ctypes.data_as(C.c_void_p), int(nz), DIV[div], C.c_double(lamW), C.c_double(lamZ),
                                int(fixW), int(fixZ), int(maxiter), C.c_double(tol), _p(cost), C.byref(it))
    return W, H, Z, cost[: it.value]


def sort_dictionary_order(W):
    W = _f(W)
    m, K = W.shape
    order = np.zeros(K, dtype=np.int32)
    lib().oracle_sort_dictionary_order(m, K, _p(W), order.ctypes.data_as(C.c_void_p))
    return order
