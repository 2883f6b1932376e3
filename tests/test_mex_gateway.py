"""The MEX gateway matlab/nmfx_mex.c, compiled (-Wall -Wextra -Werror) against tests/mock_mex/ -- a minimal MEX runtime with the
documented signatures of the calls the gateway uses -- and EXECUTED: argument checking on the CPU, and (gpu) every gateway branch
against the ctypes path on the same inputs.  MATLAB itself is not in the image: this shows the gateway is well-formed C that drives
libnmfx correctly, not that MATLAB accepts it (INTEGRATION.md says so too)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, rel_fro, synth

CLS = dict(double=6, uint8=9, int32=12, int64=14)


@pytest.fixture(scope="module")
def mex(tmp_path_factory):
    from nmf_toolbox_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        from nmf_toolbox_amd import build
        build.build()
    d = tmp_path_factory.mktemp("mex")
    so = str(d / "nmfx_mex_mock.so")
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-shared", "-fPIC", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "tests", "mock_mex"),
                           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "matlab", "nmfx_mex.c"), os.path.join(ROOT, "tests", "mock_mex", "mock_mex.c"),
                           "-L", libdir, "-lnmfx", "-Wl,-rpath," + libdir, "-o", so])
    _lib.load()                                    # the HIP runtime torch ships, first (as _lib does)
    lib = C.CDLL(so)
    lib.mock_numeric.restype = C.c_void_p
    lib.mock_numeric.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.mock_string.restype = C.c_void_p
    lib.mock_string.argtypes = [C.c_char_p]
    lib.mock_struct.restype = C.c_void_p
    lib.mxSetField.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_void_p]
    lib.mock_call.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    lib.mock_error.restype = C.c_char_p
    lib.mxGetData.restype = C.c_void_p
    lib.mxGetData.argtypes = [C.c_void_p]
    lib.mxGetDimensions.restype = C.POINTER(C.c_size_t)
    lib.mxGetDimensions.argtypes = [C.c_void_p]
    lib.mxGetNumberOfDimensions.restype = C.c_size_t
    lib.mxGetNumberOfDimensions.argtypes = [C.c_void_p]
    lib.mxGetField.restype = C.c_void_p
    lib.mxGetField.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p]
    lib.mock_class.argtypes = [C.c_void_p]
    return lib


class Mex:
    def __init__(self, lib):
        self.lib = lib

    def arr(self, a, cls="double"):
        a = np.asfortranarray(np.asarray(a, dtype={"double": np.float64, "uint8": np.uint8, "int32": np.int32, "int64": np.int64}[cls]))
        if a.ndim < 2:
            a = a.reshape(1, -1, order="F")
        dims = (C.c_size_t * a.ndim)(*a.shape)
        return self.lib.mock_numeric(CLS[cls], a.ndim, dims, a.ctypes.data_as(C.c_void_p))

    def struct(self, **fields):
        s = self.lib.mock_struct()
        for k, v in fields.items():
            self.lib.mxSetField(s, 0, k.encode(), v)
        return s

    def call(self, nlhs, *args):
        prhs = (C.c_void_p * len(args))(*[self.lib.mock_string(a.encode()) if isinstance(a, str) else a for a in args])
        plhs = (C.c_void_p * max(nlhs, 1))()
        rc = self.lib.mock_call(nlhs, plhs, len(args), prhs)
        if rc:
            raise RuntimeError(self.lib.mock_error().decode())
        return [self.get(plhs[i]) for i in range(max(nlhs, 1))]

    def get(self, a):
        if not a:
            return None
        cls = self.lib.mock_class(a)
        if cls == 2:   # struct
            out = {}
            for f in ("iters_run", "stepsize_H", "stepsize_W", "converged_early", "tries_H", "tries_W"):
                v = self.lib.mxGetField(a, 0, f.encode())
                out[f] = self.get(v) if v else None
            return out
        nd = self.lib.mxGetNumberOfDimensions(a)
        dims = [self.lib.mxGetDimensions(a)[i] for i in range(nd)]
        dt = {6: np.float64, 12: np.int32, 14: np.int64, 9: np.uint8}[cls]
        n = int(np.prod(dims))
        if n == 0:
            return np.zeros(dims, dtype=dt)
        buf = (C.c_char * (n * np.dtype(dt).itemsize)).from_address(self.lib.mxGetData(a))
        return np.frombuffer(buf, dtype=dt).reshape(dims, order="F").copy()


def test_gateway_compiles_and_checks_arguments(mex):
    M = Mex(mex)
    V, W0, H0 = synth(16, 24, 3)
    opts = M.struct(maxiter=M.arr([5.0]))
    with pytest.raises(RuntimeError, match="nmfx:usage"):
        M.call(3, "nmf", M.arr(V))
    with pytest.raises(RuntimeError, match="nmfx:algo: unknown algorithm bogus"):
        M.call(1, "bogus", M.arr(V))
    with pytest.raises(RuntimeError, match="K_s must be a non-empty int32 vector"):
        M.call(3, "nmf", M.arr(V), M.arr(W0), M.arr(H0), M.arr([3.0]), M.arr([1.0]), opts)           # K_s as double: refused, not reinterpreted
    with pytest.raises(RuntimeError, match="W_init must be 16 x 3 x 1"):
        M.call(3, "nmf", M.arr(V), M.arr(W0[:, :2]), M.arr(H0), M.arr([3], "int32"), M.arr([1.0]), opts)
    with pytest.raises(RuntimeError, match="W_fixed / H_fixed uint8"):
        M.call(3, "nmf", M.arr(V), M.arr(W0), M.arr(H0), M.arr([3], "int32"), M.arr([1.0]), M.struct(W_fixed=M.arr([1.0])))
    with pytest.raises(RuntimeError, match="at most four outputs"):
        M.call(5, "nmf", M.arr(V), M.arr(W0), M.arr(H0), M.arr([3], "int32"), M.arr([1.0]), opts)
    with pytest.raises(RuntimeError, match="segments must be an int64 vector"):
        M.call(3, "constrainednmf", M.arr(V), M.arr(W0), M.arr(H0), M.arr([0.0, 24.0]), opts)
    import nmf_toolbox_amd as A
    if A.device_count() == 0:          # no GPU here: the library's loud failure comes back as a MATLAB error, with its text
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            M.call(3, "nmf", M.arr(V), M.arr(W0), M.arr(H0), M.arr([3], "int32"), M.arr([1.0]), opts)
    mex.mock_reset()


@pytest.mark.gpu
def test_gateway_runs_every_branch_like_the_ctypes_path(mex, gpu_lib):
    M = Mex(mex)
    V, W0, H0 = synth(96, 160, 6, T=3)
    W2 = W0[:, :, 0]
    o = lambda **kw: M.struct(**{k: (v if isinstance(v, int) and v > 1000 else M.arr(*v) if isinstance(v, tuple) else M.arr([float(v)])) for k, v in kw.items()})
    # nmf (kl, two sources with sparsity / fixed flags)
    W, H, cost, info = M.call(4, "nmf", M.arr(V), M.arr(W2), M.arr(H0), M.arr([2, 4], "int32"), M.arr([1.0]),
                              o(divergence=1, maxiter=8, tolerance=1e-12, W_sparsity=([0.1, 0.0],), H_fixed=([0, 1], "uint8")))
    Wr, Hr, cr = gpu_lib.nmf(V, [2, 4], dict(divergence="kl", W_init=[W2[:, :2], W2[:, 2:]], H_init=[H0[:2], H0[2:]], W_sparsity=[0.1, 0.0],
                                             H_fixed=[False, True], maxiter=8, tolerance=1e-12))
    assert np.array_equal(W, np.hstack(Wr)) and np.array_equal(H, np.vstack(Hr)) and np.array_equal(cost.ravel(), cr) and info["iters_run"][0, 0] == 8
    # the same on three shards of one GPU (device_ids)
    Wm, Hm, cm = M.call(3, "nmf", M.arr(V), M.arr(W2), M.arr(H0), M.arr([6], "int32"), M.arr([1.0]), o(divergence=1, maxiter=5, tolerance=1e-12, device_ids=([0, 0, 0], "int32")))
    Wr, Hr, cr = gpu_lib.nmf(V, 6, dict(divergence="kl", W_init=W2, H_init=H0, maxiter=5, tolerance=1e-12, nmfx_gpus=[0, 0, 0]))
    assert np.array_equal(Wm, Wr) and np.array_equal(Hm, Hr) and np.array_equal(cm.ravel(), cr)
    # cnmf (3-D W), stop rule trims the cost vector
    W, H, cost = M.call(3, "cnmf", M.arr(V), M.arr(W0), M.arr(H0), M.arr([6], "int32"), M.arr([3.0]), o(divergence=0, maxiter=40, tolerance=5.0))
    Wr, Hr, cr = gpu_lib.cnmf(V, 6, 3, dict(W_init=W0, H_init=H0, maxiter=40, tolerance=5.0))
    assert W.shape == (96, 6, 3) and np.array_equal(W, Wr) and np.array_equal(H, Hr) and cost.shape == (len(cr), 1) and len(cr) < 40
    # cnmf on two shards of one GPU (device_ids): halo columns between the shards
    Wm, Hm, cm = M.call(3, "cnmf", M.arr(V), M.arr(W0), M.arr(H0), M.arr([6], "int32"), M.arr([3.0]), o(divergence=1, maxiter=5, tolerance=1e-12, device_ids=([0, 0], "int32")))
    Wr, Hr, cr = gpu_lib.cnmf(V, 6, 3, dict(divergence="kl", W_init=W0, H_init=H0, maxiter=5, tolerance=1e-12, nmfx_gpus=[0, 0]))
    assert np.array_equal(Wm, Wr) and np.array_equal(Hm, Hr) and np.array_equal(cm.ravel(), cr)
    # lnmf
    W, H, cost = M.call(3, "lnmf", M.arr(V), M.arr(W2), M.arr(H0), M.arr([6], "int32"), M.arr([1.0]), o(divergence=1, maxiter=6, tolerance=1e-12))
    Wr, Hr, cr = gpu_lib.lnmf(V, 6, dict(W_init=W2, H_init=H0, maxiter=6, tolerance=1e-12))
    assert np.array_equal(W, Wr) and np.array_equal(cost.ravel(), cr)
    # nmfsc / cnmfsc with the line-search bookkeeping
    i1 = {}
    W, H, cost, info = M.call(4, "nmfsc", M.arr(V), M.arr(W2), M.arr(H0), M.arr([6], "int32"), M.arr([1.0]), o(maxiter=6, tolerance=1e-12, sc_H_sparsity=0.5, sc_W_sparsity=0.3))
    Wr, Hr, cr = gpu_lib.nmfsc(V, 6, dict(W_init=W2, H_init=H0, maxiter=6, tolerance=1e-12, H_sparsity=0.5, W_sparsity=0.3), info=i1)
    assert np.array_equal(W, Wr) and np.array_equal(H, Hr) and np.array_equal(cost.ravel(), cr) and cost.shape[0] == 7
    assert [t for t in info["tries_H"].ravel() if t > 0] == i1["triesH"] and [t for t in info["tries_W"].ravel() if t > 0] == i1["triesW"]
    assert info["stepsize_H"][0, 0] == i1["stepsizeH"]
    # nmfsc on two shards of one GPU (one host thread per shard inside the library)
    i2 = {}
    W, H, cost, info = M.call(4, "nmfsc", M.arr(V), M.arr(W2), M.arr(H0), M.arr([6], "int32"), M.arr([1.0]),
                              o(maxiter=6, tolerance=1e-12, sc_H_sparsity=0.5, sc_W_sparsity=0.3, device_ids=([0, 0], "int32")))
    Wr, Hr, cr = gpu_lib.nmfsc(V, 6, dict(W_init=W2, H_init=H0, maxiter=6, tolerance=1e-12, H_sparsity=0.5, W_sparsity=0.3, nmfx_gpus=[0, 0]), info=i2)
    assert np.array_equal(W, Wr) and np.array_equal(H, Hr) and np.array_equal(cost.ravel(), cr)
    assert [t for t in info["tries_H"].ravel() if t > 0] == i2["triesH"] == i1["triesH"]
    W, H, cost, info = M.call(4, "cnmfsc", M.arr(V), M.arr(W0), M.arr(H0), M.arr([6], "int32"), M.arr([3.0]), o(maxiter=4, tolerance=1e-12, sc_H_sparsity=0.5))
    Wr, Hr, cr = gpu_lib.cnmfsc(V, 6, 3, dict(W_init=W0, H_init=H0, maxiter=4, tolerance=1e-12, H_sparsity=0.5), info=i1)
    assert np.array_equal(W, Wr) and np.array_equal(H, Hr) and np.array_equal(cost.ravel(), cr) and info["tries_W"].shape == (12, 1)
    # constrainednmf: the wrapper's label bookkeeping is host logic; here two classes of 80 columns each
    seg = np.array([0, 80, 160], dtype=np.int64)
    Z0 = H0[:, :2]
    W, H, cost, Z = M.call(4, "constrainednmf", M.arr(V), M.arr(W2), M.arr(Z0), M.arr(seg, "int64"), o(divergence=1, maxiter=5, tolerance=1e-12))
    lab = np.r_[np.zeros(80, dtype=int), np.ones(80, dtype=int)]
    Wr, Hr, Zr, Ar, cr = gpu_lib.constrainednmf(V, lab, 6, dict(divergence="kl", W_init=W2, Z_init=Z0, maxiter=5, tolerance=1e-12))
    assert np.array_equal(W, Wr) and np.array_equal(Z, Zr) and np.array_equal(H, Hr) and np.array_equal(cost.ravel(), cr)
    # helpers
    (Vh,) = M.call(1, "reconstruct", M.arr(W0), M.arr(H0))
    assert np.array_equal(Vh, gpu_lib.ReconstructFromDecomposition(W0, H0))
    s = np.abs(np.random.RandomState(3).randn(500, 4))
    v, its = M.call(2, "projfunc", M.arr(s), M.arr([9.0]), M.arr([1.0]), M.arr([1.0]))
    for c in range(4):
        vr, ir = gpu_lib.projfunc(s[:, c], 9.0, 1.0, True)
        assert np.array_equal(v[:, c], vr) and its[c, 0] == ir
    Ws, Hs, order = M.call(3, "sortdictionary", M.arr(W2), M.arr(H0))
    Wr, Hr = gpu_lib.SortDictionary(W2, H0)
    assert np.array_equal(Ws, Wr) and np.array_equal(Hs, Hr) and sorted(order.ravel().tolist()) == list(range(6))
    with pytest.raises(RuntimeError, match="nmfx:error: alpha = 0 and beta = 0"):          # the library's message, as MATLAB error() text
        M.call(3, "nmf", M.arr(V), M.arr(W2), M.arr(H0), M.arr([6], "int32"), M.arr([1.0]), o(divergence=3, alpha=0, beta=0, maxiter=2))
    mex.mock_reset()


def test_m_wrappers_call_commands_the_gateway_implements():
    """static check of matlab/*.m (source only: no MATLAB here): every wrapper calls nmfx_mex with a command string that
    mexFunction dispatches on and with the number of arguments that branch demands (nrhs checks of nmfx_mex.c)"""
    import glob
    import re
    here = os.path.dirname(os.path.abspath(__file__))
    mdir = os.path.join(os.path.dirname(here), "matlab")
    gateway = open(os.path.join(mdir, "nmfx_mex.c")).read()
    known = set(re.findall(r'!strcmp\(algo, "(\w+)"\)', gateway))
    nrhs = {"nmf": 7, "cnmf": 7, "lnmf": 7, "nmfsc": 7, "cnmfsc": 7, "constrainednmf": 6, "reconstruct": 3, "projfunc": 5, "sortdictionary": 3}
    assert known == set(nrhs)
    wrappers = sorted(glob.glob(os.path.join(mdir, "nmfx_*.m")))
    assert {os.path.basename(w)[5:-2].lower() for w in wrappers} == {"nmf", "cnmf", "lnmf", "nmfsc", "cnmfsc", "constrainednmf", "reconstructfromdecomposition",
                                                                      "projfunc", "sortdictionary"}
    seen = set()
    for w in wrappers:
        src = open(w).read()
        calls = []
        for mt in re.finditer(r"nmfx_mex\('(\w+)'", src):          # (the header comments do not quote the command)
            depth, nargs, i = 1, 1, mt.end()
            while depth > 0:             # count the top-level commas up to the matching parenthesis (the command string is argument 1)
                ch = src[i]
                depth += ch in "([{"
                depth -= ch in ")]}"
                nargs += ch == "," and depth == 1
                i += 1
            calls.append((mt.group(1), nargs))
        assert calls, w
        for cmd, nargs in calls:
            assert cmd in known, (w, cmd)
            assert nargs == nrhs[cmd], (w, cmd, nargs)
            seen.add(cmd)
        assert src.lstrip().startswith("function"), w
    assert seen == known
