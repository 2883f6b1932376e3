"""Kernel-level parity (-m gpu): each HIP kernel against NumPy float64 on the same inputs, through the C ABI."""
import ctypes as C

import numpy as np
import pytest

from conftest import rel_fro

pytestmark = pytest.mark.gpu


def _gemm(lib, torch, opA, opB, M, N, Kc, A, B, A2=None, B2=None, proA=0, proB=0, accumulate=0, C0=None):
    """A, B given in their STORED MATLAB shapes; returns C (M x N) as float64 NumPy."""
    from nmf_toolbox_amd.engine import colmajor_to_torch, torch_to_colmajor
    from nmf_toolbox_amd import _lib
    dev = "cuda:0"
    tA, tB = colmajor_to_torch(A, dev), colmajor_to_torch(B, dev)
    tA2 = colmajor_to_torch(A2, dev) if A2 is not None else None
    tB2 = colmajor_to_torch(B2, dev) if B2 is not None else None
    tC = colmajor_to_torch(C0 if C0 is not None else np.zeros((M, N)), dev)
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.nmfx_gemm_f32(st, opA, opB, M, N, Kc, tA.data_ptr(), tA2.data_ptr() if tA2 is not None else None, A.shape[0], proA,
                                 tB.data_ptr(), tB2.data_ptr() if tB2 is not None else None, B.shape[0], proB, tC.data_ptr(), M, accumulate,
                                 ws.data_ptr(), ws.numel()))
    torch.cuda.synchronize()
    return torch_to_colmajor(tC)


@pytest.mark.parametrize("M,N,Kc", [(128, 128, 64), (256, 384, 96), (64, 256, 4096), (512, 64, 2048), (100, 37, 53), (129, 131, 33), (1, 1, 1), (8192, 128, 512)])
@pytest.mark.parametrize("opA,opB", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_gemm_plain(gpu_lib, M, N, Kc, opA, opB):
    import torch
    from nmf_toolbox_amd import _lib
    lib = _lib.load()
    rs = np.random.RandomState(M + 7 * N + 13 * Kc + opA + 2 * opB)
    Aop = rs.rand(M, Kc) - 0.3
    Bop = rs.rand(Kc, N) - 0.3
    A = Aop if opA == 0 else Aop.T.copy()
    B = Bop if opB == 0 else Bop.T.copy()
    got = _gemm(lib, torch, opA, opB, M, N, Kc, A, B)
    ref = Aop.astype(np.float32).astype(np.float64) @ Bop.astype(np.float32).astype(np.float64)
    assert rel_fro(got, ref) < 2e-6  # fp32 fmaf chains, contraction up to 4096


@pytest.mark.parametrize("M,N,Kc,acc", [(37, 21, 40000, 0), (37, 21, 40000, 1), (128, 128, 32768, 0), (64, 512, 16384, 1), (5, 5, 100000, 0), (130, 126, 20000, 1)])
def test_gemm_many_slabs_small_output(gpu_lib, M, N, Kc, acc):
    """small outputs over long contractions -- the K x K Gram products of a euclidean iteration: split into many slabs and summed by ONE launch in which 16 threads
    share an output quad (aux.hip::reduce_slabs_lanes_kernel).  Odd element counts take its scalar path, `accumulate` its read-modify-write."""
    import torch
    from nmf_toolbox_amd import _lib
    lib = _lib.load()
    rs = np.random.RandomState(M + N + Kc + acc)
    A, B = rs.rand(M, Kc) - 0.3, rs.rand(Kc, N) - 0.3
    C0 = rs.rand(M, N) if acc else None
    got = _gemm(lib, torch, 0, 0, M, N, Kc, A, B, accumulate=acc, C0=C0)
    ref = A.astype(np.float32).astype(np.float64) @ B.astype(np.float32).astype(np.float64) + (C0.astype(np.float32).astype(np.float64) if acc else 0.0)
    assert rel_fro(got, ref) < 3e-6


def test_gemm_is_transpose_safe(gpu_lib):
    """A = I with an ASYMMETRIC B catches a swapped C write (guide rule 16)."""
    import torch
    from nmf_toolbox_amd import _lib
    lib = _lib.load()
    n = 128
    B = np.arange(n * n, dtype=np.float64).reshape(n, n) / 100.0
    got = _gemm(lib, torch, 0, 0, n, n, n, np.eye(n), B)
    assert np.array_equal(got.astype(np.float32), B.astype(np.float32))


@pytest.mark.parametrize("pro", [1, 2, 3, 4])
def test_gemm_prologue_and_accumulate(gpu_lib, pro):
    import torch
    from nmf_toolbox_amd import _lib
    lib = _lib.load()
    rs = np.random.RandomState(pro)
    M, N, Kc = 256, 128, 320
    X, X2 = rs.rand(M, Kc) + 0.1, rs.rand(M, Kc) + 0.1
    B = rs.rand(Kc, N)
    C0 = rs.rand(M, N)
    f = {1: X / X2, 2: X / X2 ** 2, 3: 1.0 / X2, 4: X2 - X}[pro]
    got = _gemm(lib, torch, 0, 0, M, N, Kc, X, B, A2=X2, proA=pro, accumulate=1, C0=C0)
    assert rel_fro(got, f @ B + C0) < 2e-6
    # same element map on the B operand
    Y, Y2 = rs.rand(Kc, N) + 0.1, rs.rand(Kc, N) + 0.1
    A = rs.rand(M, Kc)
    g = {1: Y / Y2, 2: Y / Y2 ** 2, 3: 1.0 / Y2, 4: Y2 - Y}[pro]
    got = _gemm(lib, torch, 0, 0, M, N, Kc, A, Y, B2=Y2, proB=pro)
    assert rel_fro(got, A @ g) < 2e-6


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("N,count,sparse", [(200, 3, 0.6), (1024, 8, 0.5), (1025, 2, 0.5), (4096, 4, 0.8), (5000, 2, 0.3), (16384, 3, 0.5), (16385, 2, 0.6),
                                            (24576, 2, 0.4), (32768, 2, 0.5), (32769, 1, 0.5), (40000, 1, 0.5), (70000, 2, 0.7), (200001, 1, 0.4)])
def test_projfunc(gpu_lib, N, count, sparse, dtype):
    """every storage variant of the kernel (registers / registers + LDS / global scratch) against the float64 oracle: identical
    iteration counts and zero sets; float64 input is projected in float64 end to end, float32 input is its exact widening"""
    from oracle import nmf_oracle as O
    import ctypes as C
    from nmf_toolbox_amd import _lib
    lib = _lib.load()
    rs = np.random.RandomState(N + count)
    S = np.abs(rs.randn(count, N))
    if dtype == "f32":
        S = S.astype(np.float32)
    k1 = np.sqrt(N) - (np.sqrt(N) - 1) * sparse
    out = np.zeros_like(S)
    its = np.zeros(count, dtype=np.int32)
    _lib.check(lib.nmfx_projfunc(N, count, _lib.F64 if dtype == "f64" else _lib.F32, S.ctypes.data_as(C.c_void_p), k1, 1.0, 1,
                                 out.ctypes.data_as(C.c_void_p), its.ctypes.data_as(C.c_void_p), 0))
    tol = 1e-12 if dtype == "f64" else 1e-6
    for c in range(count):
        v, it = O.projfunc(S[c].astype(np.float64), k1, 1.0, True)
        assert it == its[c]
        assert rel_fro(out[c], v) < tol
        assert abs(out[c].sum(dtype=np.float64) - k1) < 1e-4 * k1 and abs((out[c].astype(np.float64) ** 2).sum() - 1.0) < 1e-5 and out[c].min() >= 0
        assert np.array_equal(out[c] == 0, v == 0)   # identical zero set (discrete branch)


def test_projfunc_signed(gpu_lib):
    from oracle import nmf_oracle as O
    s = np.random.RandomState(5).randn(300)
    v, it = gpu_lib.projfunc(s, 6.0, 1.0, False)
    v0, it0 = O.projfunc(s, 6.0, 1.0, False)
    assert it == it0 and rel_fro(v, v0) < 1e-12


@pytest.mark.parametrize("M,N,Kc", [(64, 64, 16), (128, 192, 64), (358, 640, 640), (8192, 128, 128), (1000, 37, 53), (1, 1, 1), (65, 129, 17), (4096, 512, 512),
                                    (300, 77, 112), (1000, 130, 256), (257, 33, 16), (2111, 512, 496)])   # (the last four: ragged edges of the panel kernel)
@pytest.mark.parametrize("a64,b64", [(True, False), (False, False), (True, True), (False, True)])
def test_gemm64(gpu_lib, M, N, Kc, a64, b64):
    """the float64 matrix-core product behind P = W*(H*H') (gemm64.hip): every operand-type combination against NumPy float64, float64 and fp32 results;
    an asymmetric B with A = I would expose a transposed write, the relative error the accumulation type"""
    import torch
    from nmf_toolbox_amd import _lib
    lib = _lib.load()
    rs = np.random.RandomState(M + 3 * N + 7 * Kc)
    A = rs.rand(M, Kc) - 0.3
    B = rs.rand(Kc, N) - 0.3
    if not a64:
        A = A.astype(np.float32).astype(np.float64)
    if not b64:
        B = B.astype(np.float32).astype(np.float64)
    dev = "cuda:0"
    cm = lambda X, dt: torch.from_numpy(np.ascontiguousarray(X.T)).to(dt).to(dev)   # column-major image
    tA, tB = cm(A, torch.float64 if a64 else torch.float32), cm(B, torch.float64 if b64 else torch.float32)
    C64 = torch.zeros(N, M, dtype=torch.float64, device=dev)
    C32 = torch.zeros(N, M, dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.nmfx_gemm64(st, M, N, Kc, tA.data_ptr() if a64 else None, None if a64 else tA.data_ptr(), M, tB.data_ptr() if b64 else None,
                               None if b64 else tB.data_ptr(), Kc, C64.data_ptr(), C32.data_ptr(), M))
    torch.cuda.synchronize()
    ref = A @ B
    assert rel_fro(C64.cpu().numpy().T, ref) < 1e-14
    assert rel_fro(C32.cpu().numpy().T, ref) < 1e-7


def test_gemm64_is_transpose_safe(gpu_lib):
    import torch
    from nmf_toolbox_amd import _lib
    lib = _lib.load()
    n = 96
    B = np.arange(n * n, dtype=np.float64).reshape(n, n) / 7.0
    tA = torch.eye(n, dtype=torch.float64, device="cuda:0")
    tB = torch.from_numpy(np.ascontiguousarray(B.T)).to("cuda:0")
    C64 = torch.zeros(n, n, dtype=torch.float64, device="cuda:0")
    _lib.check(lib.nmfx_gemm64(torch.cuda.current_stream().cuda_stream, n, n, n, tA.data_ptr(), None, n, tB.data_ptr(), None, n, C64.data_ptr(), None, n))
    torch.cuda.synchronize()
    assert np.array_equal(C64.cpu().numpy().T, B)
