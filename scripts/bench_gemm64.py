"""time nmfx_gemm64 at the shapes the engine calls it with (P = W*(H*H'): C2 8192 x 128 x 128, C4 4096 x 512 x 512, C3-euclidean 16384 x 256 x 256) and check it
against torch float64:   python scripts/bench_gemm64.py"""
import sys
import torch
sys.path.insert(0, "/root/repo")
from nmf_toolbox_amd import _lib
lib = _lib.load()
for M, N, Kc in [(8192, 128, 128), (4096, 512, 512), (16384, 256, 256), (1000, 77, 300)]:
    A = torch.rand(Kc, M, dtype=torch.float64, device="cuda") - 0.3      # column-major M x Kc
    B = (torch.rand(N, Kc, dtype=torch.float32, device="cuda") - 0.3)    # column-major Kc x N
    C = torch.zeros(N, M, dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    run = lambda: _lib.check(lib.nmfx_gemm64(st, M, N, Kc, A.data_ptr(), None, M, None, B.data_ptr(), Kc, C.data_ptr(), None, M))
    for _ in range(5): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    ref = (B.double() @ A)          # (N x Kc) @ (Kc x M) = C' (N x M row-major = column-major M x N)
    err = float((C - ref).norm() / ref.norm())
    us = e0.elapsed_time(e1) / 50 * 1e3
    print("gemm64 %5d x %3d x %3d: %7.1f us  %5.1f TFLOP/s fp64  rel err %.1e" % (M, N, Kc, us, 2.0 * M * N * Kc / us / 1e6, err))
