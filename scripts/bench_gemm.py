#!/usr/bin/env python3
"""Dev aid: time the generic MFMA GEMM (nmfx_gemm_f32, plain views) on the shapes the cnmf / generic paths issue.
    python scripts/bench_gemm.py [M N Kc opA opB] ..."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmf_toolbox_amd import _lib  # noqa: E402

SHAPES = [  # (M, N, Kc, opA, opB, note)      opA: 0 = A stored M x Kc (RC), 1 = stored Kc x M (KC); opB: 0 = stored Kc x N (KC), 1 = stored N x Kc (RC)
    (4096, 512, 16384, 0, 1, "c4 V*Hs'"),
    (4096, 16384, 512, 0, 0, "c4 W*Hs"),
    (64, 16384, 32768, 1, 0, "c4 W'*X"),
    (16384, 65536, 256, 0, 0, "c3 W*H"),
    (16384, 256, 65536, 0, 1, "c3 V*H'"),
    (256, 65536, 16384, 1, 0, "c3 W'*V"),
    (8192, 8192, 8192, 0, 0, "square"),
]


def run(lib, M, N, Kc, opA, opB, note, reps=5):
    dev = "cuda:0"
    A = torch.rand((M * Kc,), device=dev)
    B = torch.rand((Kc * N,), device=dev)
    Cm = torch.zeros((M * N,), device=dev)
    ws = torch.empty((1 << 30,), dtype=torch.uint8, device=dev)
    lda = M if opA == 0 else Kc
    ldb = Kc if opB == 0 else N
    st = torch.cuda.current_stream().cuda_stream

    def call():
        _lib.check(lib.nmfx_gemm_f32(st, opA, opB, M, N, Kc, A.data_ptr(), None, lda, 0, B.data_ptr(), None, ldb, 0, Cm.data_ptr(), M, 0, ws.data_ptr(), ws.numel()))

    call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("%-12s M=%6d N=%6d Kc=%6d opA=%d opB=%d  %8.3f ms  %7.1f TFLOP/s" % (note, M, N, Kc, opA, opB, ms, 2.0 * M * N * Kc / ms / 1e9), flush=True)


def main():
    lib = _lib.load()
    a = sys.argv[1:]
    shapes = SHAPES if not a else [tuple(int(x) for x in a[i:i + 5]) + ("cli",) for i in range(0, len(a), 5)]
    for s in shapes:
        run(lib, *s)


if __name__ == "__main__":
    main()
