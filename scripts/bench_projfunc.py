#!/usr/bin/env python3
"""Time the Hoyer projection kernel on device-resident rows (BASELINE config 5: 128 rows of H, n = 32768, sparseness 0.5)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmf_toolbox_amd import _lib  # noqa: E402

lib = _lib.load()
count, N = (128, 32768) if len(sys.argv) < 3 else (int(sys.argv[1]), int(sys.argv[2]))
sp = 0.5
k1 = np.sqrt(N) - (np.sqrt(N) - 1) * sp
g = torch.Generator(device="cuda:0"); g.manual_seed(0)
S = torch.rand((count, N), generator=g, device="cuda:0")
X = torch.empty_like(S)
its = torch.zeros(count, dtype=torch.int32, device="cuda:0")
st = torch.cuda.current_stream().cuda_stream
def run():
    _lib.check(lib.nmfx_projfunc_dev(st, X.data_ptr(), N, count, k1, 1.0, 1, S.data_ptr(), None, 0.0, its.data_ptr()))
run(); torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    run()
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 20
print("projfunc %d x %d: %.4f ms per call, inner iterations %s, %.1f GB/s of 8*N*count bytes" % (count, N, ms, sorted(set(its.tolist())), 8.0 * N * count / ms / 1e6))
