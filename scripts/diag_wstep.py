"""Diagnostic (not part of the suite): where does the W error of an H-fixed euclidean run come from?  One engine, iterations driven phase by phase; after every
W-step partial the device's N = V*H' and G = H*H' are compared with float64 products of the SAME fp32 operands, and the device's W update is replayed on the host
in float64 from the device's own N, G and master copy.   python scripts/diag_wstep.py [m n K iters planted]"""
import ctypes as C
import sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
from conftest import synth, rel_fro
from nmf_toolbox_amd import _lib
from nmf_toolbox_amd.engine import Engine, colmajor_to_torch, torch_to_colmajor
from oracle import nmf_oracle as O

m, n, K, iters, planted = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), bool(int(sys.argv[5]))) if len(sys.argv) > 5 else (512, 2048, 256, 30, True)
V, W0, H0 = synth(m, n, K, planted=planted)
cfg = dict(divergence="euclidean", W_init=W0, H_init=H0, maxiter=iters, tolerance=1e-300, H_fixed=True)
trace = []
ref = O.nmf(V, K, cfg)
dev = "cuda:0"
e = Engine(colmajor_to_torch(V, dev), colmajor_to_torch(W0, dev), colmajor_to_torch(H0, dev), divergence="euclidean", use_dist=False, fixH=np.ones(K, np.uint8), path=2)
e.init()
lib = _lib.load()
w64p, h64p = C.c_void_p(), C.c_void_p()
_lib.check(lib.nmfx_engine_master_ptrs(e.h, C.byref(w64p), C.byref(h64p)))
EPS = 2.0 ** -52


import os
_hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))


def master_W():
    torch.cuda.synchronize()
    arr = np.zeros(m * K, dtype=np.float64)
    rc = _hip.hipMemcpy(C.c_void_p(arr.ctypes.data), w64p, C.c_size_t(8 * m * K), C.c_int(2))   # hipMemcpyDeviceToHost
    assert rc == 0, rc
    return arr.reshape(K, m).T.copy()


Vd = V.astype(np.float32).astype(np.float64)
Hd = torch_to_colmajor(e.H).astype(np.float64)      # fp32 H (fixed) as doubles
N_ref = Vd @ Hd.T
G_ref = Hd @ Hd.T
Wh = master_W()                                      # host replay starts from the device's normalised master
W_init_dev = Wh.copy()
print("init: W32 vs master %.2e, master vs oracle-normalised W0 %.2e" % (rel_fro(torch_to_colmajor(e.W), Wh), rel_fro(Wh, W0 / np.sqrt((W0 ** 2).sum(0)))))
for it in range(iters):
    e.wstep_partial()
    torch.cuda.synchronize()
    pk = e.packed.cpu().numpy().astype(np.float64)
    Nd = pk[: m * K].reshape(K, m).T
    Gd = pk[m * K: m * K + K * K].reshape(K, K).T
    e.wstep_finish()
    Wm = master_W()
    # host replay of nmf.m:149-150,168-169 in float64 from the device's N, G
    P = Wh @ Gd
    dn = (Wh * P).sum(0); dp = (Wh * Nd).sum(0)
    Wn = Wh * (Nd + Wh * dn) / np.fmax(P + Wh * dp, EPS)
    Wh = Wn / np.sqrt((Wn ** 2).sum(0))
    if it in (0, 1, 2, 5, 10, 20, iters - 1):
        print("it %2d  N err %.2e  G err %.2e  | device master vs host replay %.2e | W32 vs master %.2e" % (
            it, rel_fro(Nd, N_ref), rel_fro(Gd, G_ref), rel_fro(Wm, Wh), rel_fro(torch_to_colmajor(e.W), Wm)), flush=True)
    Wh = Wm   # follow the device
print("final: device W vs oracle %.2e" % rel_fro(torch_to_colmajor(e.W), ref[0]))
# the float64 algorithm on (a) exact products of the fp32 operands, (b) the device's N and G
for name, Nx, Gx in (("exact products of fp32 V, H", N_ref, G_ref), ("device N, exact G", Nd, G_ref), ("exact N, device G", N_ref, Gd), ("device N and G", Nd, Gd)):
    W = W0.astype(np.float32).astype(np.float64); W = W / np.sqrt((W ** 2).sum(0))
    for it in range(iters):
        P = W @ Gx
        dn = (W * P).sum(0); dp = (W * Nx).sum(0)
        Wn = W * (Nx + W * dn) / np.fmax(P + W * dp, EPS)
        W = Wn / np.sqrt((Wn ** 2).sum(0))
    print("float64 iteration on %-32s: W vs oracle %.2e" % (name, rel_fro(W, ref[0])))

for name, Wstart in (("device's initial master (fp32-normalised)", W_init_dev), ("exact W0 normalised in double", W0 / np.sqrt((W0 ** 2).sum(0))),
                     ("fl32(W0) normalised in double", None)):
    W = Wstart if Wstart is not None else (lambda x: x / np.sqrt((x ** 2).sum(0)))(W0.astype(np.float32).astype(np.float64))
    for it in range(iters):
        P = W @ Gd
        dn = (W * P).sum(0); dp = (W * Nd).sum(0)
        Wn = W * (Nd + W * dn) / np.fmax(P + W * dp, EPS)
        W = Wn / np.sqrt((Wn ** 2).sum(0))
    print("float64 iteration on device N and G from %-45s: W vs oracle %.2e" % (name, rel_fro(W, ref[0])))
