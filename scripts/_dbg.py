import sys, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from conftest import synth, rel_fro
from nmf_toolbox_amd.engine import Engine, colmajor_to_torch, torch_to_colmajor
from oracle import nmf_oracle as O
m, n, K, T = 256, 512, 64, 4
V, W0, H0 = synth(m, n, K, T=T)
dev = "cuda:0"
Vt, Ht = colmajor_to_torch(V, dev), colmajor_to_torch(H0, dev)
Wt = torch.from_numpy(np.ascontiguousarray(W0.transpose(2, 1, 0), dtype=np.float32)).to(dev).reshape(T * K, m)
e = Engine(Vt, Wt, Ht, divergence="kl", T=T, algorithm="cnmf", path=2)
e.init()
print("path kind", e.path_kind)
e.wstep_partial()
torch.cuda.synchronize()
R = e.workspace[: m * n * 4].view(torch.float32).reshape(n, m).cpu().numpy().T.astype(np.float64)
W = e.W.cpu().numpy().reshape(T, K, m).transpose(2, 1, 0).astype(np.float64)
H = e.H.cpu().numpy().T.astype(np.float64)
Vh = O.reconstruct_from_decomposition(W, H)
Rref = V / Vh
bad = np.abs(R - Rref) > 1e-4 * np.abs(Rref)
print("R rel err", rel_fro(R, Rref), "bad fraction", bad.mean())
rows, cols = np.where(bad)
print("bad rows (unique, first 20):", np.unique(rows)[:20], "count", len(np.unique(rows)))
print("bad cols (unique, first 40):", np.unique(cols)[:40], "count", len(np.unique(cols)))
np.set_printoptions(precision=4, linewidth=200)
print("R   [0, 0:12]", R[0, 0:12])
print("Rref[0, 0:12]", Rref[0, 0:12])
print("Vh  [0, 0:12]", Vh[0, 0:12])
print("V   [0, 0:12]", V[0, 0:12])
print("log2(Rref)", np.log2(Rref[0, 0:12]))
