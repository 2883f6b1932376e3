"""Parity of the multi-GPU paths on REAL devices (scripts/first_8gpu_lease.sh; the suite's sharded tests name device 0 N times because this pool has 1-GPU
boxes): nmf (KL, euclidean, IS), cnmf (halo exchange), nmfsc (distributed projfunc) behind the blocking call with nmfx_gpus = [0 .. N-1], both exchanges,
against the unsharded call on device 0 (bit-identical W is NOT expected: the summation order differs; the contract is the oracle's 1e-5 / 1e-6) and against the
float64 oracle.     python scripts/multi_gpu_parity.py <n_gpus>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import rel_fro, synth  # noqa: E402
import nmf_toolbox_amd as A  # noqa: E402
from oracle import nmf_oracle as O  # noqa: E402  (test infrastructure: this is a checker script, not product code)

N = int(sys.argv[1]) if len(sys.argv) > 1 else A.device_count()
ONE = os.environ.get("NMFX_PARITY_ONE_DEVICE") == "1"   # dry run on a 1-GPU box: N shards on device 0 (peer exchange only: RCCL refuses duplicate devices)
bad = 0
for n_gpus in sorted({2, min(4, N), N}):
    if n_gpus > N or n_gpus < 2:
        continue
    ids = [0] * n_gpus if ONE else list(range(n_gpus))
    for name, run, ref in (
        ("nmf kl 1024x4096 K=256", lambda c: A.nmf(*c[0], dict(c[1], divergence="kl")), lambda c: O.nmf(*c[0], dict(c[1], divergence="kl"))),
        ("nmf euclidean 1024x4096 K=128", lambda c: A.nmf(*c[0], dict(c[1], divergence="euclidean")), lambda c: O.nmf(*c[0], dict(c[1], divergence="euclidean"))),
        ("nmf is 1024x4096 K=128", lambda c: A.nmf(*c[0], dict(c[1], divergence="is")), lambda c: O.nmf(*c[0], dict(c[1], divergence="is"))),
    ):
        K = 256 if "K=256" in name else 128
        V, W0, H0 = synth(1024, 4096, K)
        base = dict(W_init=W0, H_init=H0, maxiter=8, tolerance=1e-300)
        r = ref(((V, K), base))
        for be in (("peer",) if ONE else ("rccl", "peer")):
            g = run(((V, K), dict(base, nmfx_gpus=ids, nmfx_multi_backend=be)))
            e = dict(W=rel_fro(g[0], r[0]), H=rel_fro(g[1], r[1]), cost=rel_fro(g[2], r[2]))
            ok = e["W"] <= 1e-5 and e["H"] <= 1e-5 and e["cost"] <= (1e-5 if " is " in name else 1e-6)
            bad += not ok
            print("%-34s N=%d %-4s %s %s" % (name, n_gpus, be, "ok " if ok else "BAD", {k: "%.2e" % v for k, v in e.items()}), flush=True)
    V, W0, H0 = synth(512, 4096, 64, T=4)
    for div in ("euclidean", "kl"):
        base = dict(divergence=div, W_init=W0, H_init=H0, maxiter=6, tolerance=1e-300)
        r = O.cnmf(V, 64, 4, base)
        g = A.cnmf(V, 64, 4, dict(base, nmfx_gpus=ids))
        e = dict(W=rel_fro(g[0], r[0]), H=rel_fro(g[1], r[1]), cost=rel_fro(g[2], r[2]))
        ok = max(e["W"], e["H"]) <= 1e-5 and e["cost"] <= 1e-6
        bad += not ok
        print("%-34s N=%d      %s %s" % ("cnmf %s 512x4096 K=64 T=4 (halos)" % div, n_gpus, "ok " if ok else "BAD", {k: "%.2e" % v for k, v in e.items()}), flush=True)
    V, W0, H0 = synth(512, 4096, 64)
    base = dict(W_init=W0, H_init=H0, H_sparsity=0.5, maxiter=8, tolerance=1e-300)
    i0, i1 = {}, {}
    r = O.nmfsc(V, 64, base, info=i0)
    g = A.nmfsc(V, 64, dict(base, nmfx_gpus=ids), info=i1)
    e = dict(W=rel_fro(g[0], r[0]), H=rel_fro(g[1], r[1]), cost=rel_fro(g[2], r[2]))
    ok = max(e["W"], e["H"]) <= 1e-5 and e["cost"] <= 1e-6 and list(i1["triesH"]) == list(i0["triesH"])
    bad += not ok
    print("%-34s N=%d      %s %s tries %s / %s" % ("nmfsc 512x4096 K=64 sH=0.5", n_gpus, "ok " if ok else "BAD", {k: "%.2e" % v for k, v in e.items()}, list(i1["triesH"]), list(i0["triesH"])), flush=True)
print("multi-GPU parity on %d real devices: %s" % (N, "all inside the contract" if not bad else "%d case(s) OUTSIDE" % bad))
sys.exit(1 if bad else 0)
