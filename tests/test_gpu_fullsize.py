"""-m gpu: BASELINE.json's full sizes, checked through size-independent properties (the float64 oracle would need minutes to
hours there): the two independent HIP paths (fused / Gram vs materialised V_hat) agree, cost is non-increasing, the
normalisation invariants of nmf.m:169 / cnmf.m:196-199 hold, Hoyer sparseness after projection is exact."""
import numpy as np
import pytest

from conftest import EPS

pytestmark = pytest.mark.gpu


def _rand(torch, shape, seed):
    g = torch.Generator(device="cuda:0")
    g.manual_seed(seed)
    return torch.rand(shape, generator=g, device="cuda:0", dtype=torch.float32).clamp_(min=EPS)


def _run(torch, V, W0, H0, iters, **kw):
    from nmf_toolbox_amd.engine import Engine
    e = Engine(V, W0.clone(), H0.clone(), use_dist=False, **kw)
    e.init()
    c = torch.zeros(iters, dtype=torch.float64, device="cuda:0")
    e.iterate(iters, c)
    torch.cuda.synchronize()
    return e, c.cpu().numpy()


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


@pytest.mark.parametrize("name,div,m,n,K", [("c2", "euclidean", 8192, 32768, 128), ("c3", "kl", 16384, 65536, 256)])
def test_nmf_full_size_paths_agree(gpu_lib, name, div, m, n, K):
    import torch
    V, W0, H0 = _rand(torch, (n, m), 1000), _rand(torch, (K, m), 1), _rand(torch, (n, K), 2)
    fused, cf = _run(torch, V, W0, H0, 3, divergence=div, path=2)
    assert fused.cost_lags
    Wf, Hf = fused.W.clone(), fused.H.clone()
    nrm = (Wf.double() ** 2).sum(dim=1).sqrt()
    assert float((nrm - 1).abs().max()) < 1e-5                      # unit-L2 columns (nmf.m:169)
    assert np.all(np.diff(cf) < 0)                                   # cost strictly decreasing on random data
    assert float(Wf.min()) >= 0 and float(Hf.min()) >= 0
    fused.close()
    del fused
    torch.cuda.empty_cache()
    gen, cg = _run(torch, V, W0, H0, 3, divergence=div, path=1)     # materialised V_hat, separate GEMMs
    assert not gen.cost_lags
    assert _rel(Wf, gen.W) < 1e-5 and _rel(Hf, gen.H) < 1e-5
    assert np.allclose(cf, cg, rtol=2e-6)


@pytest.mark.parametrize("div", ["kl", "euclidean"])
def test_nmf_beyond_4GiB_paths_agree(gpu_lib, div):
    """V = 16384 x 131072 fp32 = 8 GiB, 2^31 elements: past every 32-bit element index and byte offset (buffer descriptors are per tile,
    lane offsets 32-bit, bases 64-bit).  A 288 GB part is meant to hold shards of this size and larger; the euclidean fused path also
    carries its transposed copy of V here (another 8 GiB)."""
    import torch
    m, n, K = 16384, 131072, 64
    V, W0, H0 = _rand(torch, (n, m), 1000), _rand(torch, (K, m), 1), _rand(torch, (n, K), 2)
    fused, cf = _run(torch, V, W0, H0, 2, divergence=div, path=2)
    Wf, Hf = fused.W.clone(), fused.H.clone()
    fused.close()
    del fused
    torch.cuda.empty_cache()
    gen, cg = _run(torch, V, W0, H0, 2, divergence=div, path=1)
    assert _rel(Wf, gen.W) < 1e-5 and _rel(Hf, gen.H) < 1e-5, (_rel(Wf, gen.W), _rel(Hf, gen.H))
    assert np.allclose(cf, cg, rtol=2e-6) and np.all(np.diff(cf) < 0)
    # the last columns / rows were really reached: an index that wrapped would leave them at their initial values
    assert float((Hf[-4:] - H0[-4:]).abs().max()) > 0 and float((Wf[:, -4:] - W0[:, -4:]).abs().max()) > 0


def test_cnmf_c4_full_size_gram_vs_materialised(gpu_lib):
    import torch
    m, n, K, T = 4096, 16384, 64, 8
    V, W0, H0 = _rand(torch, (n, m), 1000), _rand(torch, (T * K, m), 1), _rand(torch, (n, K), 2)
    gram, c1 = _run(torch, V, W0, H0, 4, divergence="euclidean", T=T, algorithm="cnmf", path=0)
    mat, c2 = _run(torch, V, W0, H0, 4, divergence="euclidean", T=T, algorithm="cnmf", path=1)
    assert _rel(gram.W, mat.W) < 1e-5 and _rel(gram.H, mat.H) < 1e-5 and np.allclose(c1, c2, rtol=2e-6)
    assert np.all(np.diff(c1) < 0)
    slab = (gram.W.double().reshape(T, K, m) ** 2).sum(dim=(0, 2)).sqrt()
    assert float((slab - T).abs().max()) < 1e-4 * T                 # slab Frobenius norm == T (cnmf.m:196-199)
    klg, c3 = _run(torch, V, W0, H0, 3, divergence="kl", T=T, algorithm="cnmf")
    assert np.all(np.diff(c3) < 0) and float(klg.H.min()) >= 0


def test_nmfsc_c5_full_size_properties(gpu_lib):
    m, n, K = 8192, 32768, 128
    rs = np.random.RandomState
    V = np.asfortranarray(rs(1000).rand(m, n))
    W0 = np.asfortranarray(rs(1).rand(m, K))
    H0 = np.asfortranarray(rs(2).rand(K, n))
    cfg = dict(W_init=W0, H_init=H0, H_sparsity=0.5, maxiter=4, nmfx_disable_stop=True)
    i1, i2 = {}, {}
    W, H, c = gpu_lib.nmfsc(V, K, dict(cfg, nmfx_path=2), info=i1)
    Wg, Hg, cg = gpu_lib.nmfsc(V, K, dict(cfg, nmfx_path=1), info=i2)
    assert i1["triesH"] == i2["triesH"]                              # identical line-search branches on both HIP paths
    assert np.linalg.norm(W - Wg) / np.linalg.norm(Wg) < 1e-5 and np.linalg.norm(H - Hg) / np.linalg.norm(Hg) < 1e-5
    assert np.allclose(c, cg, rtol=2e-6) and np.all(np.diff(c) <= 0)  # the line search never increases the objective (nmfsc.m:164)
    sp = (np.sqrt(n) - np.abs(H).sum(1) / np.sqrt((H ** 2).sum(1))) / (np.sqrt(n) - 1)
    assert np.allclose(sp, 0.5, atol=2e-5) and H.min() >= 0          # Hoyer sparseness of every row is exactly the target


@pytest.mark.parametrize("div,m,n,K", [("kl", 4097, 20001, 256), ("euclidean", 2049, 50001, 96)])
def test_nmf_large_ragged_paths_agree(gpu_lib, div, m, n, K):
    """Spectrogram-shaped (odd m, arbitrary n) at scale: the masked-edge fused kernels against the pipelined-GEMM path."""
    import torch
    V, W0, H0 = _rand(torch, (n, m), 1000), _rand(torch, (K, m), 1), _rand(torch, (n, K), 2)
    fused, cf = _run(torch, V, W0, H0, 3, divergence=div, path=2)
    assert fused.cost_lags
    gen, cg = _run(torch, V, W0, H0, 3, divergence=div, path=1)
    assert _rel(fused.W, gen.W) < 1e-5 and _rel(fused.H, gen.H) < 1e-5 and np.allclose(cf, cg, rtol=2e-6)
    assert np.all(np.diff(cf) < 0)
    nrm = (fused.W.double() ** 2).sum(dim=1).sqrt()
    assert float((nrm - 1).abs().max()) < 1e-5
