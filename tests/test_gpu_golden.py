"""-m gpu: the HIP path against the COMMITTED fixtures tests/golden/*.npz -- no oracle import anywhere in this file.

The fixtures were written by tests/golden/make_golden.py from the float64 restatement (the reference is MATLAB: it ships no vectors and cannot run in
the image -- parity stays "unpinned by the reference"); inputs are regenerated from the seeds of conftest.synth, exactly as that
script did.  Contract: W, H (and W*H) within 1e-5 relative Frobenius error, the cost vector within 1e-6, equal length, identical line-search try counts."""
import os

import numpy as np
import pytest

from conftest import record_err, rel_fro, synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def test_no_oracle_in_this_module():
    src = open(os.path.abspath(__file__)).read()
    assert ("from " + "oracle") not in src and ("import " + "oracle") not in src


def _check(W, H, cost, g, tolW=1e-5, tolH=1e-5, tolc=1e-6):
    eW, eH, ec = rel_fro(W, g["W"]), rel_fro(H, g["H"]), rel_fro(cost, g["cost"])
    record_err(W=eW, H=eH, cost=ec)
    assert len(cost) == len(g["cost"]), (len(cost), len(g["cost"]))
    assert eW <= tolW and eH <= tolH and ec <= tolc, (eW, eH, ec)


@pytest.mark.parametrize("div", ["euclidean", "kl"])
def test_golden_nmf_c1(gpu_lib, div):
    """BASELINE.json configs[0]: nmf.m, V = 512 x 1024, K = 16, 50 iterations -- the states after iterations 1, 2 and 50 and the whole cost vector"""
    g = gold("nmf_c1_" + div)
    m, n, K = (int(x) for x in g["shape"])
    V, W0, H0 = synth(m, n, K)
    for iters, tag in ((1, "1"), (2, "2"), (50, "50")):
        W, H, cost = gpu_lib.nmf(V, K, dict(divergence=div, W_init=W0, H_init=H0, maxiter=iters, tolerance=1e-12))
        eW, eH = rel_fro(W[::4, :], g["W%s_sub" % tag]), rel_fro(H[:, ::4], g["H%s_sub" % tag])
        ec = rel_fro(cost, g["cost"][:iters])
        record_err(W=eW, H=eH, cost=ec)
        assert eW <= 1e-5 and eH <= 1e-5 and ec <= 1e-6, (iters, eW, eH, ec)
    assert abs(np.linalg.norm(W) / float(g["W50_fro"]) - 1) < 1e-6 and abs(np.linalg.norm(H) / float(g["H50_fro"]) - 1) < 1e-5
    assert abs(np.linalg.norm(W @ H) / float(g["WH50_fro"]) - 1) < 1e-6


@pytest.mark.parametrize("div", ["euclidean", "kl", "is"])
@pytest.mark.parametrize("path", [0, 1])
def test_golden_nmf_small(gpu_lib, div, path):
    g = gold("nmf_small_" + div)
    m, n, K = (int(x) for x in g["shape"])
    V, W0, H0 = synth(m, n, K)
    W, H, cost = gpu_lib.nmf(V, K, dict(divergence=div, W_init=W0, H_init=H0, maxiter=30, tolerance=1e-12, nmfx_path=path))
    _check(W, H, cost, g)


def test_golden_nmf_multi_source_and_stop(gpu_lib):
    g = gold("nmf_small_multi")
    m, n, K = (int(x) for x in g["shape"])
    V, W0, H0 = synth(m, n, K)
    cfg = dict(divergence="kl", W_init=[W0[:, :3], W0[:, 3:]], H_init=[H0[:3], H0[3:]], W_sparsity=[0.1, 0.0], H_sparsity=[0.0, 0.2],
               W_fixed=[False, True], H_fixed=[False, False], maxiter=20, tolerance=1e-12)
    W, H, cost = gpu_lib.nmf(V, [3, 5], cfg)
    _check(np.hstack(W), np.vstack(H), cost, g)
    gs = gold("nmf_small_stop")                       # the stop rule of nmf.m:221-224 fires at the iteration the fixture recorded
    Vp = synth(m, n, K, planted=True)[0]
    Wp, Hp, costp = gpu_lib.nmf(Vp, K, dict(W_init=W0, H_init=H0, maxiter=400, tolerance=2e-2))
    assert len(costp) == int(gs["iters"][0]) and rel_fro(costp, gs["cost"]) <= 1e-6


@pytest.mark.parametrize("div", ["euclidean", "kl", "frobenius"])
def test_golden_cnmf_small(gpu_lib, div):
    g = gold("cnmf_small_" + div)
    m, n, K, T = (int(x) for x in g["shape"])
    V, W0, H0 = synth(m, n, K, T=T)
    W, H, cost = gpu_lib.cnmf(V, K, T, dict(divergence=div, W_init=W0, H_init=H0, maxiter=20, tolerance=1e-12, W_sparsity=0.01, H_sparsity=0.02))
    _check(W, H, cost, g)                             # ('frobenius': cnmf.m:239-248 has no such case -- the data term of the cost stays zero, the sparsity terms remain)


@pytest.mark.parametrize("tag", ["h", "wh", "mu"])
@pytest.mark.parametrize("path", [0, 2])
def test_golden_nmfsc_small(gpu_lib, tag, path):
    """nmfsc.m with the Hoyer projection on H, on W and H, and its multiplicative branch; default path (float64 VALU kernels at this size) and the fused MFMA kernels"""
    g = gold("nmfsc_small_" + tag)
    V, W0, H0 = synth(64, 256, 8)
    sW, sH = (float(x) for x in g["sparsity"])
    cfg = dict(W_init=W0, H_init=H0, maxiter=25, tolerance=1e-12, nmfx_path=path)
    if sW:
        cfg["W_sparsity"] = sW
    if sH:
        cfg["H_sparsity"] = sH
    info = {}
    W, H, cost = gpu_lib.nmfsc(3.0 * V, 8, cfg, info=info)
    assert info["triesH"] == [int(t) for t in g["triesH"]] and info["triesW"] == [int(t) for t in g["triesW"]], (info, g["triesH"], g["triesW"])
    _check(W, H, cost, g)
    assert abs(info["stepsizeH"] / float(g["steps"][0]) - 1) < 1e-12 and abs(info["stepsizeW"] / float(g["steps"][1]) - 1) < 1e-12


@pytest.mark.parametrize("tag", ["mu", "h", "w"])
def test_golden_cnmfsc_small(gpu_lib, tag):
    g = gold("cnmfsc_small_" + tag)
    V, W0, H0 = synth(48, 120, 5, T=3)
    sW, sH = (float(x) for x in g["sparsity"])
    cfg = dict(W_init=W0, H_init=H0, maxiter=10, tolerance=1e-12)
    if sW:
        cfg["W_sparsity"] = sW
    if sH:
        cfg["H_sparsity"] = sH
    info = {}
    W, H, cost = gpu_lib.cnmfsc(2.0 * V, 5, 3, cfg, info=info)
    assert info["triesH"] == [int(t) for t in g["triesH"]] and info["triesW"] == [int(t) for t in g["triesW"]]
    _check(W, H, cost, g)


def test_golden_lnmf(gpu_lib):
    g = gold("lnmf_small")
    V, W0, H0 = synth(96, 160, 8)
    W, H, cost = gpu_lib.lnmf(V, 8, dict(W_init=W0, H_init=H0, maxiter=30, tolerance=1e-12))
    _check(W, H, cost, g)


def test_golden_projfunc(gpu_lib):
    g = gold("projfunc")
    k1 = float(g["k1"][0])
    for s, v, it in zip(g["S"], g["V"], g["iters"]):
        got, used = gpu_lib.projfunc(s, k1, 1.0, True)
        assert used == int(it) and rel_fro(got, v) <= 1e-12 and np.array_equal(got == 0, v == 0)      # float64 in, float64 end to end: identical zero sets
        got32, used32 = gpu_lib.projfunc(s.astype(np.float32), k1, 1.0, True)
        assert used32 == int(it) and rel_fro(got32, v) <= 1e-6
    got, used = gpu_lib.projfunc(g["s_signed"], 6.0, 1.0, False)
    assert used == int(g["it_signed"][0]) and rel_fro(got, g["v_signed"]) <= 1e-12


def test_golden_reconstruct(gpu_lib):
    g = gold("reconstruct")
    assert rel_fro(gpu_lib.ReconstructFromDecomposition(g["W"], g["H"]), g["V_hat"]) <= 1e-6
    assert rel_fro(gpu_lib.ReconstructFromDecomposition(g["W"][:, :, 0], g["H"]), g["V_hat_2d"]) <= 1e-6


@pytest.mark.parametrize("div", ["euclidean", "kl"])
def test_golden_constrainednmf(gpu_lib, div):
    g = gold("constrainednmf_" + div)
    V, W0, _ = synth(64, 120, 6)
    W, H, Z, A, cost = gpu_lib.constrainednmf(V, g["labels"], 6, dict(divergence=div, W_init=W0, Z_init=g["Z0"], maxiter=20, tolerance=1e-12, Z_sparsity=0.05))
    _check(W, H, cost, g)
    assert rel_fro(Z, g["Z"]) <= 1e-5 and np.array_equal(np.argmax(A, axis=0), g["A_nnz_cols"])


def test_golden_sort_dictionary(gpu_lib):
    g = gold("sort_dictionary")
    Ws, Hs = gpu_lib.SortDictionary(g["W"], g["H"])
    assert np.array_equal(Ws, g["W_sorted"]) and np.array_equal(Hs, g["H_sorted"])                    # bit-exact: a permutation
