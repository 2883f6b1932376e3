"""CPU suite (-m "not gpu"): the oracle against the committed golden vectors, the two independent restatements against
each other, and the invariants the MATLAB source implies.  The reference ships no vectors of its own."""
import os

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from conftest import EPS, rel_fro, synth
from oracle import c_oracle as CO
from oracle import nmf_oracle as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
load = lambda name: np.load(os.path.join(G, name + ".npz"))


@pytest.mark.parametrize("div", ["euclidean", "kl"])
def test_c1_golden(div):
    g = load("nmf_c1_" + div)
    m, n, K = g["shape"]
    V, W0, H0 = synth(m, n, K)
    tr = []
    W, H, cost = O.nmf(V, K, dict(divergence=div, W_init=W0, H_init=H0, maxiter=50, tolerance=1e-12), trace=tr)
    assert rel_fro(cost, g["cost"]) < 1e-12 and len(cost) == 50
    assert rel_fro(W[::4], g["W50_sub"]) < 1e-11 and rel_fro(H[:, ::4], g["H50_sub"]) < 1e-11
    assert rel_fro(tr[0][0][::4], g["W1_sub"]) < 1e-13 and rel_fro(tr[1][1][:, ::4], g["H2_sub"]) < 1e-13
    assert abs(np.linalg.norm(W @ H) - g["WH50_fro"]) < 1e-10 * g["WH50_fro"]
    # independent plain-C restatement (fused-diagonal form) agrees with the literal NumPy one
    Wc, Hc, cc = CO.nmf(V, W0, H0, div=div, maxiter=50, tol=1e-12)
    assert rel_fro(Wc, W) < 1e-12 and rel_fro(Hc, H) < 1e-12 and rel_fro(cc, cost) < 1e-12
    # invariants: unit-L2 columns (nmf.m:169), non-negativity, monotone cost
    assert np.allclose(np.sqrt((W ** 2).sum(0)), 1.0, atol=1e-13) and W.min() >= 0 and H.min() >= 0
    assert np.all(np.diff(cost) <= 0)


@pytest.mark.parametrize("div", ["euclidean", "kl", "is"])
def test_nmf_small_golden(div):
    g = load("nmf_small_" + div)
    m, n, K = g["shape"]
    V, W0, H0 = synth(m, n, K)
    W, H, cost = O.nmf(V, K, dict(divergence=div, W_init=W0, H_init=H0, maxiter=30, tolerance=1e-12))
    assert rel_fro(W, g["W"]) < 1e-12 and rel_fro(H, g["H"]) < 1e-12 and rel_fro(cost, g["cost"]) < 1e-12
    Wc, Hc, cc = CO.nmf(V, W0, H0, div=div, maxiter=30, tol=1e-12)
    assert rel_fro(Wc, g["W"]) < 1e-12 and rel_fro(Hc, g["H"]) < 1e-12 and rel_fro(cc, g["cost"]) < 1e-12


def test_nmf_multi_source_and_stop_golden():
    g = load("nmf_small_multi")
    m, n, K = g["shape"]
    V, W0, H0 = synth(m, n, K)
    cfg = dict(divergence="kl", W_init=[W0[:, :3], W0[:, 3:]], H_init=[H0[:3], H0[3:]], W_sparsity=[0.1, 0.0], H_sparsity=[0.0, 0.2],
               W_fixed=[False, True], H_fixed=[False, False], maxiter=20, tolerance=1e-12)
    W, H, cost = O.nmf(V, [3, 5], cfg)
    assert isinstance(W, list) and isinstance(H, list)
    assert rel_fro(np.hstack(W), g["W"]) < 1e-12 and rel_fro(np.vstack(H), g["H"]) < 1e-12 and rel_fro(cost, g["cost"]) < 1e-12
    # concatenated form with per-column / per-row masks == cell form (what the C ABI runs)
    lamW = np.r_[np.full(3, 0.1), np.zeros(5)]
    lamH = np.r_[np.zeros(3), np.full(5, 0.2)]
    Wc, Hc, cc = CO.nmf(V, W0, H0, div="kl", lamW=lamW, lamH=lamH, fixW=np.r_[np.zeros(3), np.ones(5)], maxiter=20, tol=1e-12)
    assert rel_fro(Wc, g["W"]) < 1e-12 and rel_fro(Hc, g["H"]) < 1e-12 and rel_fro(cc, g["cost"]) < 1e-12
    s = load("nmf_small_stop")
    Vp = synth(m, n, K, planted=True)[0]
    cost = O.nmf(Vp, K, dict(W_init=W0, H_init=H0, maxiter=400, tolerance=2e-2))[2]
    assert len(cost) == int(s["iters"][0]) < 400 and rel_fro(cost, s["cost"]) < 1e-12
    assert cost[-2] - cost[-1] < 2e-2 <= cost[-3] - cost[-2]          # stopped exactly where nmf.m:221 says


@pytest.mark.parametrize("div", ["euclidean", "kl", "frobenius"])
def test_cnmf_golden(div):
    g = load("cnmf_small_" + div)
    m, n, K, T = g["shape"]
    V, W0, H0 = synth(m, n, K, T=T)
    W, H, cost = O.cnmf(V, K, T, dict(divergence=div, W_init=W0, H_init=H0, maxiter=20, tolerance=1e-12, W_sparsity=0.01, H_sparsity=0.02))
    assert rel_fro(W, g["W"]) < 1e-12 and rel_fro(H, g["H"]) < 1e-12
    assert np.allclose(cost, g["cost"], rtol=1e-12, atol=0)
    Wc, Hc, cc = CO.cnmf(V, W0, H0, div=div, lamW=0.01, lamH=0.02, maxiter=20, tol=1e-12)
    assert rel_fro(Wc, W) < 1e-12 and rel_fro(Hc, H) < 1e-12 and np.allclose(cc, cost, rtol=1e-11)
    assert np.allclose(np.sqrt((W ** 2).sum((0, 2))), T, rtol=1e-13)       # slab norm == T (cnmf.m:196-199)
    if div == "frobenius":                                                   # no cost case: only the L1 terms remain
        assert np.allclose(cost, 0.01 * np.abs(W).sum() + 0.02 * np.abs(H).sum(), rtol=0.2)


def test_cnmf_T1_is_matrix_branch():
    V, W0, H0 = synth(64, 96, 5)
    W, H, cost = O.cnmf(V, 5, 1, dict(W_init=W0, H_init=H0, maxiter=5))
    assert W.shape == (64, 5)
    assert rel_fro(O.reconstruct_from_decomposition(W, H), W @ H) < 1e-15


@pytest.mark.parametrize("tag", ["h", "wh", "mu"])
def test_nmfsc_golden(tag):
    g = load("nmfsc_small_" + tag)
    V, W0, H0 = synth(64, 256, 8)
    sW, sH = g["sparsity"]
    cfg = dict(W_init=W0, H_init=H0, maxiter=25, tolerance=1e-12)
    if sW:
        cfg["W_sparsity"] = sW
    if sH:
        cfg["H_sparsity"] = sH
    info = {}
    W, H, cost = O.nmfsc(3.0 * V, 8, cfg, info=info)
    assert rel_fro(W, g["W"]) < 1e-11 and rel_fro(H, g["H"]) < 1e-11 and rel_fro(cost, g["cost"]) < 1e-12
    assert info["triesH"] == list(g["triesH"]) and info["triesW"] == list(g["triesW"])
    Wc, Hc, cc, ic = CO.nmfsc(3.0 * V, W0, H0, sW=sW, sH=sH, maxiter=25, tol=1e-12)
    assert ic["triesH"] == info["triesH"] and ic["triesW"] == info["triesW"]
    assert rel_fro(Wc, W) < 1e-9 and rel_fro(Hc, H) < 1e-9 and rel_fro(cc, cost) < 1e-11
    assert np.all(np.diff(cost) <= 1e-12)                      # line search guarantees descent (nmfsc.m:164)
    if sH:                                                     # Hoyer sparseness of every row of H is exactly sH
        n = H.shape[1]
        sp = (np.sqrt(n) - np.abs(H).sum(1) / np.sqrt((H ** 2).sum(1))) / (np.sqrt(n) - 1)
        assert np.allclose(sp, sH, atol=1e-10)


@pytest.mark.parametrize("tag", ["mu", "h", "w"])
def test_cnmfsc_golden(tag):
    """cnmfsc.m restatement: literal NumPy vs golden vs the independent plain-C one."""
    g = load("cnmfsc_small_" + tag)
    V, W0, H0 = synth(48, 120, 5, T=3)
    sW, sH = g["sparsity"]
    cfg = dict(W_init=W0, H_init=H0, maxiter=10, tolerance=1e-12)
    if sW:
        cfg["W_sparsity"] = sW
    if sH:
        cfg["H_sparsity"] = sH
    info = {}
    W, H, cost = O.cnmfsc(2.0 * V, 5, 3, cfg, info=info)
    assert rel_fro(W, g["W"]) < 1e-11 and rel_fro(H, g["H"]) < 1e-11 and rel_fro(cost, g["cost"]) < 1e-12
    assert info["triesH"] == list(g["triesH"]) and info["triesW"] == list(g["triesW"])
    Wc, Hc, cc, ic = CO.cnmfsc(2.0 * V, W0, H0, sW=sW, sH=sH, maxiter=10, tol=1e-12)
    assert ic["triesH"] == info["triesH"] and ic["triesW"] == info["triesW"]
    assert rel_fro(Wc, W) < 1e-10 and rel_fro(Hc, H) < 1e-10 and rel_fro(cc, cost) < 1e-11
    if tag == "w":      # the reference's sparse-W line search compares against a shift-less product: it gives up by step-size underflow
        assert max(info["triesW"]) > 600 and len(cost) <= 3
    if tag == "mu":
        assert np.all(np.diff(cost[1:]) <= 1e-9 * cost[1])


def test_lnmf_golden():
    g = load("lnmf_small")
    V, W0, H0 = synth(96, 160, 8)
    W, H, cost = O.lnmf(V, 8, dict(W_init=W0, H_init=H0, maxiter=30, tolerance=1e-12))
    assert rel_fro(W, g["W"]) < 1e-12 and rel_fro(H, g["H"]) < 1e-12 and rel_fro(cost, g["cost"]) < 1e-12
    assert np.allclose(W.sum(0), 1.0, atol=1e-13) and np.all(np.diff(cost) <= 0)
    Wc, Hc, cc = CO.lnmf(V, W0, H0, maxiter=30, tol=1e-12)
    assert rel_fro(Wc, W) < 1e-12 and rel_fro(Hc, H) < 1e-12 and rel_fro(cc, cost) < 1e-12
    c = O.lnmf(V, 8, dict(W_init=W0, H_init=H0, maxiter=60, tolerance=1.0))[2]
    assert len(c) == 60 and 1 < np.count_nonzero(c) < 60 and np.all(c[np.count_nonzero(c):] == 0)   # not trimmed on break (lnmf.m:84-86)


def test_projfunc_golden():
    g = load("projfunc")
    for s, v0, it0 in zip(g["S"], g["V"], g["iters"]):
        v, it = O.projfunc(s, g["k1"][0], 1.0, True)
        vc, itc = CO.projfunc(s, g["k1"][0], 1.0, True)
        assert it == it0 == itc and rel_fro(v, v0) < 1e-13 and rel_fro(vc, v0) < 1e-13
        assert abs(v.sum() - g["k1"][0]) < 1e-12 and abs((v ** 2).sum() - 1) < 1e-12 and v.min() >= 0   # projfunc.m:3-7
    v, it = O.projfunc(g["s_signed"], 6.0, 1.0, False)
    assert it == g["it_signed"][0] and rel_fro(v, g["v_signed"]) < 1e-13
    assert np.all(np.sign(v[v != 0]) == np.sign(g["s_signed"][v != 0]))


def test_reconstruct_golden():
    g = load("reconstruct")
    assert rel_fro(O.reconstruct_from_decomposition(g["W"], g["H"]), g["V_hat"]) < 1e-14
    assert rel_fro(CO.reconstruct(g["W"], g["H"]), g["V_hat"]) < 1e-14
    assert rel_fro(O.reconstruct_from_decomposition(g["W"][:, :, 0], g["H"]), g["V_hat_2d"]) < 1e-14
    W, H = g["W"], g["H"]
    assert rel_fro(O.reconstruct_from_decomposition([W[:, :2], W[:, 2:]], [H[:2], H[2:]]), g["V_hat"]) < 1e-14   # cell inputs (RFD.m:23-28)
    # RFD == W_flat * H_stack
    m, K, T = W.shape
    n = H.shape[1]
    Hst = np.zeros((K * T, n))
    for t in range(T):
        Hst[t * K:(t + 1) * K, t:] = H[:, :n - t]
    assert rel_fro(W.transpose(0, 2, 1).reshape(m, T * K) @ Hst, g["V_hat"]) < 1e-14


@pytest.mark.parametrize("div", ["euclidean", "kl"])
def test_constrainednmf_golden(div):
    g = load("constrainednmf_" + div)
    V, W0, _ = synth(64, 120, 6)
    W, H, Z, A, cost = O.constrainednmf(V, g["labels"], 6, dict(divergence=div, W_init=W0, Z_init=g["Z0"], maxiter=20, tolerance=1e-12, Z_sparsity=0.05))
    assert rel_fro(W, g["W"]) < 1e-12 and rel_fro(H, g["H"]) < 1e-12 and rel_fro(Z, g["Z"]) < 1e-12 and rel_fro(cost, g["cost"]) < 1e-12
    assert np.array_equal(np.argmax(A, axis=0), g["A_nnz_cols"]) and np.all(A.sum(0) == 1) and np.allclose(H, Z @ A)
    # samples of one class share their encoding (the point of the constraint), unlabelled ones do not
    lab = g["labels"]
    for c in np.unique(lab[lab >= 0]):
        cols = np.nonzero(lab == c)[0]
        assert np.all(H[:, cols] == H[:, cols[:1]])
    assert np.all(np.diff(cost) <= 1e-9 * cost[0])
    # the independent C restatement works on the label-sorted problem: same W / Z / cost, H up to the sample permutation
    sidx = np.argsort(np.where(lab < 0, -1, np.searchsorted(np.unique(lab[lab >= 0]), lab) + 1), kind="stable")
    zcol = g["A_nnz_cols"][sidx]
    seg = np.concatenate([[0], np.cumsum(np.bincount(zcol, minlength=g["Z0"].shape[1]))])
    Wc, Hc, Zc, cc = CO.constrainednmf_sorted(V[:, sidx], W0, g["Z0"], seg, div=div, lamZ=0.05, maxiter=20, tol=1e-12)
    assert rel_fro(Wc, W) < 1e-11 and rel_fro(Zc, Z) < 1e-11 and rel_fro(cc, cost) < 1e-11 and rel_fro(Hc, H[:, sidx]) < 1e-11


def test_constrainednmf_reduces_to_nmf_when_nothing_is_labelled():
    """All labels -1: A = I, Z = H, and constrainednmf.m:183-258 is nmf.m:143-225 -- which pins this restatement to the nmf oracle."""
    V, W0, H0 = synth(48, 70, 5)
    for div in ("euclidean", "kl", "is"):
        cfg = dict(divergence=div, W_init=W0, maxiter=15, tolerance=1e-12, W_sparsity=0.02)
        W, H, Z, A, cost = O.constrainednmf(V, -np.ones(70, dtype=int), 5, dict(cfg, Z_init=H0, Z_sparsity=0.03))
        Wn, Hn, costn = O.nmf(V, 5, dict(cfg, H_init=H0, H_sparsity=0.03))
        assert np.array_equal(A, np.eye(70)) and np.array_equal(H, Z)
        assert rel_fro(W, Wn) < 1e-12 and rel_fro(H, Hn) < 1e-12 and rel_fro(cost, costn) < 1e-12
    with pytest.raises(ValueError, match="Length of the label vector"):
        O.constrainednmf(V, np.zeros(3), 5)
    with pytest.raises(ValueError, match="Matrix dimensions must agree"):      # constrainednmf.m:229 is ill-formed for alpha ~= 0
        O.constrainednmf(V, np.zeros(70, dtype=int), 5, dict(divergence="ab", alpha=0.5, beta=0.5, maxiter=1))


def test_sort_dictionary_golden():
    g = load("sort_dictionary")
    Ws, Hs = O.sort_dictionary(g["W"], g["H"])
    assert np.array_equal(Ws, g["W_sorted"]) and np.array_equal(Hs, g["H_sorted"])
    # hand-checkable: cumsum <= half-total, last index (SortDictionary.m:36-41)
    W = np.array([[1.0, 0.0, 5.0], [1.0, 0.0, 1.0], [1.0, 4.0, 1.0], [1.0, 0.0, 1.0]])      # cog = 2, 2, 1 (none <= 4 -> 1)
    Ws, Hs = O.sort_dictionary(W, np.array([[1.0], [2.0], [3.0]]))
    assert np.array_equal(Ws, W[:, [2, 0, 1]]) and np.array_equal(Hs.ravel(), [3.0, 1.0, 2.0])
    assert O.sort_dictionary(W)[1] is None
    assert np.array_equal(CO.sort_dictionary_order(W), [2, 0, 1])
    assert np.array_equal(g["W"][:, CO.sort_dictionary_order(g["W"])], g["W_sorted"])


def test_matlab_semantics():
    V, W0, H0 = synth(24, 30, 3)
    with pytest.raises(ValueError, match="No update equations"):
        O.nmf(V, 3, dict(divergence="nope", maxiter=1))
    with pytest.raises(ValueError, match="Requested 2 sources. Given 1 initial basis matrices."):
        O.nmf(V, [1, 2], dict(W_init=[W0]))
    with pytest.raises(ValueError, match="Negative values in data!"):
        O.nmfsc(-V, 3)
    c = O.nmf(V, 3, dict(maxiter=-3, tolerance=0, W_init=W0, H_init=H0))[2]     # <=0 -> 100 / 1e-3 (nmf.m:404-411)
    assert len(c) <= 100
    Vz = V.copy()
    Vz[0, 0] = 0.0                                                                # log(0): NaN cost never satisfies `<` (A.1)
    c = O.nmf(Vz, 3, dict(divergence="kl", W_init=W0, H_init=H0, maxiter=7))[2]
    assert len(c) == 7 and np.all(np.isnan(c))
    c = O.cnmf(V, 3, 2, dict(divergence="anything-goes", maxiter=4, rng=np.random.RandomState(1)))[2]   # cnmf.m:137-147 no otherwise
    assert len(c) == 4 and np.all(c == 0)


@settings(max_examples=25, deadline=None)
@given(st.integers(2, 40), st.floats(0.05, 0.95), st.integers(0, 10 ** 6))
def test_projfunc_postconditions_property(N, sparse, seed):
    s = np.abs(np.random.RandomState(seed).randn(N)) + 1e-3
    k1 = np.sqrt(N) - (np.sqrt(N) - 1) * sparse
    v, it = O.projfunc(s, k1, 1.0, True)
    vc, itc = CO.projfunc(s, k1, 1.0, True)
    assert abs(v.sum() - k1) < 1e-9 and abs((v ** 2).sum() - 1.0) < 1e-9 and v.min() >= 0 and it >= 1
    assert it == itc and np.allclose(v, vc, atol=1e-12)


@settings(max_examples=10, deadline=None)
@given(st.integers(3, 24), st.integers(3, 24), st.integers(1, 4), st.sampled_from(["euclidean", "kl"]), st.integers(0, 10 ** 6))
def test_nmf_invariants_property(m, n, K, div, seed):
    rs = np.random.RandomState(seed)
    V = np.fmax(rs.rand(m, n), EPS)
    W, H, cost = O.nmf(V, K, dict(divergence=div, maxiter=12, tolerance=1e-300, rng=rs))
    assert np.allclose(np.sqrt((W ** 2).sum(0)), 1.0, atol=1e-12) and W.min() >= 0 and H.min() >= 0
    assert np.all(np.diff(cost) <= 1e-9 * abs(cost[0]))
