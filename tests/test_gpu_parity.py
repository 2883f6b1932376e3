"""Parity proper (-m gpu): the HIP path behind the reference call surface vs the float64 oracle on the same
seeded inputs.  Tolerance is the north-star contract: <= 1e-5 relative Frobenius on W, H and W*H; cost rel <= 1e-6
(1e-5 where noted); identical iteration counts / line-search try counts."""
import numpy as np
import pytest

from conftest import record_err, rel_fro, synth

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _check(got, ref, tol=TOL, cost_tol=1e-6, cost_atol=0.0):
    (W, H, c), (W0, H0, c0) = got, ref
    cat = lambda x, ax: np.concatenate(x, axis=ax) if isinstance(x, list) else x
    assert type(W) is type(W0) and type(H) is type(H0)
    W, W0, H, H0 = cat(W, 1), cat(W0, 1), cat(H, 0), cat(H0, 0)
    assert W.shape == W0.shape and H.shape == H0.shape
    assert len(c) == len(c0), (len(c), len(c0))
    record_err(W=rel_fro(W, W0), H=rel_fro(H, H0))
    if np.all(np.isfinite(c0)) and np.linalg.norm(c0) > 0 and cost_atol == 0:
        record_err(cost=rel_fro(c, c0))
    assert rel_fro(W, W0) <= tol, rel_fro(W, W0)
    assert rel_fro(H, H0) <= tol, rel_fro(H, H0)
    if not np.all(np.isfinite(c0)):
        assert np.array_equal(np.isnan(c), np.isnan(c0)) and np.array_equal(c[~np.isnan(c0)], c0[~np.isnan(c0)])   # same +-Inf / NaN pattern
    elif cost_atol > 0:      # exact-fit cases: the cost is rounding noise around zero, compare on the scale of the data
        assert np.allclose(c, c0, rtol=cost_tol, atol=cost_atol), (c, c0)
    elif np.linalg.norm(c0) > 0:
        assert rel_fro(c, c0) <= cost_tol, rel_fro(c, c0)
    else:
        assert np.all(c == 0)


def _check_stop(c, c0, tol, le=False):
    """Stop rule of nmf.m:221-224 (lnmf.m:84 with <=): the HIP path must stop at the SAME iteration as the float64 oracle whenever
    the decisive comparisons cost(i-1) - cost(i) < tol are farther from the threshold than the cost noise of the fp32 path (twice
    the largest |cost difference| observed on the common prefix); only inside that band a +-1 shift is accepted."""
    k = min(len(c), len(c0))
    noise = 2.0 * float(np.max(np.abs(np.asarray(c[:k]) - np.asarray(c0[:k]))))
    d = -np.diff(np.asarray(c0, dtype=np.float64))
    margin = float(np.min(np.abs(d - tol))) if len(d) else np.inf
    record_err(stop_noise_over_margin=noise / margin if margin > 0 else np.inf)
    if margin > noise:
        assert len(c) == len(c0), (len(c), len(c0), margin, noise)
    else:
        assert abs(len(c) - len(c0)) <= 1, (len(c), len(c0), margin, noise)
    assert rel_fro(c[:k], c0[:k]) <= 1e-6


@pytest.mark.parametrize("div", ["euclidean", "kl_divergence", "is"])
@pytest.mark.parametrize("m,n,K,iters", [(512, 1024, 16, 50), (192, 200, 7, 30), (128, 256, 32, 20)])
def test_nmf_matches_oracle(gpu_lib, div, m, n, K, iters):
    from oracle import nmf_oracle as O
    V, W0, H0 = synth(m, n, K)
    cfg = dict(divergence=div, W_init=W0, H_init=H0, maxiter=iters, tolerance=1e-12)
    _check(gpu_lib.nmf(V, K, cfg), O.nmf(V, K, cfg), cost_tol=1e-6 if div != "is" else 1e-5)


def test_nmf_stop_rule_and_defaults(gpu_lib):
    from oracle import nmf_oracle as O
    V, W0, H0 = synth(96, 160, 8, planted=True)
    cfg = dict(W_init=W0, H_init=H0, maxiter=400, tolerance=2e-2)
    got, ref = gpu_lib.nmf(V, 8, cfg), O.nmf(V, 8, cfg)
    assert len(ref[2]) < 400            # the stop rule fired in the oracle ...
    _check_stop(got[2], ref[2], 2e-2)    # ... and at the same place
    W, H, cost = gpu_lib.nmf(V, 8, dict(maxiter=0, tolerance=-5, seed=3))   # <=0 -> defaults 100 / 1e-3 (nmf.m:404-411)
    assert len(cost) <= 100 and W.shape == (96, 8) and H.shape == (8, 160)
    assert np.allclose(np.sqrt((W ** 2).sum(0)), 1.0, atol=1e-5)            # unit-L2 columns (nmf.m:169)
    assert np.all(np.diff(cost) <= 1e-6 * cost[0])                          # monotone non-increasing


@pytest.mark.parametrize("div", ["euclidean", "kl"])
def test_nmf_multi_source_sparsity_fixed(gpu_lib, div):
    from oracle import nmf_oracle as O
    V, W0, H0 = synth(160, 224, 12)
    Ks = [3, 4, 5]
    cfg = dict(divergence=div, W_init=[W0[:, :3], W0[:, 3:7], W0[:, 7:]], H_init=[H0[:3], H0[3:7], H0[7:]], W_sparsity=[0.1, 0.0, 0.05],
               H_sparsity=[0.0, 0.2, 0.0], W_fixed=[False, True, False], H_fixed=[False, False, True], maxiter=25, tolerance=1e-12)
    got, ref = gpu_lib.nmf(V, Ks, cfg), O.nmf(V, Ks, cfg)
    _check(got, ref)
    assert rel_fro(got[0][1], W0[:, 3:7] / np.sqrt((W0[:, 3:7] ** 2).sum(0))) < 1e-6   # fixed source: only the init normalisation
    assert rel_fro(got[1][2], H0[7:]) < 1e-7


def test_nmf_errors(gpu_lib):
    V, W0, H0 = synth(32, 40, 4)
    with pytest.raises(ValueError, match="No update equations defined"):
        gpu_lib.nmf(V, 4, dict(divergence="bogus"))
    with pytest.raises(ValueError, match="Requested 2 sources. Given 1 initial encoding matrices."):
        gpu_lib.nmf(V, [2, 2], dict(H_init=[H0]))
    with pytest.raises(ValueError, match="Requested 2 sources. Given 3 sparsity levels."):
        gpu_lib.nmf(V, [2, 2], dict(W_sparsity=[0.1, 0.2, 0.3]))
    with pytest.raises(ValueError, match="alpha = 0 and beta = 0"):
        gpu_lib.nmf(V, 4, dict(divergence="ab", alpha=0, beta=0))


@pytest.mark.parametrize("div", ["euclidean", "kl", "is", "frobenius"])
@pytest.mark.parametrize("m,n,K,T,iters", [(256, 1024, 16, 8, 30), (96, 130, 5, 3, 20), (128, 256, 8, 1, 15)])
def test_cnmf_matches_oracle(gpu_lib, div, m, n, K, T, iters):
    from oracle import nmf_oracle as O
    V, W0, H0 = synth(m, n, K, T=T)
    cfg = dict(divergence=div, W_init=W0 if T > 1 else W0[:, :, 0], H_init=H0, maxiter=iters, tolerance=1e-12, W_sparsity=0.01, H_sparsity=0.02)
    got, ref = gpu_lib.cnmf(V, K, T, cfg), O.cnmf(V, K, T, cfg)
    _check(got, ref, cost_tol=1e-5 if div == "is" else 1e-6)
    W = got[0].reshape(m, K, -1)
    assert np.allclose(np.sqrt((W ** 2).sum((0, 2))), T, rtol=1e-5)   # slab Frobenius norm == T (cnmf.m:196-199)


def test_cnmf_multi_source(gpu_lib):
    from oracle import nmf_oracle as O
    V, W0, H0 = synth(128, 192, 8, T=4)
    cfg = dict(divergence="kl", W_init=[W0[:, :3], W0[:, 3:]], H_init=[H0[:3], H0[3:]], W_fixed=[True, False], H_sparsity=[0.1, 0.0], maxiter=15, tolerance=1e-12)
    _check(gpu_lib.cnmf(V, [3, 5], 4, cfg), O.cnmf(V, [3, 5], 4, cfg))


@pytest.mark.parametrize("sW,sH", [(0.0, 0.5), (0.4, 0.6), (0.0, 0.0), (0.3, 0.0)])
def test_nmfsc_matches_oracle(gpu_lib, sW, sH):
    from oracle import nmf_oracle as O
    V, W0, H0 = synth(256, 2048, 16)
    V = 3.0 * V     # exercises the V / max(V) rescale (nmfsc.m:62)
    cfg = dict(W_init=W0, H_init=H0, maxiter=30, tolerance=1e-12)
    if sW:
        cfg["W_sparsity"] = sW
    if sH:
        cfg["H_sparsity"] = sH
    i0, i1 = {}, {}
    got = gpu_lib.nmfsc(V, 16, cfg, info=i1)
    ref = O.nmfsc(V, 16, cfg, info=i0)
    assert i1["triesH"] == i0["triesH"] and i1["triesW"] == i0["triesW"]      # identical line-search branches
    _check(got, ref)
    assert abs(i1["stepsizeH"] - i0["stepsizeH"]) <= 1e-12 * i0["stepsizeH"]


def test_nmfsc_negative_data(gpu_lib):
    with pytest.raises(ValueError, match="Negative values in data!"):
        gpu_lib.nmfsc(-np.ones((4, 4)), 2)


def test_reconstruct(gpu_lib):
    from oracle import nmf_oracle as O
    V, W0, H0 = synth(200, 300, 6, T=5)
    assert rel_fro(gpu_lib.ReconstructFromDecomposition(W0, H0), O.reconstruct_from_decomposition(W0, H0)) < 1e-6
    assert rel_fro(gpu_lib.ReconstructFromDecomposition(W0[:, :, 0], H0), W0[:, :, 0] @ H0) < 1e-6
    assert rel_fro(gpu_lib.ReconstructFromDecomposition([W0[:, :2], W0[:, 2:]], [H0[:2], H0[2:]]), O.reconstruct_from_decomposition(W0, H0)) < 1e-6


def test_host_staging_crosses_chunk_boundaries(gpu_lib):
    """the pinned double-buffer staging of the blocking calls (csrc/host_io.hip: 16 Mi elements per chunk, conversion on host threads):
    arrays of 2.1 chunks in and out, both host precisions; K = 1 so V_hat = w*h' is exact in fp32 up to one rounding"""
    rs = np.random.RandomState(5)
    m, n = 4100, 8300                                            # 34.0 M elements
    w, h = rs.rand(m, 1), rs.rand(1, n)
    Vh = gpu_lib.ReconstructFromDecomposition(w, h)
    ref = w.astype(np.float32).astype(np.float64) @ h.astype(np.float32).astype(np.float64)
    assert Vh.shape == (m, n) and np.abs(Vh - ref).max() <= 1.2e-7 * ref.max()
    # nmf on a float32, column-major V is taken as is (no float64 detour) and equals the run on the same values widened to float64
    V32 = np.asfortranarray(np.fmax(rs.rand(4096, 4400), 2.0 ** -52).astype(np.float32))   # 18.0 M elements: two chunks
    W0, H0 = np.fmax(rs.rand(4096, 32), 2.0 ** -52), np.fmax(rs.rand(32, 4400), 2.0 ** -52)
    cfg = dict(divergence="kl", W_init=W0.astype(np.float32), H_init=H0.astype(np.float32), maxiter=2)
    Wa, Ha, ca = gpu_lib.nmf(V32, 32, cfg)
    Wb, Hb, cb = gpu_lib.nmf(V32.astype(np.float64), 32, dict(cfg, W_init=W0.astype(np.float32).astype(np.float64), H_init=H0.astype(np.float32).astype(np.float64)))
    assert Wa.dtype == np.float32 and Wb.dtype == np.float64
    assert np.array_equal(Wa.astype(np.float64), Wb) and np.array_equal(Ha.astype(np.float64), Hb) and np.array_equal(ca, cb)
    from nmf_toolbox_amd import _lib
    tm = _lib.last_call_timing()
    assert tm["host_bytes_in"] == 8.0 * (V32.size + W0.size + H0.size) and tm["ingest_s"] > 0 and tm["iterate_s"] > 0 and tm["egress_s"] > 0


@pytest.mark.parametrize("alg,div,path,K,T", [("nmf", "euclidean", 2, 64, 1), ("nmf", "euclidean", 2, 256, 1), ("nmf", "kl", 2, 128, 1), ("nmf", "is", 2, 64, 1),
                                              ("nmf", "euclidean", 1, 64, 1), ("nmf", "kl", 1, 40, 1), ("nmf", "euclidean", 0, 320, 1),
                                              ("cnmf", "euclidean", 0, 64, 4), ("cnmf", "kl", 0, 64, 4), ("cnmf", "is", 0, 16, 3), ("nmfsc", "euclidean", 2, 64, 1)])
def test_run_to_run_determinism(gpu_lib, alg, div, path, K, T):
    """SURVEY section 5's substitute for a race detector: the same inputs twice give bit-identical W, H and cost on every kernel path -- fused
    (Gram-form cost, dual-map, KL), Gram form on the GEMM (K > 256), materialised V_hat, the cnmf shift-sum passes, nmfsc's line searches.
    (Every reduction has a fixed order: split slabs are summed deterministically, nothing uses floating-point atomics.)"""
    m, n = 384, 1280
    V, W0, H0 = synth(m, n, K, T=(T if alg == "cnmf" else None))
    runs = []
    for _ in range(2):
        if alg == "nmf":
            out = gpu_lib.nmf(V, K, dict(divergence=div, W_init=W0, H_init=H0, maxiter=6, tolerance=1e-12, nmfx_path=path))
        elif alg == "cnmf":
            out = gpu_lib.cnmf(V, K, T, dict(divergence=div, W_init=W0, H_init=H0, maxiter=5, tolerance=1e-12, nmfx_path=path))
        else:
            out = gpu_lib.nmfsc(V, K, dict(W_init=W0, H_init=H0, H_sparsity=0.5, W_sparsity=0.3, maxiter=4, tolerance=1e-12, nmfx_path=path))
        runs.append(out)
    for a, b in zip(runs[0], runs[1]):
        assert np.array_equal(np.asarray(a), np.asarray(b))


@pytest.mark.parametrize("noise", [1.0, 0.3, 0.1, 0.03, 0.01, 0.001, 0.0])
def test_gram_form_cost_over_residual_levels(gpu_lib, noise):
    """Euclidean fused path: the cost comes in Gram form, 0.5*||V||^2 - <W, V*H'> + 0.5*<W, W*(H*H')>, out of the W update's column sums while the
    residual is large enough for fp32 to resolve that difference, and from the explicit residual pass once it is not (device-side switch at
    cost < 5 % of 0.5*||V||^2).  Planted data from pure noise (ratio 0.5) to an exact factorisation: every reported cost holds the 1e-6
    contract, whichever mode produced it, and the stop rule fires where the oracle's does."""
    from oracle import nmf_oracle as O
    rs = np.random.RandomState(11)
    m, n, K = 384, 1536, 64
    Wt, Ht = rs.rand(m, K), rs.rand(K, n)
    V = np.fmax(Wt @ Ht / K + noise * rs.rand(m, n), 2.0 ** -52)
    W0 = np.fmax(Wt + 0.2 * rs.rand(m, K), 2.0 ** -52)
    H0 = np.fmax(Ht / K * 4 + 0.2 * rs.rand(K, n), 2.0 ** -52)
    cfg = dict(divergence="euclidean", W_init=W0, H_init=H0, maxiter=14, tolerance=1e-12)
    ref = O.nmf(V, K, cfg)
    out = gpu_lib.nmf(V, K, dict(cfg, nmfx_path=2))
    _check(out, ref)
    ratio = ref[2] / (0.5 * np.sum(V ** 2))
    record_err(cost=np.max(np.abs(out[2] - ref[2]) / np.abs(ref[2])))
    assert np.max(np.abs(out[2] - ref[2]) / np.abs(ref[2])) < 1e-6, (noise, ratio, out[2], ref[2])
    # the same with the reference's default tolerance: same number of iterations, same state
    cfg2 = dict(cfg, maxiter=60, tolerance=1e-3 * max(1.0, ref[2][-1]))
    ref2 = O.nmf(V, K, cfg2)
    out2 = gpu_lib.nmf(V, K, dict(cfg2, nmfx_path=2))
    _check_stop(out2[2], ref2[2], cfg2["tolerance"])
    if len(out2[2]) == len(ref2[2]):
        _check(out2, ref2)


# ---- fused kernels (S = W*H never stored): eligible shapes, both split and un-split epilogues -----------------------
@pytest.mark.parametrize("div", ["kl", "euclidean"])
@pytest.mark.parametrize("m,n,K,iters", [(256, 1024, 64, 25), (384, 640, 128, 15), (128, 32768, 64, 4), (256, 512, 256, 10),
                                         (256, 512, 32, 12), (256, 640, 96, 12), (128, 512, 160, 10), (256, 384, 192, 10), (128, 256, 224, 10),
                                         (256, 512, 40, 12), (128, 384, 100, 10), (128, 256, 7, 15), (256, 256, 250, 8)])   # K padded to 64 / 128 / 32 / 256
def test_nmf_fused_matches_oracle_and_generic(gpu_lib, div, m, n, K, iters):
    from oracle import nmf_oracle as O
    V, W0, H0 = synth(m, n, K)
    cfg = dict(divergence=div, W_init=W0, H_init=H0, maxiter=iters, tolerance=1e-12)
    ref = O.nmf(V, K, cfg)
    fused = gpu_lib.nmf(V, K, dict(cfg, nmfx_path=2))
    generic = gpu_lib.nmf(V, K, dict(cfg, nmfx_path=1))
    _check(fused, ref)
    _check(generic, ref)
    _check(fused, generic)


def test_nmf_fused_multi_source_and_stop(gpu_lib):
    from oracle import nmf_oracle as O
    V, W0, H0 = synth(256, 512, 64, planted=True)
    Ks = [24, 40]
    cfg = dict(divergence="kl", W_init=[W0[:, :24], W0[:, 24:]], H_init=[H0[:24], H0[24:]], W_sparsity=[0.05, 0.0], H_sparsity=[0.0, 0.1],
               W_fixed=[False, True], H_fixed=[False, False], maxiter=40, tolerance=1e-12, nmfx_path=2)
    _check(gpu_lib.nmf(V, Ks, cfg), O.nmf(V, Ks, cfg))
    cfg = dict(divergence="euclidean", W_init=W0, H_init=H0, maxiter=300, tolerance=5e-2, nmfx_path=2)
    got, ref = gpu_lib.nmf(V, 64, cfg), O.nmf(V, 64, cfg)
    assert len(ref[2]) < 300
    _check_stop(got[2], ref[2], 5e-2)


# ---- alpha-beta divergence (nmf.m:157-164,188-195,213-214; cnmf.m:179-194,227-231), incl. the dual form alpha == 0 ----
@pytest.mark.parametrize("alpha,beta", [(0.5, 1.5), (2.0, -0.5), (1.0, 1.0), (0.0, 1.0), (0.0, 2.0)])
def test_nmf_ab_divergence(gpu_lib, alpha, beta):
    from oracle import nmf_oracle as O
    V, W0, H0 = synth(160, 224, 12)
    # the reference's dual equations (alpha == 0) diverge double-exponentially on this data even in float64 -- max(H) = 1.4e6, 5e18, 6e43,
    # 1e94, 3e194 after iterations 1..5, NaN from iteration 6 (oracle run, beta = 1; beta = 2: 6e2, 2e10, 4e26, 3e59) -- float32 holds the
    # iterates for 2 iterations and not one more (beta = 1: max(H) = 6e43 > FLT_MAX at the third; beta = 2: H*H' overflows in the third and
    # the +-Inf cost of the reference comes out NaN): that is what is compared
    iters = 20 if alpha != 0 else 2
    cfg = dict(divergence="ab_divergence", alpha=alpha, beta=beta, W_init=W0, H_init=H0, maxiter=iters, tolerance=1e-12, W_sparsity=0.01)
    _check(gpu_lib.nmf(V, 12, cfg), O.nmf(V, 12, cfg))


@pytest.mark.parametrize("beta", [1.0, 2.0, 0.5])
@pytest.mark.parametrize("m,n,K", [(256, 768, 64), (384, 1024, 256), (321, 515, 100), (128, 640, 12)])
def test_nmf_ab_dual_form_on_the_fused_kernels(gpu_lib, beta, m, n, K):
    """alpha == 0 selects the reference's DUAL update equations (nmf.m:124-128,159-160,190-191): numerators (V.^(-1) .* V_hat.^beta) * H' through S = W*H (functor 17),
    denominators V.^(beta-1) * H' without it, outer exponent 1/beta, and a cost that divides by alpha*beta = 0.  The equations diverge double-exponentially even in
    float64, so -- as in test_nmf_ab_divergence -- two iterations are what float32 can be held to; the fused path (default), the materialised path and three column
    shards against the oracle, +-Inf cost pattern included."""
    from oracle import nmf_oracle as O
    V, W0, H0 = synth(m, n, K)
    if K % 32 == 0 and m % 128 == 0:
        import torch
        from nmf_toolbox_amd.engine import Engine, colmajor_to_torch
        e = Engine(colmajor_to_torch(V, "cuda:0"), colmajor_to_torch(W0, "cuda:0"), colmajor_to_torch(H0, "cuda:0"), divergence="ab", alpha=0.0, beta=beta, use_dist=False)
        assert e.path_kind == 1           # the fused kernels, not the materialised path
        e.close()
    cfg = dict(divergence="ab", alpha=0.0, beta=beta, W_init=W0, H_init=H0, maxiter=2, tolerance=1e-12, W_sparsity=0.01)
    with np.errstate(all="ignore"):
        ref = O.nmf(V, K, cfg)
    _check(gpu_lib.nmf(V, K, cfg), ref)
    _check(gpu_lib.nmf(V, K, dict(cfg, nmfx_path=1)), ref)
    _check(gpu_lib.nmf(V, K, dict(cfg, nmfx_gpus=[0, 0, 0])), ref)


@pytest.mark.parametrize("alpha,beta", [(0.5, 1.5), (0.0, 1.0)])
def test_cnmf_ab_divergence(gpu_lib, alpha, beta):
    from oracle import nmf_oracle as O
    V, W0, H0 = synth(96, 130, 5, T=3)
    cfg = dict(divergence="ab", alpha=alpha, beta=beta, W_init=W0, H_init=H0, maxiter=15 if alpha != 0 else 2, tolerance=1e-12)
    _check(gpu_lib.cnmf(V, 5, 3, cfg), O.cnmf(V, 5, 3, cfg))


@pytest.mark.parametrize("div,ab,m,n,K,T", [("is", None, 192, 512, 64, 8), ("is", None, 128, 400, 32, 4), ("ab", (0.5, 1.5), 128, 400, 32, 4), ("ab", (2.0, -0.5), 192, 512, 64, 8),
                                            ("ab", (1.0, 0.5), 132, 777, 64, 2), ("is", None, 516, 1031, 32, 8), ("ab", (1.5, -1.5), 256, 300, 128, 2), ("is", None, 64, 2050, 32, 16),
                                            ("is", None, 260, 640, 64, 4), ("ab", (0.5, 0.5), 320, 515, 128, 4)])
def test_cnmf_is_and_alpha_beta_on_the_fused_passes(gpu_lib, div, ab, m, n, K, T):
    """IS / alpha-beta cnmf (cnmf.m:179-194,227-231) without V_hat, round 5 (engine.fusedT_dual): an S pass on the cnmf kernel stores BOTH element maps' values
    (functors 11 / 13 in the cost-only form: A = V./S.^2 | V.^a.*S.^(b-1) with the cost terms, B = 1./S | S.^(a+b-1)), two numerator passes contract them with
    H_stack', the H step two W_flat'*(.) GEMMs + shift-sums.  All eight (K, T) pairs it is instantiated for, ragged m and n, sparsity terms on: against the oracle at the
    contract, against the materialised path (nmfx_path = 1), and by name (nmfx_path = 2 must not fall back)."""
    from oracle import nmf_oracle as O
    V, W0, H0 = synth(m, n, K, T=T)
    cfg = dict(divergence=div, W_init=W0, H_init=H0, maxiter=6, tolerance=1e-12, W_sparsity=0.01, H_sparsity=0.02)
    if ab:
        cfg["alpha"], cfg["beta"] = ab
    ref = O.cnmf(V, K, T, cfg)
    got = gpu_lib.cnmf(V, K, T, cfg)
    _check(got, ref)
    named = gpu_lib.cnmf(V, K, T, dict(cfg, nmfx_path=2))
    assert np.array_equal(named[0], got[0]) and np.array_equal(named[1], got[1]) and np.array_equal(named[2], got[2])   # the default IS that path
    _check(gpu_lib.cnmf(V, K, T, dict(cfg, nmfx_path=1)), ref)


# the 17 (K, T) pairs round 5 left on the materialised path for IS / alpha-beta: every pair the euclidean / KL passes serve, fused_supported_T
_R6_DUAL_PAIRS = [(32, 3), (32, 5), (32, 6), (64, 3), (32, 10), (32, 12), (64, 5), (64, 6), (32, 7), (32, 9), (32, 11), (64, 7), (32, 13), (32, 14), (32, 15), (128, 3), (256, 2)]


@pytest.mark.parametrize("K,T", _R6_DUAL_PAIRS)
def test_cnmf_is_and_alpha_beta_on_the_fused_passes_every_pair(gpu_lib, K, T):
    """cnmf.m:179-194,227-231 on the fused passes for EVERY instantiated (K, T) pair (round 6): IS on a ragged shape and one alpha-beta pair per case, by name
    (nmfx_path = 2 refuses a silent materialised V_hat), against the oracle and the materialised path"""
    from oracle import nmf_oracle as O
    m, n = 132 + 4 * (K % 7), 300 + 7 * T + K // 2            # m a multiple of 4 (the H-step GEMMs on the stored maps), n ragged
    V, W0, H0 = synth(m, n, K, T=T)
    for div, ab in (("is", None), ("ab", [(0.5, 1.5), (2.0, -0.5), (1.0, 0.5)][(K // 32 + T) % 3])):
        cfg = dict(divergence=div, W_init=W0, H_init=H0, maxiter=5, tolerance=1e-12, W_sparsity=0.01, H_sparsity=0.02)
        if ab:
            cfg["alpha"], cfg["beta"] = ab
        ref = O.cnmf(V, K, T, cfg)
        got = gpu_lib.cnmf(V, K, T, dict(cfg, nmfx_path=2))
        _check(got, ref)
        dflt = gpu_lib.cnmf(V, K, T, cfg)
        assert np.array_equal(dflt[0], got[0]) and np.array_equal(dflt[2], got[2])        # the default IS that path
    _check(gpu_lib.cnmf(V, K, T, dict(cfg, nmfx_path=1)), ref)


@pytest.mark.parametrize("div,ab", [("is", None), ("ab", (0.5, 1.5))])
@pytest.mark.parametrize("m,n,K,T", [(512, 700, 20, 8), (128, 333, 40, 4), (256, 1024, 100, 2), (200, 600, 7, 16), (256, 700, 20, 2), (200, 500, 96, 4), (300, 900, 150, 2), (256, 600, 30, 13)])
def test_cnmf_is_and_alpha_beta_any_K_on_the_fused_passes(gpu_lib, div, ab, m, n, K, T):
    """the parametrisation of test_cnmf_any_K_on_the_fused_passes with div in {is, ab}: K that is not a multiple of 32 (or has no pair of its own) is padded with zero,
    fixed components up to an instantiated pair by the blocking call, for IS / alpha-beta as for euclidean / KL -- no shape silently materialises V_hat"""
    from oracle import nmf_oracle as O
    V, W0, H0 = synth(m, n, K, T=T)
    cfg = dict(divergence=div, W_init=W0, H_init=H0, maxiter=6, tolerance=1e-12, W_sparsity=0.01, H_sparsity=0.02)
    if ab:
        cfg["alpha"], cfg["beta"] = ab
    ref = O.cnmf(V, K, T, cfg)
    got = gpu_lib.cnmf(V, K, T, dict(cfg, nmfx_path=2))
    assert got[0].shape == (m, K, T) and got[1].shape == (K, n)
    _check(got, ref)
    _check(gpu_lib.cnmf(V, K, T, cfg), ref)
    if K >= 7:
        k1 = K // 3
        cfg2 = dict(cfg, W_init=[W0[:, :k1], W0[:, k1:]], H_init=[H0[:k1], H0[k1:]], W_sparsity=[0.02, 0.0], H_sparsity=[0.0, 0.0], H_fixed=[False, True], maxiter=4)
        ref2 = O.cnmf(V, [k1, K - k1], T, cfg2)
        got2 = gpu_lib.cnmf(V, [k1, K - k1], T, dict(cfg2, nmfx_path=2))
        assert rel_fro(np.concatenate(got2[0], 1), np.concatenate(ref2[0], 1)) <= 1e-5 and rel_fro(np.vstack(got2[1]), np.vstack(ref2[1])) <= 1e-5
        assert rel_fro(got2[2], ref2[2]) <= 1e-5


def test_cnmf_is_on_the_fused_passes_fixed_factors_sources_and_stop(gpu_lib):
    """the same path with what the engine's generic update kernels add around it: two sources with one fixed W / one fixed H, all of W fixed (the S pass is then
    cost-only), all of H fixed, and the stop rule with the cost lagging one pass"""
    from oracle import nmf_oracle as O
    m, n, K, T = 192, 640, 64, 4
    V, W0, H0 = synth(m, n, K, T=T)
    base = dict(divergence="is", maxiter=5, tolerance=1e-12)
    for extra in (dict(W_sparsity=[0.05, 0.0], H_fixed=[False, True]), dict(W_fixed=[True, False], H_sparsity=[0.0, 0.03])):
        cfg = dict(base, W_init=[W0[:, :24], W0[:, 24:]], H_init=[H0[:24], H0[24:]], **extra)
        _check(gpu_lib.cnmf(V, [24, 40], T, dict(cfg, nmfx_path=2)), O.cnmf(V, [24, 40], T, cfg))
    for fixed in ("W_fixed", "H_fixed"):
        cfg = dict(base, W_init=W0, H_init=H0, **{fixed: True})
        _check(gpu_lib.cnmf(V, K, T, dict(cfg, nmfx_path=2)), O.cnmf(V, K, T, cfg))
    cfg = dict(divergence="is", W_init=W0, H_init=H0, maxiter=200, tolerance=100.0)
    got, ref = gpu_lib.cnmf(V, K, T, dict(cfg, nmfx_path=2)), O.cnmf(V, K, T, cfg)
    assert len(ref[2]) < 200
    _check_stop(got[2], ref[2], 100.0)
    if len(got[2]) == len(ref[2]):
        assert rel_fro(got[0], ref[0]) <= TOL and rel_fro(got[1], ref[1]) <= TOL


@pytest.mark.parametrize("div,ab,m,n,K", [("is", None, 200, 600, 320), ("is", None, 256, 700, 512), ("ab", (0.5, 1.5), 200, 600, 320), ("ab", (1.5, -1.5), 256, 700, 512),
                                          ("is", None, 257, 4160, 288), ("ab", (1.0, 0.5), 384, 1024, 800), ("is", None, 130, 2051, 2048), ("ab", (2.0, -0.5), 512, 768, 544)])
def test_nmf_is_and_alpha_beta_above_256_in_column_blocks(gpu_lib, div, ab, m, n, K):
    """IS / alpha-beta nmf with K > 256 (nmf.m:154-164,185-195 have no K limit) without V_hat, round 5 (engine.dualw): S = W*H accumulated over <= 256-wide column
    blocks (functor 7), the last block runs the first element map with its cost terms on the accumulated S and stores BOTH maps' values (functors 19 / 20), numerator
    passes block by block on either buffer, W'*A and W'*B as plain products.  Two to eight blocks, ragged shapes, sparsity terms on: against the oracle at the
    contract and against the materialised path (nmfx_path = 1)."""
    from oracle import nmf_oracle as O
    V, W0, H0 = synth(m, n, K)
    cfg = dict(divergence=div, W_init=W0, H_init=H0, maxiter=6, tolerance=1e-12, W_sparsity=0.01, H_sparsity=0.02)
    if ab:
        cfg["alpha"], cfg["beta"] = ab
    ref = O.nmf(V, K, cfg)
    _check(gpu_lib.nmf(V, K, cfg), ref)
    _check(gpu_lib.nmf(V, K, dict(cfg, nmfx_path=1)), ref)


def test_nmf_is_above_256_fixed_factors_sources_and_stop(gpu_lib):
    from oracle import nmf_oracle as O
    m, n, K = 256, 900, 320
    V, W0, H0 = synth(m, n, K)
    base = dict(divergence="is", maxiter=5, tolerance=1e-12)
    cfg = dict(base, W_init=[W0[:, :120], W0[:, 120:]], H_init=[H0[:120], H0[120:]], W_sparsity=[0.05, 0.0], H_fixed=[False, True])
    _check(gpu_lib.nmf(V, [120, 200], cfg), O.nmf(V, [120, 200], cfg))
    for fixed in ("W_fixed", "H_fixed"):
        cfg = dict(base, W_init=W0, H_init=H0, **{fixed: True})
        _check(gpu_lib.nmf(V, K, cfg), O.nmf(V, K, cfg))
    # column shards of nmf carry no halos: every shard's engine takes the same path, [N | P] is what the shards sum (three ragged shards on one device)
    cfg = dict(base, W_init=W0, H_init=H0, W_sparsity=0.02)
    ref = O.nmf(V, K, cfg)
    _check(gpu_lib.nmf(V, K, dict(cfg, nmfx_gpus=[0, 0, 0])), ref)
    cfg = dict(divergence="ab", alpha=0.5, beta=1.5, W_init=W0, H_init=H0, maxiter=4, tolerance=1e-12)
    _check(gpu_lib.nmf(V, K, dict(cfg, nmfx_gpus=[0, 0])), O.nmf(V, K, cfg))


@pytest.mark.parametrize("sW,sH", [(0.0, 0.5), (0.4, 0.6), (0.0, 0.0), (0.3, 0.0)])
@pytest.mark.parametrize("K", [64, 96])
def test_nmfsc_fused_path_matches_oracle(gpu_lib, sW, sH, K):
    """nmfsc on the fused kernels (objective = fused cost pass, gradients in Gram form) vs the oracle: identical line-search branches."""
    from oracle import nmf_oracle as O
    V, W0, H0 = synth(256, 1024, K)
    cfg = dict(W_init=W0, H_init=H0, maxiter=20, tolerance=1e-12)
    if sW:
        cfg["W_sparsity"] = sW
    if sH:
        cfg["H_sparsity"] = sH
    i0, i1, i2 = {}, {}, {}
    ref = O.nmfsc(V, K, cfg, info=i0)
    got = gpu_lib.nmfsc(V, K, dict(cfg, nmfx_path=2), info=i1)
    gen = gpu_lib.nmfsc(V, K, dict(cfg, nmfx_path=1), info=i2)
    assert i1["triesH"] == i0["triesH"] and i1["triesW"] == i0["triesW"]
    assert i2["triesH"] == i0["triesH"] and i2["triesW"] == i0["triesW"]
    _check(got, ref)
    _check(gen, ref)


@pytest.mark.parametrize("div", ["euclidean", "frobenius"])
@pytest.mark.parametrize("m,n,K,T", [(256, 1024, 16, 8), (96, 130, 5, 3), (128, 256, 8, 1), (200, 333, 7, 4)])
def test_cnmf_gram_form_matches_materialised_and_oracle(gpu_lib, div, m, n, K, T):
    """cnmf euclidean without V_hat in HBM (denominators from the KT x KT Grams) == the materialised GEMM path == oracle."""
    from oracle import nmf_oracle as O
    V, W0, H0 = synth(m, n, K, T=T)
    cfg = dict(divergence=div, W_init=W0 if T > 1 else W0[:, :, 0], H_init=H0, maxiter=20, tolerance=1e-12, W_sparsity=0.01, H_sparsity=0.02)
    ref = O.cnmf(V, K, T, cfg)
    gram = gpu_lib.cnmf(V, K, T, dict(cfg, nmfx_path=0))
    mat = gpu_lib.cnmf(V, K, T, dict(cfg, nmfx_path=1))
    _check(gram, ref)
    _check(mat, ref)
    _check(gram, mat)


@pytest.mark.parametrize("div", ["euclidean", "kl", "frobenius"])
@pytest.mark.parametrize("m,n,K,T", [(513, 700, 20, 8), (129, 333, 40, 4), (256, 1024, 100, 2), (200, 600, 7, 16), (256, 700, 20, 2), (200, 500, 96, 4), (130, 400, 33, 10), (300, 900, 150, 2), (257, 600, 30, 13)])   # (the last three: the nearest pair is not the next multiple of 32, or does not exist)
def test_cnmf_any_K_on_the_fused_passes(gpu_lib, div, m, n, K, T):
    """cnmf with K that is not a multiple of 32: the blocking call pads every time slice of W (and the rows of H) with zero, fixed components up to an
    instantiated (K, T) pair -- K = 20, T = 8 runs as (32, 8) -- so spectrogram-sized problems with any number of bases take the fused shift-sum passes
    (nmfx_path = 2 refuses anything else).  The padding adds exact zeros to every product, is skipped by the slab normalisation (cnmf.m:161-165, 196-199:
    0/0) and is stripped on the way out; against the oracle, the GEMM path, and with sparsity / fixed factors / two sources."""
    from oracle import nmf_oracle as O
    V, W0, H0 = synth(m, n, K, T=T)
    cfg = dict(divergence=div, W_init=W0, H_init=H0, maxiter=8, tolerance=1e-12, W_sparsity=0.01, H_sparsity=0.02)
    ref = O.cnmf(V, K, T, cfg)
    if (K, T) == (33, 10):                                        # 33 > 32 = the only instantiated K for T = 10: refused by name, served by the GEMM path by default
        with pytest.raises(Exception, match="not eligible"):
            gpu_lib.cnmf(V, K, T, dict(cfg, nmfx_path=2))
        _check(gpu_lib.cnmf(V, K, T, cfg), ref)
        return
    got = gpu_lib.cnmf(V, K, T, dict(cfg, nmfx_path=2))
    assert got[0].shape == (m, K, T) and got[1].shape == (K, n)
    _check(got, ref)
    _check(gpu_lib.cnmf(V, K, T, cfg), ref)                       # the default takes the same route
    _check(gpu_lib.cnmf(V, K, T, dict(cfg, nmfx_path=1)), ref)
    if K >= 7 and div != "frobenius":
        k1 = K // 3
        cfg2 = dict(divergence=div, W_init=[W0[:, :k1], W0[:, k1:]], H_init=[H0[:k1], H0[k1:]], W_sparsity=[0.02, 0.0], H_fixed=[False, True], maxiter=5, tolerance=1e-12)
        ref2 = O.cnmf(V, [k1, K - k1], T, cfg2)
        got2 = gpu_lib.cnmf(V, [k1, K - k1], T, dict(cfg2, nmfx_path=2))
        assert rel_fro(np.concatenate(got2[0], 1), np.concatenate(ref2[0], 1)) <= 1e-5 and rel_fro(np.vstack(got2[1]), np.vstack(ref2[1])) <= 1e-5
        assert rel_fro(got2[2], ref2[2]) <= 1e-6


# ---- cnmfsc: convolutive NMF with Hoyer sparseness, reference quirks included ---------------------
@pytest.mark.parametrize("sW,sH", [(0.0, 0.0), (0.0, 0.5), (0.3, 0.0), (0.4, 0.6)])
@pytest.mark.parametrize("m,n,K,T", [(96, 200, 6, 4), (128, 256, 8, 1)])
def test_cnmfsc_matches_oracle(gpu_lib, sW, sH, m, n, K, T):
    from oracle import nmf_oracle as O
    V, W0, H0 = synth(m, n, K, T=T)
    cfg = dict(W_init=W0, H_init=H0, maxiter=12, tolerance=1e-12)
    if sW:
        cfg["W_sparsity"] = sW
    if sH:
        cfg["H_sparsity"] = sH
    i0, i1 = {}, {}
    ref = O.cnmfsc(2.0 * V, K, T, cfg, info=i0)
    got = gpu_lib.cnmfsc(2.0 * V, K, T, cfg, info=i1)
    assert i1["triesH"] == i0["triesH"] and i1["triesW"] == i0["triesW"]      # incl. the 665-try step-size underflow of the sparse-W branch
    _check(got, ref)


def _cnmfsc_fused_cases():
    for m, n, K, T in [(256, 1024, 32, 4), (192, 777, 64, 2), (320, 2048, 64, 8), (129 * 4, 1031, 32, 8)]:
        for sW, sH in [(0.0, 0.5), (0.0, 0.0), (0.6, 0.0), (0.4, 0.6)]:
            if sW > 0 and m * n * K * T > (1 << 27):
                continue        # the sparse-W search ends by step-size underflow after 665 tries per slice (cnmfsc.m:235): the oracle alone takes 10 - 40 s there
            yield m, n, K, T, sW, sH


@pytest.mark.parametrize("m,n,K,T,sW,sH", list(_cnmfsc_fused_cases()))
def test_cnmfsc_fused_passes_match_oracle(gpu_lib, sW, sH, m, n, K, T):
    """cnmfsc.m:155-277 with every whole-matrix contraction on the register-stationary kernels (nmfx_path = 2; the default above the float64-gradient
    sizes): objectives of the H line search without a stored V_hat, V_hat + objective in one pass (cnmfsc.m:215,269), the T products V*rshift_t(H)' in
    one pass, the multiplicative W branch from the Gram of the stacked shifts without V_hat (cnmfsc.m:257-263, aux.hip::cnmfsc_w_slices), dH through Q + shift-sum.
    Identical line-search tries; also against the two-operand GEMM path (nmfx_path = 1), which keeps V_hat = max(V_hat + dW*rshift_t(H), 0) slice by slice."""
    from oracle import nmf_oracle as O
    V, W0, H0 = synth(m, n, K, T=T)
    cfg = dict(W_init=W0, H_init=H0, maxiter=8, tolerance=1e-12)
    if sW:
        cfg["W_sparsity"] = sW
    if sH:
        cfg["H_sparsity"] = sH
    i0, i1, i2 = {}, {}, {}
    ref = O.cnmfsc(2.0 * V, K, T, cfg, info=i0)
    got = gpu_lib.cnmfsc(2.0 * V, K, T, dict(cfg, nmfx_path=2), info=i1)
    assert i1["triesH"] == i0["triesH"] and i1["triesW"] == i0["triesW"]
    _check(got, ref)
    gen = gpu_lib.cnmfsc(2.0 * V, K, T, dict(cfg, nmfx_path=1), info=i2)
    assert i2["triesH"] == i1["triesH"] and rel_fro(got[0], gen[0]) <= 5e-6 and rel_fro(got[1], gen[1]) <= 5e-6


@pytest.mark.parametrize("m,n,K,T", [(8200, 256, 64, 2), (8192, 192, 32, 3), (260, 640, 128, 2), (512, 512, 32, 16), (4096, 320, 64, 8), (132, 200, 64, 3)])
def test_cnmfsc_multiplicative_w_branch_every_slice_kernel(gpu_lib, m, n, K, T):
    """cnmfsc.m:257-263 on the fused passes (no W / H sparsity: both branches multiplicative), one case per instantiation of the slice-loop kernel
    (aux.hip::cnmfsc_w_slices: K = 32 / 64 / 128, 8 or 16 rows of W per workgroup -- 16 from m >= 8177 on --, G in chunks of 32 or 64 rows), ragged m, K*T up to
    512 and an odd multiple of 32; against the oracle, whose V_hat = max(V_hat + dW*rshift_t(H), 0) this path never forms, and against the GEMM path, which does."""
    from oracle import nmf_oracle as O
    V, W0, H0 = synth(m, n, K, T=T)
    cfg = dict(W_init=W0, H_init=H0, maxiter=6, tolerance=1e-12)
    ref = O.cnmfsc(V, K, T, cfg)
    got = gpu_lib.cnmfsc(V, K, T, dict(cfg, nmfx_path=2))
    _check(got, ref)
    gen = gpu_lib.cnmfsc(V, K, T, dict(cfg, nmfx_path=1))
    assert rel_fro(got[0], gen[0]) <= 5e-6 and rel_fro(got[1], gen[1]) <= 5e-6


# ---- lnmf on the generic and the fused KL kernels ------------------------------------------------
@pytest.mark.parametrize("m,n,K,path", [(96, 160, 8, 0), (256, 1024, 64, 2), (256, 1024, 64, 1), (128, 32768, 64, 2)])
def test_lnmf_matches_oracle(gpu_lib, m, n, K, path):
    from oracle import nmf_oracle as O
    V, W0, H0 = synth(m, n, K)
    iters = 25 if n < 10000 else 4
    cfg = dict(W_init=W0, H_init=H0, maxiter=iters, tolerance=1e-12)
    W, H, c = gpu_lib.lnmf(V, K, dict(cfg, nmfx_path=path))
    Wr, Hr, cr = O.lnmf(V, K, cfg)
    assert len(c) == len(cr) and rel_fro(W, Wr) < 1e-5 and rel_fro(H, Hr) < 1e-5 and rel_fro(c, cr) < 1e-6
    assert np.allclose(W.sum(0), 1.0, atol=1e-5)                       # L1-normalised columns (lnmf.m:70)


def test_lnmf_untrimmed_cost_and_stop(gpu_lib):
    from oracle import nmf_oracle as O
    V, W0, H0 = synth(96, 160, 8)
    cfg = dict(W_init=W0, H_init=H0, maxiter=60, tolerance=1.0)
    c = gpu_lib.lnmf(V, 8, cfg)[2]
    cr = O.lnmf(V, 8, cfg)[2]
    assert len(c) == 60 and np.count_nonzero(cr) < 60                  # zeros after the stop (lnmf.m:84-86)
    _check_stop(c[:np.count_nonzero(c)], cr[:np.count_nonzero(cr)], 1.0, le=True)


# ---- constrainednmf + SortDictionary ---------------------------------------------------------------
def _labels(n, n_classes, frac_unlabelled, seed):
    rs = np.random.RandomState(seed)
    lab = rs.randint(0, n_classes, size=n) * 2 + 3          # non-consecutive ids
    lab[rs.rand(n) < frac_unlabelled] = -1
    return lab


@pytest.mark.parametrize("div,cfgx", [("euclidean", {}), ("kl", {}), ("is", {}), ("kl", dict(Z_sparsity=0.1, W_sparsity=0.05)),
                                       ("euclidean", dict(Z_fixed=True)), ("kl", dict(W_fixed=True))])
@pytest.mark.parametrize("m,n,K,path", [(96, 200, 7, 0), (256, 384, 64, 2), (256, 384, 64, 1)])
def test_constrainednmf_matches_oracle(gpu_lib, div, cfgx, m, n, K, path):
    from oracle import nmf_oracle as O
    V, W0, _ = synth(m, n, K)
    lab = _labels(n, 5, 0.3, 5)
    nz = int(np.count_nonzero(lab == -1)) + len(np.unique(lab[lab >= 0]))
    Z0 = np.fmax(np.random.RandomState(9).rand(K, nz), 2.0 ** -52)
    cfg = dict(divergence=div, W_init=W0, Z_init=Z0, maxiter=15, tolerance=1e-12, **cfgx)
    W, H, Z, A, cost = gpu_lib.constrainednmf(V, lab, K, dict(cfg, nmfx_path=path))
    W0_, H0_, Z0_, A0_, cost0 = O.constrainednmf(V, lab, K, cfg)
    assert np.array_equal(A, A0_)
    assert rel_fro(W, W0_) <= TOL and rel_fro(H, H0_) <= TOL and rel_fro(Z, Z0_) <= TOL, (rel_fro(W, W0_), rel_fro(H, H0_), rel_fro(Z, Z0_))
    assert len(cost) == len(cost0) and rel_fro(cost, cost0) <= (1e-5 if div == "is" else 1e-6)
    assert rel_fro(H, Z @ A) <= 1e-6


def test_constrainednmf_edge_cases(gpu_lib):
    from oracle import nmf_oracle as O
    V, W0, H0 = synth(64, 90, 6)
    # nothing labelled: constrainednmf == nmf (A = I)
    W, H, Z, A, cost = gpu_lib.constrainednmf(V, -np.ones(90, dtype=int), 6, dict(W_init=W0, Z_init=H0, maxiter=10, tolerance=1e-12))
    Wn, Hn, costn = gpu_lib.nmf(V, 6, dict(W_init=W0, H_init=H0, maxiter=10, tolerance=1e-12))
    assert np.array_equal(A, np.eye(90)) and rel_fro(W, Wn) < 1e-6 and rel_fro(H, Hn) < 1e-6 and rel_fro(cost, costn) < 1e-6
    # everything labelled, one class: Z is K x 1
    W, H, Z, A, cost = gpu_lib.constrainednmf(V, np.full(90, 4), 6, dict(divergence="kl", W_init=W0, Z_init=H0[:, :1], maxiter=8, tolerance=1e-12))
    Wo, Ho, Zo, Ao, costo = O.constrainednmf(V, np.full(90, 4), 6, dict(divergence="kl", W_init=W0, Z_init=H0[:, :1], maxiter=8, tolerance=1e-12))
    assert Z.shape == (6, 1) and rel_fro(W, Wo) <= TOL and rel_fro(Z, Zo) <= TOL and rel_fro(cost, costo) <= 1e-6
    # alpha-beta: dual form runs (2 iterations: it degenerates like nmf's), alpha ~= 0 is refused with the reference line
    lab = _labels(90, 3, 0.5, 1)
    nz = int(np.count_nonzero(lab == -1)) + len(np.unique(lab[lab >= 0]))
    cfg = dict(divergence="ab", alpha=0.0, beta=1.0, W_init=W0, Z_init=H0[:, :nz], maxiter=2, tolerance=1e-12)
    got, ref = gpu_lib.constrainednmf(V, lab, 6, cfg), O.constrainednmf(V, lab, 6, cfg)
    assert rel_fro(got[0], ref[0]) <= TOL and rel_fro(got[2], ref[2]) <= TOL
    with pytest.raises(Exception, match="constrainednmf.m:229"):
        gpu_lib.constrainednmf(V, lab, 6, dict(divergence="ab", alpha=0.5, beta=0.5, maxiter=2))
    with pytest.raises(ValueError, match="Length of the label vector"):
        gpu_lib.constrainednmf(V, np.zeros(5), 6)
    with pytest.raises(ValueError, match="alpha = 0 and beta = 0"):
        gpu_lib.constrainednmf(V, lab, 6, dict(divergence="ab", alpha=0, beta=0))
    W, H, Z, A, cost = gpu_lib.constrainednmf(V, lab, 6, dict(seed=4, maxiter=0))      # defaults: rand inits, 100 iterations, 1e-3
    assert len(cost) <= 100 and Z.shape == (6, nz) and np.all(np.diff(cost) <= 1e-6 * cost[0])


def test_sort_dictionary_matches_oracle(gpu_lib):
    from oracle import nmf_oracle as O
    g = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "sort_dictionary.npz"))
    Ws, Hs = gpu_lib.SortDictionary(g["W"], g["H"])
    assert np.array_equal(Ws, g["W_sorted"]) and np.array_equal(Hs, g["H_sorted"])
    for m, K, seed in [(513, 40, 0), (4096, 300, 1), (7, 3, 2), (1, 5, 3)]:
        rs = np.random.RandomState(seed)
        W = np.abs(rs.randn(m, K)) * (rs.rand(m, K) < 0.4)
        W[:, K // 2] = 0.0                                  # an all-zero atom: every cumsum <= 0, centre = m
        H = rs.rand(K, 33)
        Ws, Hs = gpu_lib.SortDictionary(W, H)
        Wo, Ho = O.sort_dictionary(W, H)
        assert np.array_equal(Ws, Wo) and np.array_equal(Hs, Ho)
    assert gpu_lib.SortDictionary(g["W"])[1] is None


# ---- edge shapes and degenerate inputs (the reference has no tests; these follow its semantics through the oracle) ---------
@pytest.mark.parametrize("m,n,K", [(1, 1, 1), (3, 5, 1), (2, 40, 1), (40, 1, 2), (5, 4, 7), (129, 127, 3), (128, 128, 64), (130, 256, 64)])
@pytest.mark.parametrize("div", ["euclidean", "kl", "is"])
def test_nmf_edge_shapes(gpu_lib, m, n, K, div):
    from oracle import nmf_oracle as O
    V, W0, H0 = synth(m, n, K)
    cfg = dict(divergence=div, W_init=W0, H_init=H0, maxiter=6, tolerance=1e-12)
    _check(gpu_lib.nmf(V, K, cfg), O.nmf(V, K, cfg), cost_atol=1e-6 * float((V ** 2).sum()))


@pytest.mark.parametrize("m,n,K,T", [(4, 3, 1, 3), (7, 9, 2, 1), (33, 64, 4, 9), (64, 40, 32, 2), (128, 128, 64, 3)])
def test_cnmf_edge_shapes(gpu_lib, m, n, K, T):
    from oracle import nmf_oracle as O
    V, W0, H0 = synth(m, n, K, T=T)
    for div in ("euclidean", "kl"):
        cfg = dict(divergence=div, W_init=W0 if T > 1 else W0[:, :, 0], H_init=H0, maxiter=5, tolerance=1e-12)
        _check(gpu_lib.cnmf(V, K, T, cfg), O.cnmf(V, K, T, cfg), cost_atol=1e-6 * float((V ** 2).sum()))
    with pytest.raises(Exception):
        gpu_lib.cnmf(V, K, n + 1, dict(maxiter=1))          # context longer than the data


def test_degenerate_inputs(gpu_lib):
    from oracle import nmf_oracle as O
    V, W0, H0 = synth(64, 96, 4)
    # everything fixed: nothing moves, the cost is constant and the strict-decrease stop rule (nmf.m:221) never fires
    cfg = dict(W_init=W0, H_init=H0, W_fixed=True, H_fixed=True, maxiter=5)
    W, H, c = gpu_lib.nmf(V, 4, cfg)
    Wo, Ho, co = O.nmf(V, 4, cfg)
    assert len(c) == len(co) == 5 and np.allclose(c, c[0]) and rel_fro(W, Wo) < 1e-6 and rel_fro(H, H0) < 1e-7
    # zeros in V: KL cost is NaN every iteration (0*log(0)), W/H stay finite where the oracle's do
    Vz = V.copy()
    Vz[::7, ::5] = 0.0
    cfg = dict(divergence="kl", W_init=W0, H_init=H0, maxiter=4)
    W, H, c = gpu_lib.nmf(Vz, 4, cfg)
    Wo, Ho, co = O.nmf(Vz, 4, cfg)
    assert len(c) == len(co) == 4 and np.all(np.isnan(c)) and np.all(np.isnan(co))
    assert rel_fro(W, Wo) < 1e-5 and rel_fro(H, Ho) < 1e-5
    # an all-zero column of V and a zero row of H_init: euclidean keeps them at zero
    V2, H2 = V.copy(), H0.copy()
    V2[:, 3] = 0.0
    H2[1, :] = 0.0
    cfg = dict(W_init=W0, H_init=H2, maxiter=5, tolerance=1e-12)
    got, ref = gpu_lib.nmf(V2, 4, cfg), O.nmf(V2, 4, cfg)
    assert np.array_equal(np.isnan(got[0]), np.isnan(ref[0])) and np.array_equal(np.isnan(got[1]), np.isnan(ref[1]))
    ok = ~np.isnan(ref[1])
    assert np.allclose(got[1][ok], ref[1][ok], rtol=1e-4, atol=1e-6) and np.all(got[1][1, ~np.isnan(got[1][1])] == 0)
    # maxiter = 1 and a huge tolerance: one iteration, one cost entry (the stop rule needs iter > 1)
    W, H, c = gpu_lib.nmf(V, 4, dict(W_init=W0, H_init=H0, maxiter=1, tolerance=1e9))
    assert len(c) == 1
    W, H, c = gpu_lib.nmf(V, 4, dict(W_init=W0, H_init=H0, maxiter=50, tolerance=1e9))
    assert len(c) == 2                                      # stops at the first comparison it is allowed to make


@pytest.mark.parametrize("div", ["euclidean", "kl"])
def test_fused_path_with_padded_K_multi_source(gpu_lib, div):
    """K = 12 (three sources, sparsity, fixed ones) on a tileable shape: the blocking API pads K to 32 with zero fixed components
    and runs the fused kernels; the padding must be invisible."""
    from oracle import nmf_oracle as O
    V, W0, H0 = synth(256, 384, 12)
    Ks = [3, 4, 5]
    cfg = dict(divergence=div, W_init=[W0[:, :3], W0[:, 3:7], W0[:, 7:]], H_init=[H0[:3], H0[3:7], H0[7:]], W_sparsity=[0.1, 0.0, 0.05],
               H_sparsity=[0.0, 0.2, 0.0], W_fixed=[False, True, False], H_fixed=[False, False, True], maxiter=25, tolerance=1e-12)
    ref = O.nmf(V, Ks, cfg)
    _check(gpu_lib.nmf(V, Ks, dict(cfg, nmfx_path=2)), ref)
    _check(gpu_lib.nmf(V, Ks, dict(cfg, nmfx_path=1)), ref)
    lab = _labels(384, 4, 0.4, 3)
    nz = int(np.count_nonzero(lab == -1)) + len(np.unique(lab[lab >= 0]))
    Z0 = np.fmax(np.random.RandomState(9).rand(12, nz), 2.0 ** -52)
    c2 = dict(divergence=div, W_init=W0, Z_init=Z0, maxiter=10, tolerance=1e-12)
    got, want = gpu_lib.constrainednmf(V, lab, 12, dict(c2, nmfx_path=2)), O.constrainednmf(V, lab, 12, c2)
    assert rel_fro(got[0], want[0]) <= TOL and rel_fro(got[2], want[2]) <= TOL and rel_fro(got[4], want[4]) <= 1e-6


# ---- ragged shapes on the fused kernels (masked edges): m, n arbitrary, K arbitrary <= 256 ---------------------------------------
@pytest.mark.parametrize("div", ["kl", "euclidean"])
@pytest.mark.parametrize("m,n,K,iters", [(129, 131, 32, 10), (513, 1000, 40, 10), (200, 70, 64, 12), (64, 64, 32, 8), (1025, 300, 100, 8),
                                         (65, 65, 7, 10), (128, 65, 64, 8), (129, 128, 96, 8), (300, 4097, 160, 5), (2049, 257, 256, 4)])
def test_nmf_fused_ragged_shapes(gpu_lib, div, m, n, K, iters):
    from oracle import nmf_oracle as O
    V, W0, H0 = synth(m, n, K)
    cfg = dict(divergence=div, W_init=W0, H_init=H0, maxiter=iters, tolerance=1e-12, W_sparsity=0.01, H_sparsity=0.02)
    ref = O.nmf(V, K, cfg)
    fused = gpu_lib.nmf(V, K, dict(cfg, nmfx_path=2))
    _check(fused, ref)
    _check(gpu_lib.nmf(V, K, cfg), ref)        # auto picks the same kernels for these shapes


def test_fused_ragged_other_algorithms(gpu_lib):
    from oracle import nmf_oracle as O
    V, W0, H0 = synth(257, 333, 24)
    cfg = dict(W_init=W0 / W0.sum(0), H_init=H0, maxiter=10, tolerance=1e-12)
    got, ref = gpu_lib.lnmf(V, 24, dict(cfg, nmfx_path=2)), O.lnmf(V, 24, cfg)
    record_err(W=rel_fro(got[0], ref[0]), H=rel_fro(got[1], ref[1]), cost=rel_fro(got[2], ref[2]))
    assert rel_fro(got[0], ref[0]) <= TOL and rel_fro(got[1], ref[1]) <= TOL and rel_fro(got[2], ref[2]) <= 1e-6
    lab = _labels(333, 5, 0.3, 8)
    nz = int(np.count_nonzero(lab == -1)) + len(np.unique(lab[lab >= 0]))
    Z0 = np.fmax(np.random.RandomState(9).rand(24, nz), 2.0 ** -52)
    for div in ("kl", "euclidean"):
        c2 = dict(divergence=div, W_init=W0, Z_init=Z0, maxiter=8, tolerance=1e-12, Z_sparsity=0.05)
        got, want = gpu_lib.constrainednmf(V, lab, 24, dict(c2, nmfx_path=2)), O.constrainednmf(V, lab, 24, c2)
        record_err(W=rel_fro(got[0], want[0]), H=rel_fro(got[2], want[2]), cost=rel_fro(got[4], want[4]))
        assert rel_fro(got[0], want[0]) <= TOL and rel_fro(got[2], want[2]) <= TOL and rel_fro(got[4], want[4]) <= 1e-6
    # zeros in V on a ragged shape: NaN cost exactly like the reference, finite factors where the reference's are
    Vz = V.copy()
    Vz[::9, ::7] = 0.0
    cfg = dict(divergence="kl", W_init=W0, H_init=H0, maxiter=3)
    got, ref = gpu_lib.nmf(Vz, 24, dict(cfg, nmfx_path=2)), O.nmf(Vz, 24, cfg)
    assert np.all(np.isnan(got[2])) and np.all(np.isnan(ref[2])) and rel_fro(got[0], ref[0]) < TOL and rel_fro(got[1], ref[1]) < TOL


@pytest.mark.parametrize("sW,sH", [(0.0, 0.5), (0.4, 0.6), (0.0, 0.0)])
def test_nmfsc_fused_ragged(gpu_lib, sW, sH):
    from oracle import nmf_oracle as O
    V, W0, H0 = synth(129, 200, 32)
    cfg = dict(W_init=W0, H_init=H0, maxiter=12, tolerance=1e-12)
    if sW:
        cfg["W_sparsity"] = sW
    if sH:
        cfg["H_sparsity"] = sH
    i0, i1 = {}, {}
    ref = O.nmfsc(V, 32, cfg, info=i0)
    got = gpu_lib.nmfsc(V, 32, dict(cfg, nmfx_path=2), info=i1)
    assert i1["triesH"] == i0["triesH"] and i1["triesW"] == i0["triesW"]
    _check(got, ref)


@pytest.mark.parametrize("m,n,K,sW,sH,iters,fixed", [(194, 635, 3, 0.0, 0.7, 3, None), (431, 641, 3, 0.0, 0.5, 7, None), (325, 202, 3, 0.0, 0.7, 5, None),
                                                     (334, 280, 3, 0.3, 0.4, 6, "W_fixed"), (292, 252, 3, 0.0, 0.7, 6, "W_fixed"), (300, 500, 1, 0.0, 0.5, 5, None),
                                                     (256, 384, 5, 0.4, 0.6, 6, None), (130, 700, 8, 0.5, 0.0, 6, None), (257, 129, 2, 0.3, 0.0, 6, "H_fixed"), (257, 129, 2, 0.3, 0.7, 3, "H_fixed")])   # (the last one converges in 3 iterations: beyond them the
                                                     # line search decides on cost differences of 4e-10 relative, below what fp32 storage of W resolves)
def test_nmfsc_small_K_float64_gradients(gpu_lib, m, n, K, sW, sH, iters, fixed):
    """A handful of components: the Hoyer projection after a gradient step amplifies perturbations (70x on the K = 3 cases below, which the
    fp32-accumulated gradients of the fused kernels missed by 1.2e-5 in scripts/fuzz_campaign_sc.py), so K <= 8 takes its gradients and
    objective in float64 (aux.hip::smallk_grad) and steps along a float64 direction inside projfunc.  Held to a TIGHTER bar than the contract."""
    from oracle import nmf_oracle as O
    V, W0, H0 = synth(m, n, K)
    cfg = dict(W_init=W0, H_init=H0, maxiter=iters, tolerance=1e-300)
    if sW:
        cfg["W_sparsity"] = sW
    if sH:
        cfg["H_sparsity"] = sH
    if fixed:
        cfg[fixed] = True
    i0, i1 = {}, {}
    ref = O.nmfsc(V, K, cfg, info=i0)
    got = gpu_lib.nmfsc(V, K, cfg, info=i1)
    assert i1["triesH"] == i0["triesH"] and i1["triesW"] == i0["triesW"]
    _check(got, ref, tol=3e-6)


@pytest.mark.parametrize("m,n,K,T,iters", [(71, 218, 32, 4, 5), (145, 390, 32, 2, 7), (388, 156, 20, 4, 4), (200, 300, 16, 3, 6)])
@pytest.mark.parametrize("h_fixed", [True, False])
def test_cnmfsc_sparse_W_float64_gradients(gpu_lib, m, n, K, T, iters, h_fixed):
    """cnmfsc's sparse-W line search on short columns: the Hoyer projection amplifies the accumulation noise of an fp32 MFMA contraction over n
    (W off by 1.3e-5 / 1.6e-5 on the first three problems in scripts/fuzz_campaign_sc.py), so small problems take dW from a float64 VALU
    kernel (aux.hip::resid_xht64) and step along a float64 direction.  Held to a tighter bar than the contract."""
    from oracle import nmf_oracle as O
    V, W0, H0 = synth(m, n, K, T=T)
    cfg = dict(W_init=W0, H_init=H0, tolerance=1e-300, maxiter=iters, W_sparsity=0.6)
    if h_fixed:
        cfg["H_fixed"] = True
    i0, i1 = {}, {}
    ref = O.cnmfsc(V, K, T, cfg, info=i0)
    got = gpu_lib.cnmfsc(V, K, T, cfg, info=i1)
    assert i1["triesW"] == i0["triesW"] and i1["triesH"] == i0["triesH"]
    _check(got, ref, tol=3e-6)


@pytest.mark.parametrize("m,n,K,T,iters,in_contract", [(222, 100, 32, 2, 3, True), (222, 100, 32, 3, 3, True), (222, 120, 32, 2, 3, True), (222, 100, 32, 2, 6, False)])
def test_cnmfsc_sparse_W_ill_conditioned_stays_at_the_algorithms_own_sensitivity(gpu_lib, m, n, K, T, iters, in_contract):
    """The one campaign problem of round 5 outside the contract (profiles/r5_29_fuzz_campaign_sc.log: 222 x 100, K = 32, T = 2, W_sparsity 0.6, H fixed -- W at 4.4e-5) and its
    neighbours.  cnmfsc.m:229-249's search on these is not a descent method and amplifies what its gradient carries by two orders of magnitude per iteration: the float64
    ALGORITHM moves 5.6e-6 in three iterations (1.3e-4 in six) when its inputs are merely rounded to fp32, which is what any fp32-storage implementation starts from.  Since
    round 6 the residual the gradient contracts is float64 too (aux.hip::recon_resid64; it was the fp32 V_hat: seven times the input rounding) and the result sits AT that
    intrinsic figure.  The test bounds the class both ways: identical try counts, W within 1.5x of the float64 algorithm's own movement under fp32 input rounding, and inside
    the 1e-5 contract wherever the algorithm itself is."""
    from oracle import nmf_oracle as O
    V, W0, H0 = synth(m, n, K, T=T)
    cfg = dict(W_init=W0, H_init=H0, tolerance=1e-300, maxiter=iters, W_sparsity=0.6, H_fixed=True)
    i0, i1 = {}, {}
    ref = O.cnmfsc(V, K, T, cfg, info=i0)
    got = gpu_lib.cnmfsc(V, K, T, cfg, info=i1)
    f32 = lambda x: np.asarray(x, np.float64).astype(np.float32).astype(np.float64)
    r32 = O.cnmfsc(f32(V / V.max()), K, T, dict(cfg, W_init=f32(W0), H_init=f32(H0)))
    intrinsic = rel_fro(r32[0], ref[0])
    eW = rel_fro(got[0], ref[0])
    assert i1["triesW"] == i0["triesW"] and len(got[2]) == len(ref[2])
    assert eW <= 1.5 * intrinsic + 1e-7, (eW, intrinsic)
    if in_contract:
        assert intrinsic <= 1e-5 and eW <= 1e-5 and rel_fro(got[2], ref[2]) <= 1e-6, (eW, intrinsic)


@pytest.mark.parametrize("m,n,K,iters,sW,sH", [(256, 2048, 16, 30, 0.4, 0.6), (256, 1024, 64, 20, 0.4, 0.6), (500, 700, 128, 8, 0.3, 0.5),
                                                (129, 200, 32, 12, 0.4, 0.6), (400, 300, 20, 8, 0.6, 0.0), (257, 333, 40, 8, 0.0, 0.7)])
def test_nmfsc_small_problems_float64_gradients(gpu_lib, m, n, K, iters, sW, sH):
    """nmfsc on problems of up to 2^27 multiply-adds per evaluation (default path): the residual W*H - V, both gradient contractions and
    the objective in float64 on the VALU, the step along a float64 direction.  An order of magnitude inside the contract (the MFMA path on
    the same problems, nmfx_path=2: 1e-6 ... 3e-6); the fused kernels stay covered by the path=2 tests and the BASELINE-size tests."""
    from oracle import nmf_oracle as O
    V, W0, H0 = synth(m, n, K)
    cfg = dict(W_init=W0, H_init=H0, tolerance=1e-300, maxiter=iters)
    if sW:
        cfg["W_sparsity"] = sW
    if sH:
        cfg["H_sparsity"] = sH
    i0, i1 = {}, {}
    ref = O.nmfsc(V, K, cfg, info=i0)
    got = gpu_lib.nmfsc(V, K, cfg, info=i1)
    assert i1["triesH"] == i0["triesH"] and i1["triesW"] == i0["triesW"]
    _check(got, ref, tol=1e-6, cost_tol=1e-8)


def test_nmf_random_shapes_fuzz(gpu_lib):
    """40 seeded random (m, n, K, divergence, sparsity, fixed) problems between 64 and 400 rows / columns: whichever kernels the
    engine picks (masked-edge fused with padded K, or the pipelined GEMM path when forced) must match the oracle."""
    from oracle import nmf_oracle as O
    rs = np.random.RandomState(2024)
    worst = 0.0
    for trial in range(40):
        m, n = int(rs.randint(64, 400)), int(rs.randint(64, 400))
        K = int(rs.choice([1, 2, 3, 5, 8, 13, 20, 32, 33, 50, 64, 70]))
        div = str(rs.choice(["kl", "euclidean"]))
        V, W0, H0 = synth(m, n, K)
        cfg = dict(divergence=div, W_init=W0, H_init=H0, maxiter=int(rs.randint(2, 9)), tolerance=1e-300)   # no early stop: converged
        if rs.rand() < 0.5:                                                                                  # rank-1 cases sit at fp32 noise
            cfg["W_sparsity"], cfg["H_sparsity"] = float(rs.rand() * 0.1), float(rs.rand() * 0.1)
        if rs.rand() < 0.2:
            cfg["W_fixed"] = True
        elif rs.rand() < 0.2:
            cfg["H_fixed"] = True
        path = int(rs.choice([0, 1, 2]))
        ref = O.nmf(V, K, cfg)
        got = gpu_lib.nmf(V, K, dict(cfg, nmfx_path=path))
        assert len(got[2]) == len(ref[2]), (trial, m, n, K, div, path)
        e = max(rel_fro(got[0], ref[0]), rel_fro(got[1], ref[1]))
        worst = max(worst, e)
        record_err(W=rel_fro(got[0], ref[0]), H=rel_fro(got[1], ref[1]), cost=rel_fro(got[2], ref[2]))
        assert e <= TOL and rel_fro(got[2], ref[2]) <= 1e-6, (trial, m, n, K, div, path, e, rel_fro(got[2], ref[2]))
    assert worst <= TOL


def test_cnmf_random_shapes_fuzz(gpu_lib):
    from oracle import nmf_oracle as O
    rs = np.random.RandomState(77)
    for trial in range(20):
        m, n = int(rs.randint(20, 300)), int(rs.randint(40, 300))
        K, T = int(rs.choice([1, 3, 4, 6, 10, 16, 25, 32])), int(rs.randint(1, 7))
        div = str(rs.choice(["kl", "euclidean", "is"]))
        V, W0, H0 = synth(m, n, K, T=T)
        cfg = dict(divergence=div, W_init=W0 if T > 1 else W0[:, :, 0], H_init=H0, maxiter=int(rs.randint(2, 6)), tolerance=1e-300,
                   W_sparsity=float(rs.rand() * 0.05), H_sparsity=float(rs.rand() * 0.05))
        ref = O.cnmf(V, K, T, cfg)
        got = gpu_lib.cnmf(V, K, T, dict(cfg, nmfx_path=int(rs.choice([0, 1]))))
        assert len(got[2]) == len(ref[2]), (trial, m, n, K, T, div)
        e = dict(W=rel_fro(got[0], ref[0]), H=rel_fro(got[1], ref[1]), cost=rel_fro(got[2], ref[2]))
        record_err(**e)
        assert e["W"] <= TOL and e["H"] <= TOL and e["cost"] <= (1e-5 if div == "is" else 1e-6), (trial, m, n, K, T, div, e)


def test_nmfsc_random_shapes_fuzz(gpu_lib):
    from oracle import nmf_oracle as O
    rs = np.random.RandomState(5)
    for trial in range(10):
        m, n = int(rs.randint(64, 300)), int(rs.randint(64, 300))
        K = int(rs.choice([4, 9, 32, 64]))
        sW, sH = float(rs.choice([0.0, 0.3, 0.6])), float(rs.choice([0.0, 0.4, 0.7]))
        V, W0, H0 = synth(m, n, K)
        cfg = dict(W_init=W0, H_init=H0, maxiter=int(rs.randint(3, 8)), tolerance=1e-300)
        if sW:
            cfg["W_sparsity"] = sW
        if sH:
            cfg["H_sparsity"] = sH
        i0, i1 = {}, {}
        ref = O.nmfsc(V, K, cfg, info=i0)
        got = gpu_lib.nmfsc(V, K, cfg, info=i1)
        assert i1["triesH"] == i0["triesH"] and i1["triesW"] == i0["triesW"], (trial, m, n, K, sW, sH, i0, i1)
        e = dict(W=rel_fro(got[0], ref[0]), H=rel_fro(got[1], ref[1]), cost=rel_fro(got[2], ref[2]))
        record_err(**e)
        assert e["W"] <= TOL and e["H"] <= TOL and e["cost"] <= 1e-6, (trial, m, n, K, sW, sH, e)


# ---- K > 256 and other shapes outside the register-stationary kernels: euclidean still runs without V_hat in HBM (Gram form on the GEMM) ----
@pytest.mark.parametrize("div", ["euclidean", "kl"])
@pytest.mark.parametrize("m,n,K", [(512, 768, 320), (300, 1000, 257), (640, 512, 512)])
def test_nmf_K_above_256(gpu_lib, div, m, n, K):
    from oracle import nmf_oracle as O
    V, W0, H0 = synth(m, n, K)
    cfg = dict(divergence=div, W_init=W0, H_init=H0, maxiter=8, tolerance=1e-12, W_sparsity=0.01)
    ref = O.nmf(V, K, cfg)
    _check(gpu_lib.nmf(V, K, cfg), ref)
    _check(gpu_lib.nmf(V, K, dict(cfg, nmfx_path=1)), ref)


@pytest.mark.parametrize("m,n,K,algo", [(512, 768, 320, "nmf"), (300, 1000, 257, "nmf"), (640, 512, 512, "nmf"), (257, 4160, 288, "nmf"), (384, 1024, 800, "nmf"),
                                        (512, 768, 320, "lnmf"), (200, 900, 300, "src2")])
def test_nmf_kl_K_above_256_in_column_blocks(gpu_lib, m, n, K, algo):
    """nmf.m:152-153,183-184 with K > 256 (the reference has no K limit): S = W*H accumulated over column blocks of <= 256 components by the stationary kernel
    (functors 7 / 8), R = V./S in HBM, V_hat never -- engine path 5.  Any K through the blocking call (padded to a multiple of 32), ragged m / n, three blocks
    (K = 800), lnmf, two sources with sparsity and fixed flags, the stop rule handing back the state of the iteration it fired on, column shards."""
    import torch
    from oracle import nmf_oracle as O
    from nmf_toolbox_amd.engine import Engine, colmajor_to_torch
    V, W0, H0 = synth(m, n, K)
    if K % 32 == 0:
        e = Engine(colmajor_to_torch(V, "cuda:0"), colmajor_to_torch(W0, "cuda:0"), colmajor_to_torch(H0, "cuda:0"), divergence="kl", use_dist=False,
                   algorithm="lnmf" if algo == "lnmf" else "nmf")
        assert e.path_kind == 5 and e.cost_lag == 1
        e.close()
    if algo == "lnmf":
        cfg = dict(W_init=W0 / W0.sum(0), H_init=H0, maxiter=6, tolerance=1e-12)
        ref = O.lnmf(V, K, cfg)
        got = gpu_lib.lnmf(V, K, cfg)
        assert rel_fro(got[0], ref[0]) <= 1e-5 and rel_fro(got[1], ref[1]) <= 1e-5 and rel_fro(got[2], ref[2]) <= 1e-6
        return
    if algo == "src2":
        Ks = [120, 180]
        cfg = dict(divergence="kl", W_init=[W0[:, :120], W0[:, 120:]], H_init=[H0[:120], H0[120:]], W_sparsity=[0.05, 0.0], H_sparsity=[0.0, 0.1],
                   W_fixed=[False, True], maxiter=6, tolerance=1e-12)
        ref = O.nmf(V, Ks, cfg)
        got = gpu_lib.nmf(V, Ks, cfg)
        assert rel_fro(np.hstack(got[0]), np.hstack(ref[0])) <= 1e-5 and rel_fro(np.vstack(got[1]), np.vstack(ref[1])) <= 1e-5 and rel_fro(got[2], ref[2]) <= 1e-6
        return
    cfg = dict(divergence="kl", W_init=W0, H_init=H0, maxiter=8, tolerance=1e-12, W_sparsity=0.01, H_sparsity=0.02)
    ref = O.nmf(V, K, cfg)
    _check(gpu_lib.nmf(V, K, cfg), ref)
    _check(gpu_lib.nmf(V, K, dict(cfg, nmfx_gpus=[0, 0, 0])), ref)          # three column shards of one GPU
    probe = O.nmf(V, K, dict(cfg, maxiter=14, tolerance=1e-300))[2]
    dec = -np.diff(probe)
    if np.all(dec[:8] > 0) and dec[5] > dec[6]:
        cfg2 = dict(cfg, maxiter=14, tolerance=float(0.5 * (dec[5] + dec[6])))
        ref2 = O.nmf(V, K, cfg2)
        got2 = gpu_lib.nmf(V, K, cfg2)
        assert len(got2[2]) == len(ref2[2]) < 14
        _check(got2, ref2)


@pytest.mark.parametrize("m,n,K,planted", [(512, 768, 320, False), (300, 1000, 257, False), (640, 512, 512, True), (257, 4160, 288, True), (384, 1024, 800, False)])
def test_nmf_euclidean_K_above_256_in_column_blocks(gpu_lib, m, n, K, planted):
    """nmf.m:149-150,180-181 with K > 256: V*H' and W'*V block by block on the stationary kernel (the latter over the transposed copy of V), denominators from
    K x K Gram products, the cost in Gram form out of the W update's column sums -- engine path 6, cost lag 2.  Planted data drive the residual below 5 % of
    ||V||^2, where the device-side switch turns the explicit residual on: the S chain of functors 7 / 10.  Also: any K through the blocking call, the stop
    rule handing back the state of the iteration it fired on, column shards, and the two-operand GEMM path (nmfx_path = 1) as a second opinion."""
    from oracle import nmf_oracle as O
    from nmf_toolbox_amd.engine import Engine, colmajor_to_torch
    V, W0, H0 = synth(m, n, K, planted=planted)
    if K % 32 == 0:
        e = Engine(colmajor_to_torch(V, "cuda:0"), colmajor_to_torch(W0, "cuda:0"), colmajor_to_torch(H0, "cuda:0"), divergence="euclidean", use_dist=False)
        assert e.path_kind == 6 and e.cost_lag == 2
        e.close()
    cfg = dict(divergence="euclidean", W_init=W0, H_init=H0, maxiter=10, tolerance=1e-12, W_sparsity=0.01, H_sparsity=0.02)
    ref = O.nmf(V, K, cfg)
    got = gpu_lib.nmf(V, K, cfg)
    _check(got, ref)
    if planted:
        assert ref[2][-1] < 0.05 * 0.5 * float((V ** 2).sum())                # ... so the explicit residual pass did take over on the way
    _check(gpu_lib.nmf(V, K, dict(cfg, nmfx_gpus=[0, 0, 0])), ref)            # three column shards of one GPU
    _check(gpu_lib.nmf(V, K, dict(cfg, nmfx_path=1)), ref)
    again = gpu_lib.nmf(V, K, cfg)
    assert np.array_equal(again[0], got[0]) and np.array_equal(again[1], got[1]) and np.array_equal(again[2], got[2])   # run to run
    probe = O.nmf(V, K, dict(cfg, maxiter=14, tolerance=1e-300))[2]
    dec = -np.diff(probe)
    if np.all(dec[:8] > 0) and dec[5] > dec[6]:
        cfg2 = dict(cfg, maxiter=14, tolerance=float(0.5 * (dec[5] + dec[6])))
        ref2 = O.nmf(V, K, cfg2)
        got2 = gpu_lib.nmf(V, K, cfg2)
        assert len(got2[2]) == len(ref2[2]) < 14
        _check(got2, ref2)


# ---- IS and alpha-beta on the fused kernels (two element maps / two accumulator sets per pass, K <= 128): split and un-split epilogues,
# ragged shapes, padded K, sources with sparsity / fixed flags; against the oracle and against the generic (materialised V_hat) path ----
@pytest.mark.parametrize("div,ab", [("is", None), ("ab", (0.5, 1.5)), ("ab", (2.0, -0.5)), ("ab", (1.0, 0.5)), ("ab", (1.5, -1.5))])
@pytest.mark.parametrize("m,n,K,iters", [(256, 1024, 64, 15), (384, 640, 128, 10), (128, 8192, 32, 4), (513, 300, 40, 8), (129, 131, 96, 8), (2049, 257, 100, 4), (384, 1024, 192, 8), (300, 700, 150, 6),
                                         (384, 1024, 224, 6), (256, 768, 256, 6), (321, 515, 250, 5)])   # K above 128 (two accumulator sets still fit up to 192); above 192: two single-map passes
def test_nmf_fused_is_and_alpha_beta(gpu_lib, div, ab, m, n, K, iters):
    from oracle import nmf_oracle as O
    V, W0, H0 = synth(m, n, K)
    cfg = dict(divergence=div, W_init=W0, H_init=H0, maxiter=iters, tolerance=1e-12, W_sparsity=0.01, H_sparsity=0.02)
    if ab:
        cfg["alpha"], cfg["beta"] = ab
    ref = O.nmf(V, K, cfg)
    fused = gpu_lib.nmf(V, K, dict(cfg, nmfx_path=2))
    generic = gpu_lib.nmf(V, K, dict(cfg, nmfx_path=1))
    ctol = 1e-6 if np.all(np.isfinite(ref[2])) and abs(ref[2][-1]) > 1e-3 * float(V.sum()) else 1e-5   # IS / AB costs are small differences of large sums
    _check(fused, ref, cost_tol=ctol)
    _check(generic, ref, cost_tol=1e-5)


@pytest.mark.parametrize("ab", [(0.5, 1.0), (0.5, 0.5), (2.0, -1.0)])      # beta == 1: S.^(beta-1) = S.^0;  alpha + beta == 1: S.^(alpha+beta-1) = S.^0
def test_nmf_fused_alpha_beta_zero_exponent_on_zero_vhat(gpu_lib, ab):
    """x.^0 == 1 also where x == 0 (MATLAB).  A zero row of W makes a zero row of V_hat = W*H; with an exponent of exactly 0 the fused
    element map (exp2(e * log2(S))) must give 1 there, not 0 * (-Inf) = NaN -- one NaN would poison a whole column of the numerators."""
    from oracle import nmf_oracle as O
    V, W0, H0 = synth(256, 512, 64)
    W0 = W0.copy()
    W0[5, :] = 0.0
    cfg = dict(divergence="ab", alpha=ab[0], beta=ab[1], W_init=W0, H_init=H0, maxiter=4, tolerance=1e-12)
    with np.errstate(all="ignore"):
        ref = O.nmf(V, 64, cfg)
    fused = gpu_lib.nmf(V, 64, dict(cfg, nmfx_path=2))
    generic = gpu_lib.nmf(V, 64, dict(cfg, nmfx_path=1))
    for got in (fused, generic):
        assert np.all(np.isfinite(got[0])) == np.all(np.isfinite(ref[0])) and np.all(np.isfinite(got[1])) == np.all(np.isfinite(ref[1]))
        if np.all(np.isfinite(ref[0])) and np.all(np.isfinite(ref[1])):
            assert rel_fro(got[0], ref[0]) < 1e-5 and rel_fro(got[1], ref[1]) < 1e-5, (ab, rel_fro(got[0], ref[0]), rel_fro(got[1], ref[1]))


def test_nmf_fused_is_multi_source_fixed_and_shards(gpu_lib):
    from oracle import nmf_oracle as O
    V, W0, H0 = synth(256, 512, 64)
    Ks = [24, 40]
    cfg = dict(divergence="is", W_init=[W0[:, :24], W0[:, 24:]], H_init=[H0[:24], H0[24:]], W_sparsity=[0.05, 0.0], H_sparsity=[0.0, 0.1],
               W_fixed=[False, True], H_fixed=[False, False], maxiter=15, tolerance=1e-12, nmfx_path=2)
    ref = O.nmf(V, Ks, cfg)
    _check(gpu_lib.nmf(V, Ks, cfg), ref, cost_tol=1e-5)
    _check(gpu_lib.nmf(V, Ks, dict(cfg, nmfx_gpus=[0, 0, 0])), ref, cost_tol=1e-5)          # [N | P] through the peer exchange on three shards
    with pytest.raises(Exception, match="not eligible"):
        gpu_lib.nmf(V, 288, dict(divergence="is", maxiter=1, nmfx_path=2))                   # IS / alpha-beta on the fused kernels: K <= 256


@pytest.mark.parametrize("div,ab", [("is", None), ("ab", (0.5, 1.5))])
def test_nmf_is_ab_above_192_two_single_map_passes(gpu_lib, div, ab):
    """IS / alpha-beta with 192 < K <= 256 (nmf.m:154-164,185-195 have no K limit): the dual-map kernel's second accumulator set no longer fits, so every pass runs
    twice with one element map each (functors 11 + 12 / 13 + 14) -- V_hat still never formed.  The device-level engine says which path it took; sources with
    sparsity and a fixed factor, three column shards ([N | P] through the peer exchange), the stop rule, and run to run."""
    import torch
    from oracle import nmf_oracle as O
    from nmf_toolbox_amd.engine import Engine, colmajor_to_torch
    m, n, K = 384, 1280, 256
    V, W0, H0 = synth(m, n, K)
    kw = dict(alpha=ab[0], beta=ab[1]) if ab else {}
    e = Engine(colmajor_to_torch(V, "cuda:0"), colmajor_to_torch(W0, "cuda:0"), colmajor_to_torch(H0, "cuda:0"), divergence=div, use_dist=False, **kw)
    assert e.path_kind == 1 and e.cost_lag == 1           # the fused kernels (1), not the materialised path (0)
    e.close()
    Ks = [100, 156]
    cfg = dict(divergence=div, W_init=[W0[:, :100], W0[:, 100:]], H_init=[H0[:100], H0[100:]], W_sparsity=[0.02, 0.0], H_sparsity=[0.0, 0.05],
               W_fixed=[False, False], H_fixed=[True, False], maxiter=8, tolerance=1e-12, **kw)
    ref = O.nmf(V, Ks, cfg)
    got = gpu_lib.nmf(V, Ks, cfg)
    _check(got, ref, cost_tol=1e-5)
    sh = gpu_lib.nmf(V, Ks, dict(cfg, nmfx_gpus=[0, 0, 0]))
    _check(sh, ref, cost_tol=1e-5)
    again = gpu_lib.nmf(V, Ks, cfg)
    assert np.array_equal(np.hstack(again[0]), np.hstack(got[0])) and np.array_equal(np.vstack(again[1]), np.vstack(got[1])) and np.array_equal(again[2], got[2])   # run to run
    # ... and on the three shards (427 / 427 / 426 columns: the masked-edge instantiations, three engines sharing the device), eight times over.  The denominator
    # passes (functors 12 / 14) never look at V; while their V loads were merely unused the compiler dropped them and the tile-top `vmcnt(32)` returned before the
    # LDS-DMA rows had landed -- one run in four came out different (fused_kernel.h, NO_V)
    for _ in range(8):
        sh2 = gpu_lib.nmf(V, Ks, dict(cfg, nmfx_gpus=[0, 0, 0]))
        assert np.array_equal(np.hstack(sh2[0]), np.hstack(sh[0])) and np.array_equal(np.vstack(sh2[1]), np.vstack(sh[1])) and np.array_equal(sh2[2], sh[2])
    probe = O.nmf(V, K, dict(divergence=div, W_init=W0, H_init=H0, maxiter=12, tolerance=1e-300, **kw))[2]
    dec = -np.diff(probe)
    if np.all(np.isfinite(probe)) and np.all(dec[:8] > 0) and dec[4] > dec[5]:
        cfg2 = dict(divergence=div, W_init=W0, H_init=H0, maxiter=12, tolerance=float(0.5 * (dec[4] + dec[5])), **kw)
        ref2, got2 = O.nmf(V, K, cfg2), gpu_lib.nmf(V, K, cfg2)
        assert len(got2[2]) == len(ref2[2]) < 12
        _check(got2, ref2, cost_tol=1e-5)


# ---- cnmf on the register-stationary kernels (fused_kernel TT > 1): every instantiated (K, T) pair, aligned and ragged shapes, sparsity,
# fixed factors, 'frobenius' (no cost); against the oracle and against the GEMM formulations -----------------------------------------
@pytest.mark.parametrize("div", ["euclidean", "kl"])
@pytest.mark.parametrize("K,T", [(64, 8), (64, 4), (64, 2), (32, 4), (32, 8), (32, 16), (128, 2), (128, 4),
                                 (32, 3), (32, 5), (32, 6), (64, 3), (32, 10), (32, 12), (64, 5), (64, 6),           # from here: round 3
                                 (32, 7), (32, 9), (32, 11), (64, 7), (32, 13), (32, 14), (32, 15), (128, 3), (256, 2)])
@pytest.mark.parametrize("m,n", [(256, 512), (129, 333), (640, 65)])
def test_cnmf_fused_shift_sum_passes(gpu_lib, K, T, m, n, div):
    from oracle import nmf_oracle as O
    V, W0, H0 = synth(m, n, K, T=T)
    cfg = dict(divergence=div, W_init=W0, H_init=H0, maxiter=6, tolerance=1e-12, W_sparsity=0.01, H_sparsity=0.02)
    ref = O.cnmf(V, K, T, cfg)
    fused = gpu_lib.cnmf(V, K, T, dict(cfg, nmfx_path=2))          # 2 = the fused passes or an error
    _check(fused, ref)
    _check(gpu_lib.cnmf(V, K, T, dict(cfg, nmfx_path=1)), ref)     # materialised V_hat
    assert np.allclose(np.sqrt((fused[0] ** 2).sum((0, 2))), T, rtol=1e-5)


def test_cnmf_fused_fixed_factors_frobenius_and_refusals(gpu_lib):
    from oracle import nmf_oracle as O
    V, W0, H0 = synth(192, 400, 64, T=4)
    for extra in (dict(W_fixed=True), dict(H_fixed=True), dict(divergence="frobenius"), dict(divergence="kl", W_fixed=True), dict(divergence="kl", H_fixed=True)):
        cfg = dict(dict(divergence="euclidean", W_init=W0, H_init=H0, maxiter=5, tolerance=1e-12), **extra)
        _check(gpu_lib.cnmf(V, 64, 4, dict(cfg, nmfx_path=2)), O.cnmf(V, 64, 4, cfg))
    cfg = dict(W_init=[W0[:, :24], W0[:, 24:]], H_init=[H0[:24], H0[24:]], W_sparsity=[0.05, 0.0], H_fixed=[False, True], maxiter=5, tolerance=1e-12)
    _check(gpu_lib.cnmf(V, [24, 40], 4, dict(cfg, nmfx_path=2)), O.cnmf(V, [24, 40], 4, cfg))
    cfg = dict(divergence="kl", W_init=W0, H_init=H0, maxiter=200, tolerance=2.0, nmfx_path=2)     # KL: the cost lags one pass; the stop rule must
    got, ref = gpu_lib.cnmf(V, 64, 4, cfg), O.cnmf(V, 64, 4, cfg)                                   # return the state of the iteration it fired on
    assert len(ref[2]) < 200
    _check_stop(got[2], ref[2], 2.0)
    if len(got[2]) == len(ref[2]):
        assert rel_fro(got[0], ref[0]) <= TOL and rel_fro(got[1], ref[1]) <= TOL
    # (round 5 refused IS / alpha-beta by name outside eight (K, T) pairs; since round 6 every pair has the S pass: test_cnmf_is_and_alpha_beta_on_the_fused_passes_every_pair)
    rs = np.random.RandomState(7)
    Wi, Hi = np.fmax(rs.rand(V.shape[0], 32, 3), 1e-3), np.fmax(rs.rand(32, 300), 1e-3)
    cfg_is = dict(divergence="is", W_init=Wi, H_init=Hi, maxiter=2, tolerance=1e-12)
    _check(gpu_lib.cnmf(V[:, :300], 32, 3, dict(cfg_is, nmfx_path=2)), O.cnmf(V[:, :300], 32, 3, cfg_is))
    with pytest.raises(Exception, match="not eligible"):
        gpu_lib.cnmf(V[:, :300], 100, 7, dict(maxiter=1, nmfx_path=2))                                  # no pair with T = 7 reaches K = 100, padded or not
