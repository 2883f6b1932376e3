"""CPU suite: the oracle against pins that depend on neither restatement (tests/pins.py: hand-derived exact rationals,
fixed points, closed forms), and a MUTATION check -- each pin set must reject deliberately broken copies of the oracle, one
cited line flipped at a time -- so a green run means the pins can actually see the lines they claim to cover.

Parity with the MATLAB toolbox remains unpinned BY THE REFERENCE (it ships no vectors and cannot run here); these tests
remove the common-mode risk of the two restatements sharing one misreading."""
import os
import types

import numpy as np
import pytest

import pins
from oracle import c_oracle as CO
from oracle import nmf_oracle as O

TOL, CTOL = 1e-13, 1e-12


def test_exact_helpers_reproduce_the_hand_derivation():
    pins.pin_kat1_selfcheck()


def test_oracle_passes_nmf_pins():
    pins.run_nmf_pins(O, TOL, CTOL, fp_tol=1e-13, fp_cost_rel=1e-25)


def test_oracle_passes_cnmf_fixed_point():
    pins.pin_cnmf_fixed_point(O, 1e-13, 1e-25)
    pins.pin_cnmf_kat(O, TOL, CTOL)


def test_oracle_passes_projfunc_closed_forms():
    pins.pin_projfunc(O, 1e-13)


class _CImpl:
    """the plain-C restatement behind the toolbox call surface (single source, scalar options)"""

    @staticmethod
    def nmf(V, K, cfg):
        if isinstance(K, (list, tuple)):
            Ks = list(K)
            W0, H0 = np.hstack(cfg["W_init"]), np.vstack(cfg["H_init"])
            rep = lambda name, d: np.concatenate([np.full(k, float(v)) for k, v in zip(Ks, cfg.get(name, [d] * len(Ks)))])
            W, H, c = CO.nmf(V, W0, H0, div=cfg["divergence"], lamW=rep("W_sparsity", 0), lamH=rep("H_sparsity", 0),
                             fixW=rep("W_fixed", 0), fixH=rep("H_fixed", 0), maxiter=cfg["maxiter"], tol=cfg.get("tolerance", 1e-3))
            cut = np.cumsum(Ks)[:-1]
            return np.split(W, cut, axis=1), np.split(H, cut, axis=0), c
        return CO.nmf(V, cfg["W_init"], cfg["H_init"], div=cfg["divergence"], lamW=cfg.get("W_sparsity", 0.0), lamH=cfg.get("H_sparsity", 0.0),
                      fixW=int(bool(cfg.get("W_fixed", False))), fixH=int(bool(cfg.get("H_fixed", False))), maxiter=cfg["maxiter"],
                      tol=cfg.get("tolerance", 1e-3))

    @staticmethod
    def cnmf(V, K, T, cfg):
        return CO.cnmf(V, cfg["W_init"], cfg["H_init"], div=cfg["divergence"], lamW=cfg.get("W_sparsity", 0.0), lamH=cfg.get("H_sparsity", 0.0),
                       fixW=int(bool(cfg.get("W_fixed", False))), fixH=int(bool(cfg.get("H_fixed", False))), maxiter=cfg["maxiter"],
                       tol=cfg.get("tolerance", 1e-3))


def test_c_oracle_passes_the_same_pins():
    pins.run_nmf_pins(_CImpl, TOL, CTOL, fp_tol=1e-13, fp_cost_rel=1e-25)
    pins.pin_cnmf_fixed_point(_CImpl, 1e-13, 1e-25)
    pins.pin_cnmf_kat(_CImpl, TOL, CTOL)


# ---- mutation check ---------------------------------------------------------------------------------------------------
_SRC = open(os.path.join(os.path.dirname(O.__file__), "nmf_oracle.py")).read()


def _mutant(old, new, count=1):
    assert _SRC.count(old) >= 1, "mutation target vanished from the oracle: " + old
    mod = types.ModuleType("nmf_oracle_mutant")
    mod.__file__ = O.__file__
    exec(compile(_SRC.replace(old, new, count), "nmf_oracle_mutant", "exec"), mod.__dict__)
    return mod


# (what breaks, text of the cited oracle line, its replacement)
NMF_MUTATIONS = [
    ("nmf.m:149 diag term uses V instead of V_hat", "neg = V @ Hs.T + Ws * _ddiag(Hs @ V_hat.T @ Ws)[None, :]", "neg = V @ Hs.T + Ws * _ddiag(Hs @ V.T @ Ws)[None, :]"),
    ("nmf.m:150 diag term dropped", "pos = V_hat @ Hs.T + Ws * _ddiag(Hs @ V.T @ Ws)[None, :]", "pos = V_hat @ Hs.T"),
    ("nmf.m:149/150 diag terms swapped", "neg = V @ Hs.T + Ws * _ddiag(Hs @ V_hat.T @ Ws)[None, :]\n                    pos = V_hat @ Hs.T + Ws * _ddiag(Hs @ V.T @ Ws)[None, :]",
     "neg = V @ Hs.T + Ws * _ddiag(Hs @ V.T @ Ws)[None, :]\n                    pos = V_hat @ Hs.T + Ws * _ddiag(Hs @ V_hat.T @ Ws)[None, :]"),
    ("nmf.m:152 KL numerator without the quotient", "neg = (V / V_hat) @ Hs.T + Ws * _ddiag(Hs @ ones_nm @ Ws)[None, :]", "neg = V @ Hs.T + Ws * _ddiag(Hs @ ones_nm @ Ws)[None, :]"),
    ("nmf.m:153 KL diag term dropped", "pos = ones_mn @ Hs.T + Ws * _ddiag(Hs @ (V.T / V_hat.T) @ Ws)[None, :]", "pos = ones_mn @ Hs.T"),
    ("nmf.m:155 IS numerator V./V_hat instead of V./V_hat.^2", "neg = (V / V_hat ** 2) @ Hs.T +", "neg = (V / V_hat) @ Hs.T +"),
    ("nmf.m:168 sparsity added to the numerator side", "Ws = Ws * (neg / np.fmax(pos + cfg[\"W_sparsity\"][s], EPS))", "Ws = Ws * ((neg + cfg[\"W_sparsity\"][s]) / np.fmax(pos, EPS))"),
    ("nmf.m:169 normalisation dropped", "W[s] = _col_normalize(Ws)                 # nmf.m:169", "W[s] = Ws"),
    ("nmf.m:169 L1 instead of L2 normalisation", "return Ws * (1.0 / np.sqrt(np.sum(Ws ** 2, axis=0)))[None, :]", "return Ws * (1.0 / np.sum(np.abs(Ws), axis=0))[None, :]"),
    ("nmf.m:173 V_hat not refreshed before the H step", "V_hat = reconstruct_from_decomposition(W_all, H_all)   # nmf.m:173", "pass"),
    ("nmf.m:180/181 swapped", "neg = Ws.T @ V\n                    pos = Ws.T @ V_hat", "neg = Ws.T @ V_hat\n                    pos = Ws.T @ V"),
    ("nmf.m:184 KL H denominator from V_hat", "pos = Ws.T @ ones_mn", "pos = Ws.T @ V_hat"),
    ("nmf.m:187 IS H denominator", "pos = Ws.T @ (ones_mn / V_hat)", "pos = Ws.T @ ones_mn"),
    ("nmf.m:199 H sparsity ignored", "H[s] = Hs * (neg / np.fmax(pos + cfg[\"H_sparsity\"][s], EPS))", "H[s] = Hs * (neg / np.fmax(pos, EPS))"),
    ("nmf.m:208 cost without the 0.5", "return 0.5 * np.sum((V - V_hat) ** 2)", "return np.sum((V - V_hat) ** 2)"),
    ("nmf.m:210 KL cost without -V + V_hat", "return np.sum(V * np.log(V / V_hat) - V + V_hat)", "return np.sum(V * np.log(V / V_hat))"),
    ("nmf.m:212 IS cost sign", "return np.sum(np.log(V_hat / V) + (V / V_hat) - 1.0)", "return np.sum(np.log(V / V_hat) + (V / V_hat) - 1.0)"),
    ("nmf.m:217 sparsity cost term dropped", "c = c + cfg[\"W_sparsity\"][s] * np.sum(np.abs(W[s])) + cfg[\"H_sparsity\"][s] * np.sum(np.abs(H[s]))", "c = c"),
    ("nmf.m:130-134 init normalisation dropped", "W = [_col_normalize(w) for w in W]                    # nmf.m:130-134", "W = list(W)"),
]


@pytest.mark.parametrize("what,old,new", NMF_MUTATIONS, ids=[m[0] for m in NMF_MUTATIONS])
def test_nmf_pins_reject_mutant(what, old, new):
    mod = _mutant(old, new)
    with pytest.raises(AssertionError):
        pins.run_nmf_pins(mod, TOL, CTOL, fp_tol=1e-13, fp_cost_rel=1e-25)


CNMF_MUTATIONS = [
    ("cnmf.m:220-221 KL quirk 'fixed' (V_pos shifted like the others)", "Vp_sh = V_pos if is_kl else _lshift(V_pos, t, n)", "Vp_sh = _lshift(V_pos, t, n)"),
    ("cnmf.m:188 shift direction", "return np.concatenate([np.zeros((K, t - 1)), Hs[:, : n - t + 1]], axis=1)", "return np.concatenate([Hs[:, t - 1:], np.zeros((K, t - 1))], axis=1)"),
    ("cnmf.m:191 diag term dropped", "gneg = _pw(Vn @ Hsh.T + Wt * _ddiag(Hsh @ _pw(V_hat.T, alpha + beta - 1) @ Wt)[None, :], ex)\n                        gpos = _pw(Vp @ Hsh.T + Wt * _ddiag(Hsh @ Vn.T @ Wt)[None, :], ex)\n                    W[s]", "gneg = _pw(Vn @ Hsh.T, ex)\n                        gpos = _pw(Vp @ Hsh.T + Wt * _ddiag(Hsh @ Vn.T @ Wt)[None, :], ex)\n                    W[s]"),
    ("cnmf.m:196-199 slab renormalisation dropped", "W[s] = W[s] / w_norm[None, :, None]\n            W_all", "pass\n            W_all"),
    ("cnmf.m:166 init does not rescale H", "H[s] = w_norm[:, None] * H[s]", "pass"),
]


@pytest.mark.parametrize("what,old,new", CNMF_MUTATIONS, ids=[m[0] for m in CNMF_MUTATIONS])
def test_cnmf_pins_reject_mutant(what, old, new):
    mod = _mutant(old, new, count=99)
    with pytest.raises(AssertionError):
        pins.pin_cnmf_fixed_point(mod, 1e-13, 1e-25)
        pins.pin_cnmf_kat(mod, TOL, CTOL)


PROJ_MUTATIONS = [
    ("projfunc.m:22 initial shift dropped", "v = s + (k1 - s.sum()) / N", "v = s.copy()"),
    ("projfunc.m:37 other root of the quadratic", "alphap = (-b + sq) / (2.0 * a)", "alphap = (-b - sq) / (2.0 * a)"),
    ("projfunc.m:31 midpoint over all N", "midpoint = np.ones(N) * k1 / (N - zerocoeff.size)", "midpoint = np.ones(N) * k1 / N"),
    ("projfunc.m:49 strict negativity test", "zerocoeff = np.flatnonzero(v <= 0)", "zerocoeff = np.flatnonzero(v < -0.05)"),
    ("projfunc.m:52 redistribution over all N", "v = v + (k1 - tempsum) / (N - zerocoeff.size)", "v = v + (k1 - tempsum) / N"),
]


@pytest.mark.parametrize("what,old,new", PROJ_MUTATIONS, ids=[m[0] for m in PROJ_MUTATIONS])
def test_projfunc_pins_reject_mutant(what, old, new):
    global _SRC
    keep = _SRC
    try:     # a broken projection may never reach all(v >= 0): bound the loop, falling out of it (usediters unset) counts as caught
        _SRC = keep.replace("    while True:\n        midpoint", "    for _guard in range(500):\n        midpoint")
        assert _SRC != keep
        mod = _mutant(old, new)
    finally:
        _SRC = keep
    with pytest.raises((AssertionError, FloatingPointError, ValueError, ZeroDivisionError, UnboundLocalError)):
        with np.errstate(all="ignore"):
            pins.pin_projfunc(mod, 1e-13)
