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
import pins_sc
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


# ---- nmfsc / cnmfsc / lnmf / constrainednmf: pins of tests/pins_sc.py (hand-derived cases, exact rationals, 50-digit decimals with the
# closed form of the Hoyer projection) and their mutation checks ------------------------------------------------------------------
SC_TOL, SC_CTOL = 1e-12, 1e-12


def test_sc_transcriptions_reproduce_the_hand_derivation():
    pins_sc.selfcheck()


def _nmfsc_pins(mod):
    pins_sc.pin_nmfsc_underflow(mod, SC_TOL, SC_CTOL)
    pins_sc.pin_nmfsc_mu(mod, SC_TOL, SC_CTOL)
    pins_sc.pin_nmfsc_linesearch(mod, SC_TOL, SC_CTOL)


def test_oracle_passes_nmfsc_pins():
    _nmfsc_pins(O)


def test_oracle_passes_cnmfsc_pins():
    pins_sc.pin_cnmfsc(O, SC_TOL, SC_CTOL)


def test_oracle_passes_lnmf_pins():
    pins_sc.pin_lnmf(O, SC_TOL, SC_CTOL)


def test_oracle_passes_constrainednmf_pins():
    pins_sc.pin_constrainednmf(O, SC_TOL, SC_CTOL)


def _mutant_in(func, old, new, count=1):
    """mutate inside ONE top-level function of the oracle (nmfsc and cnmfsc share many lines verbatim)"""
    a = _SRC.index("\ndef %s(" % func)
    b = _SRC.find("\ndef ", a + 1)
    b = len(_SRC) if b < 0 else b
    body = _SRC[a:b]
    assert body.count(old) >= 1, "mutation target vanished from oracle.%s: %s" % (func, old)
    mod = types.ModuleType("nmf_oracle_mutant")
    mod.__file__ = O.__file__
    exec(compile(_SRC[:a] + body.replace(old, new, count) + _SRC[b:], "nmf_oracle_mutant", "exec"), mod.__dict__)
    return mod


_CAUGHT = (AssertionError, ValueError, FloatingPointError, IndexError, ZeroDivisionError)

NMFSC_MUTATIONS = [
    ("nmfsc.m:62 V not divided by max(V(:))", "V = V / V.max()                                       # nmfsc.m:62", "pass", 1),
    ("nmfsc.m:106 L1s = sqrt(n) - sqrt(n)*sparsity (the -1 forgotten)", "L1s = np.sqrt(n) - (np.sqrt(n) - 1) * sH", "L1s = np.sqrt(n) - np.sqrt(n) * sH", 1),
    ("nmfsc.m:107-109 H not projected at init", "H[k, :] = projfunc(H[k, :], L1s, 1.0, True)[0]", "pass", 1),
    ("nmfsc.m:148 gradient sign", "dH = pos - neg                            # nmfsc.m:148", "dH = neg - pos", 1),
    ("nmfsc.m:169 step divided by 4", "stepsizeH = stepsizeH / 2             # nmfsc.m:169", "stepsizeH = stepsizeH / 4", 1),
    ("nmfsc.m:178 no 1.2x growth after an accepted step", "stepsizeH = 1.2 * stepsizeH               # nmfsc.m:178", "pass", 1),
    ("nmfsc.m:170-174 / 221-225 early return without trimming cost", "return _finish(it, True)", "return _finish(maxiter + 1, True)", 2),
    ("nmfsc.m:187 W not rescaled by the row norms of H", "W = W * norms[None, :]                    # nmfsc.m:187", "pass", 1),
    ("nmfsc.m:186 H not normalised", "H = (1.0 / norms)[:, None] * H            # nmfsc.m:186", "pass", 1),
    ("nmfsc.m:197 begobj = cost(iter) instead of the recomputed objective", "begobj = 0.5 * np.sum((V - V_hat) ** 2)   # nmfsc.m:197", "begobj = cost[it - 1]", 1),
    ("nmfsc.m:232 W MU followed by a column normalisation (as nmf.m:169 has)", "W = W * (neg / np.fmax(pos, EPS))     # nmfsc.m:232", "W = _col_normalize(W * (neg / np.fmax(pos, EPS)))", 1),
    ("nmfsc.m:228 step growth applied to the wrong step size", "stepsizeW = 1.2 * stepsizeW               # nmfsc.m:228", "stepsizeH = 1.2 * stepsizeH", 1),
]


@pytest.mark.parametrize("what,old,new,count", NMFSC_MUTATIONS, ids=[m[0] for m in NMFSC_MUTATIONS])
def test_nmfsc_pins_reject_mutant(what, old, new, count):
    mod = _mutant_in("nmfsc", old, new, count)
    with pytest.raises(_CAUGHT):
        with np.errstate(all="ignore"):
            _nmfsc_pins(mod)


CNMFSC_MUTATIONS = [
    ("cnmfsc.m:202 max(pos, eps) instead of pos + eps", "H = H * (neg / (pos + EPS))           # cnmfsc.m:202", "H = H * (neg / np.fmax(pos, EPS))", 1),
    ("cnmfsc.m:235 line search on the full 3-D reconstruction instead of the shift-less Wnew*H",
     "V_hat = reconstruct_from_decomposition(Wnew, H)   # cnmfsc.m:235: 2-D slice => plain Wnew*H",
     "Wtmp = W0.copy(); Wtmp[:, :, t - 1] = Wnew; V_hat = rfd3(Wtmp, H)", 1),
    ("cnmfsc.m:105-109 init projection applied to W0 as well", "W[:, k, t] = projfunc(W[:, k, t], L1a, 1.0, True)[0]", "W[:, k, t] = W0[:, k, t] = projfunc(W[:, k, t], L1a, 1.0, True)[0]", 1),
    ("cnmfsc.m:262 V_hat not updated between the slices of the MU W step", "V_hat = np.fmax(V_hat + np.ascontiguousarray(W[:, :, t - 1] - W0[:, :, t - 1]) @ Hsh, 0.0)", "pass", 1),
    ("cnmfsc.m:207-209 W0 not rescaled by the row norms of H", "W0[:, :, t] = W0[:, :, t] * norms[None, :]", "pass", 1),
    ("cnmfsc.m:245-249 early return without trimming cost", "return _finish(it, True)", "return _finish(maxiter + 1, True)", 2),
    ("cnmfsc.m:266 W0 = W dropped", "W0 = W.copy()                                     # cnmfsc.m:266 (value semantics)", "pass", 1),
    ("cnmfsc.m:221 shift direction in the W step", "Hsh = _rshift(H, t, n)                # cnmfsc.m:221", "Hsh = _lshift(H, t, n)", 1),
]


@pytest.mark.parametrize("what,old,new,count", CNMFSC_MUTATIONS, ids=[m[0] for m in CNMFSC_MUTATIONS])
def test_cnmfsc_pins_reject_mutant(what, old, new, count):
    mod = _mutant_in("cnmfsc", old, new, count)
    with pytest.raises(_CAUGHT):
        with np.errstate(all="ignore"):
            pins_sc.pin_cnmfsc(mod, SC_TOL, SC_CTOL)


LNMF_MUTATIONS = [
    ("lnmf.m:59 L2 instead of L1 column normalisation at init", "W = W * (1.0 / np.sum(W, axis=0))[None, :]            # lnmf.m:59", "W = _col_normalize(W)", 1),
    ("lnmf.m:69 numerator without the quotient", "W = W * (((V / V_hat) @ H.T) / np.fmax(ones_mn @ H.T, EPS))", "W = W * ((V @ H.T) / np.fmax(ones_mn @ H.T, EPS))", 1),
    ("lnmf.m:70 normalisation after the update dropped", "W = W * (1.0 / np.sum(W, axis=0))[None, :]\n                V_hat = W @ H", "V_hat = W @ H", 1),
    ("lnmf.m:71 V_hat not refreshed before the H update", "W = W * (1.0 / np.sum(W, axis=0))[None, :]\n                V_hat = W @ H", "W = W * (1.0 / np.sum(W, axis=0))[None, :]", 1),
    ("lnmf.m:76 sqrt dropped (nmf's H update)", "H = np.sqrt(H * (W.T @ (V / V_hat)))", "H = H * (W.T @ (V / V_hat))", 1),
    ("lnmf.m:84 strict < as in nmf.m:221", "if it > 1 and cost[it - 1] <= cost[it - 2] and cost[it - 2] - cost[it - 1] <= tol:", "if it > 1 and cost[it - 1] < cost[it - 2] and cost[it - 2] - cost[it - 1] < tol:", 1),
    ("lnmf.m:85 cost trimmed on break", "return W, H, cost                                     # cost is NOT trimmed", "return W, H, cost[:it]   # mutant: trimmed", 1),
]


@pytest.mark.parametrize("what,old,new,count", LNMF_MUTATIONS, ids=[m[0] for m in LNMF_MUTATIONS])
def test_lnmf_pins_reject_mutant(what, old, new, count):
    mod = _mutant_in("lnmf", old, new, count)
    with pytest.raises(_CAUGHT):
        with np.errstate(all="ignore"):
            pins_sc.pin_lnmf(mod, SC_TOL, SC_CTOL)


CONSTRAINED_MUTATIONS = [
    ("constrainednmf.m:251 Z_sparsity charged on H = Z*A instead of Z", "lamZ * np.sum(np.abs(Z))", "lamZ * np.sum(np.abs(H))", 1),
    ("constrainednmf.m:170 unlabelled samples last (as in the paper) instead of first",
     "A = np.block([[np.eye(n_u), np.zeros((n_u, num_labeled))],\n                  [np.zeros((num_classes, n_u)), Cm]])",
     "A = np.block([[np.zeros((num_classes, n_u)), Cm],\n                  [np.eye(n_u), np.zeros((n_u, num_labeled))]])", 1),
    ("constrainednmf.m:163 descending sort", "sorted_idx = np.argsort(processed, kind=\"stable\")", "sorted_idx = np.argsort(-processed, kind=\"stable\")", 1),
    ("constrainednmf.m:235 Z_sparsity ignored in the update", "Z = Z * (neg / np.fmax(pos + lamZ, EPS))", "Z = Z * (neg / np.fmax(pos, EPS))", 1),
    ("constrainednmf.m:263-267 H / A left in label-sorted order", "A[:, sorted_idx[samp]] = A_temp[:, samp]", "pass", 1),
    ("constrainednmf.m:219 KL denominator from V_hat", "pos = W.T @ ones_mn @ A.T", "pos = W.T @ V_hat @ A.T", 1),
    ("constrainednmf.m:215 numerator without the class sum (A' dropped, H-shaped update)", "neg = W.T @ V @ A.T\n                    pos = W.T @ V_hat @ A.T", "neg = (W.T @ V)[:, : Z.shape[1]]\n                    pos = (W.T @ V_hat)[:, : Z.shape[1]]", 1),
]


@pytest.mark.parametrize("what,old,new,count", CONSTRAINED_MUTATIONS, ids=[m[0] for m in CONSTRAINED_MUTATIONS])
def test_constrainednmf_pins_reject_mutant(what, old, new, count):
    mod = _mutant_in("constrainednmf", old, new, count)
    with pytest.raises(_CAUGHT):
        with np.errstate(all="ignore"):
            pins_sc.pin_constrainednmf(mod, SC_TOL, SC_CTOL)
