import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

EPS = 2.0 ** -52


_THREAD_LIMIT = []


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    wi = getattr(config, "workerinput", None)
    if wi is not None:
        # an xdist worker shares the host with the others: the oracle's BLAS pool gets its share of the cores (128-thread pools of four workers spin against each
        # other and against the single-threaded element-wise phases: the full-size oracle tests ran 3.5x slower that way, profiles/r5_09_gputests.log)
        n = max(1, (os.cpu_count() or 4) // max(1, int(wi.get("workercount", 1))))
        try:
            from threadpoolctl import threadpool_limits
            _THREAD_LIMIT.append(threadpool_limits(limits=n))
        except Exception:
            pass
        try:
            import torch
            torch.set_num_threads(n)
        except Exception:
            pass


_FIRST = ("test_gpu_golden.py", "test_gpu_fullsize_oracle.py", "test_gpu_pins.py")
# the BASELINE-sized oracle comparisons, longest first (seconds on the GPU box's host cores, profiles/r5_10_gputests.log)
_LONGEST = ("test_c3_full_kl", "test_c3_shard_kl", "test_c3_stop_rule_near_convergence", "test_c5_full_nmfsc", "test_c2_full_euclidean", "test_c4_full_cnmf[kl]",
            "test_default_100_iterations[kl]", "test_c4_full_cnmf[euclidean]", "test_default_100_iterations[euclidean]")


def pytest_collection_modifyitems(config, items):
    """The golden fixtures, the BASELINE-sized oracle comparisons and the pinned values first: a run that is cut short (-x, a time limit) has then already covered
    C1-C5.  The long oracle comparisons lead, longest first and each followed by one short golden test: xdist's load scheduler (two consecutive tests per worker to
    begin with, then one at a time: maxschedchunk = 1) then starts them on different workers at once and fills the gaps with the short tests."""
    def key(it):
        name = os.path.basename(str(it.fspath))
        return _FIRST.index(name) if name in _FIRST else len(_FIRST)
    items.sort(key=key)          # stable: the order inside a module and among the other modules is unchanged
    long_ = {it.name: it for it in items if os.path.basename(str(it.fspath)) == "test_gpu_fullsize_oracle.py" and it.name in _LONGEST}
    if not long_:
        return
    rest = [it for it in items if long_.get(it.name) is not it]
    lead = []
    for name in _LONGEST:
        if name in long_:
            lead.append(long_[name])
            if rest and os.path.basename(str(rest[0].fspath)) == "test_gpu_golden.py":
                lead.append(rest.pop(0))
    items[:] = lead + rest


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """NMFX_TEST_WORKERS=<n> spreads a session over pytest-xdist workers (every worker its own process and HIP context on the one GPU).  Opt-in only: with the
    full-size oracle comparisons served from fixtures (tests/golden/fullsize_*.npz) the suite is GPU-bound, and four workers sharing the host made the float64
    oracle tests slower than they were in sequence (profiles/r5_13_gputests_xdist4.log: 730 s against 663 s).  Runs before xdist's own hook of the same name."""
    if hasattr(config, "workerinput") or not config.pluginmanager.hasplugin("xdist") or getattr(config.option, "numprocesses", None) is not None:
        return
    nw = os.environ.get("NMFX_TEST_WORKERS")
    if nw and int(nw) > 1:
        config.option.numprocesses = int(nw)
        if getattr(config.option, "maxschedchunk", None) is None:
            config.option.maxschedchunk = 1        # one test at a time after the first two: the long tests do not queue up behind each other on one worker


# ---- worst relative errors per test, written to gpurun_out/parity_errors.json at the end of a -m gpu session -----------------
_ERRORS = {}
_CURRENT = [None]


@pytest.fixture(autouse=True)
def _track_current_test(request):
    _CURRENT[0] = request.node.nodeid
    yield
    _CURRENT[0] = None


def record_err(**vals):
    """remember max(value) per key for the running test (keys: W, H, WH, cost, ...); also returned for printing"""
    d = _ERRORS.setdefault(_CURRENT[0] or "?", {})
    for k, v in vals.items():
        v = float(v)
        if not (d.get(k, -1.0) >= v):      # NaN sticks
            d[k] = v
    return vals


def pytest_sessionfinish(session, exitstatus):
    import glob
    import json
    out = os.path.join(ROOT, "gpurun_out")
    wid = getattr(session.config, "workerinput", {}).get("workerid")
    try:
        if wid is not None:        # an xdist worker: its share goes to a file of its own, the controller merges (its sessionfinish runs after the workers')
            if _ERRORS:
                os.makedirs(out, exist_ok=True)
                with open(os.path.join(out, ".parity_errors.%s.json" % wid), "w") as f:
                    json.dump(_ERRORS, f)
            return
        tests = dict(_ERRORS)
        for fn in glob.glob(os.path.join(out, ".parity_errors.*.json")):
            with open(fn) as f:
                tests.update(json.load(f))
            os.remove(fn)
        if not tests:
            return
        os.makedirs(out, exist_ok=True)
        worst = {}
        for t, d in tests.items():
            for k, v in d.items():
                if k not in worst or not (worst[k][0] >= v):
                    worst[k] = (v, t)
        with open(os.path.join(out, "parity_errors.json"), "w") as f:
            json.dump(dict(contract=dict(W=1e-5, H=1e-5, WH=1e-5, cost=1e-6), worst={k: dict(value=v, test=t) for k, (v, t) in worst.items()},
                           tests=tests), f, indent=1, sort_keys=True)
    except OSError:
        pass


def rel_fro(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def synth(m, n, K, T=None, seed_v=1000, planted=False):
    """SURVEY.md 8(d) synthetic inputs: V = max(U(0,1), eps) (seed 1000+b), W_init seed 1, H_init seed 2."""
    rs = np.random.RandomState
    V = np.fmax(rs(seed_v).rand(m, n), EPS)
    if planted:
        V = rs(seed_v + 1).rand(m, K) @ rs(seed_v + 2).rand(K, n) / K + 0.01 * V
    W0 = np.fmax(rs(1).rand(m, K) if T is None else rs(1).rand(m, K, T), EPS)
    H0 = np.fmax(rs(2).rand(K, n), EPS)
    return V, W0, H0


SKETCH_R, SKETCH_SEED = 64, 20261001


def fullsize_sketch(W, H):
    """What tests/golden/make_fullsize_golden.py keeps of a full-size factorisation and what the tests form of the HIP path's result (the same function on both
    sides): norms, Gaussian sketches Om_W*W, H*Om_H, Om_W*V_hat*Om_H (Om from RandomState(SKETCH_SEED), r = SKETCH_R) and exact strided rows of W / columns of H."""
    W = np.asarray(W, dtype=np.float64)
    H = np.asarray(H, dtype=np.float64)
    m, n, K = W.shape[0], H.shape[1], H.shape[0]
    rs = np.random.RandomState(SKETCH_SEED)
    OmW, OmH = rs.standard_normal((SKETCH_R, m)), rs.standard_normal((n, SKETCH_R))
    Wf = W.reshape(m, -1, order="F")                       # (m, K) or (m, K*T) with slice t in columns t*K .. t*K+K-1
    SW, SH = OmW @ Wf, H @ OmH
    if W.ndim == 2:
        SWH = SW @ SH
    else:                                                  # V_hat = sum_t W_t * rshift_t(H)  (RFD.m:36-38):  Om_W*V_hat*Om_H = sum_t (Om_W*W_t) * (rshift_t(H)*Om_H)
        SWH = np.zeros((SKETCH_R, SKETCH_R))
        for t in range(W.shape[2]):
            SWH += SW[:, t * K:(t + 1) * K] @ (H[:, :n - t] @ OmH[t:, :])
    rows, cols = np.arange(0, m, max(1, m // 64)), np.arange(0, n, max(1, n // 64))
    return dict(W_fro=np.linalg.norm(Wf), H_fro=np.linalg.norm(H), SW=SW, SH=SH, SWH=SWH, W_rows=Wf[rows], H_cols=H[:, cols], rows=rows, cols=cols)


def fullsize_errors(got, name):
    """relative errors of the HIP path's (W, H, cost) against the oracle fixture tests/golden/fullsize_<name>.npz -> dict(W, H, WH, cost, W_rows, H_cols) + the fixture"""
    fx = np.load(os.path.join(ROOT, "tests", "golden", "fullsize_" + name + ".npz"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_fullsize_golden as G
    have = str(fx["stamp"]) if "stamp" in fx.files else "(none)"
    if have != G.stamp(name):      # oracle, synth, sketch or the case's configuration changed since the fixture was made
        pytest.fail("tests/golden/fullsize_%s.npz is stale (stamp %s, tree %s): regenerate it with tests/golden/make_fullsize_golden.py on a box with the memory" % (name, have, G.stamp(name)))
    W, H, c = got
    sk = fullsize_sketch(W, H)
    nrm = np.linalg.norm
    e = dict(W=nrm(sk["SW"] - fx["SW"]) / nrm(fx["SW"]), H=nrm(sk["SH"] - fx["SH"]) / nrm(fx["SH"]), WH=nrm(sk["SWH"] - fx["SWH"]) / nrm(fx["SWH"]),
             W_rows=nrm(sk["W_rows"] - fx["W_rows"]) / nrm(fx["W_rows"]), H_cols=nrm(sk["H_cols"] - fx["H_cols"]) / nrm(fx["H_cols"]),
             W_fro=abs(sk["W_fro"] - fx["W_fro"]) / fx["W_fro"], H_fro=abs(sk["H_fro"] - fx["H_fro"]) / fx["H_fro"],
             cost=(rel_fro(c, fx["cost"]) if len(c) == len(fx["cost"]) else float("inf")))
    return {k: float(v) for k, v in e.items()}, fx


@pytest.fixture(scope="session")
def gpu_lib():
    import nmf_toolbox_amd as A
    if A.device_count() < 1:
        pytest.fail("no MI355X visible: the gpu-marked tests must run on the GPU box")
    return A
