import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

EPS = 2.0 ** -52


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
    if not _ERRORS:
        return
    import json
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        worst = {}
        for t, d in _ERRORS.items():
            for k, v in d.items():
                if k not in worst or not (worst[k][0] >= v):
                    worst[k] = (v, t)
        with open(os.path.join(out, "parity_errors.json"), "w") as f:
            json.dump(dict(contract=dict(W=1e-5, H=1e-5, WH=1e-5, cost=1e-6), worst={k: dict(value=v, test=t) for k, (v, t) in worst.items()},
                           tests=_ERRORS), f, indent=1, sort_keys=True)
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


@pytest.fixture(scope="session")
def gpu_lib():
    import nmf_toolbox_amd as A
    if A.device_count() < 1:
        pytest.fail("no MI355X visible: the gpu-marked tests must run on the GPU box")
    return A
