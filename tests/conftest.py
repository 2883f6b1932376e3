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
