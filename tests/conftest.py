import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _has_gpu():
    return os.path.exists("/dev/kfd") and os.path.isdir("/sys/class/kfd/kfd/topology/nodes")


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU should skip, not crash
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


_IU6 = np.triu_indices(6)


def h21_err(H21, H21_ref):
    """Per-entry error of a 6x6 normal matrix given as its upper triangle (21):
    max_ij |H_ij - Href_ij| / sqrt(Href_ii Href_jj).  The entries span orders of
    magnitude (translation vs rotation blocks), so each one is measured against
    its own Cauchy-Schwarz scale instead of against the largest entry."""
    H21 = np.asarray(H21, dtype=np.float64).reshape(21)
    ref = np.asarray(H21_ref, dtype=np.float64).reshape(21)
    M = np.zeros((6, 6))
    M[_IU6] = ref
    d = np.sqrt(np.maximum(np.diag(M), 1e-300))
    scale = np.outer(d, d)[_IU6]
    return float(np.max(np.abs(H21 - ref) / scale))


def b6_err(b, b_ref, H21_ref):
    """Per-entry error of J^T W r: |b_i - bref_i| / (sqrt(Href_ii) rho) with
    rho = max_i |bref_i| / sqrt(Href_ii) (a lower bound of sqrt(sum w r^2))."""
    b = np.asarray(b, dtype=np.float64).reshape(6)
    ref = np.asarray(b_ref, dtype=np.float64).reshape(6)
    M = np.zeros((6, 6))
    M[_IU6] = np.asarray(H21_ref, dtype=np.float64).reshape(21)
    d = np.sqrt(np.maximum(np.diag(M), 1e-300))
    rho = max(float(np.max(np.abs(ref) / d)), 1e-300)
    return float(np.max(np.abs(b - ref) / (d * rho)))
