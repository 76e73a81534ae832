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
    config.addinivalue_line("markers", "skimage_pyramid: the drop-in PoseChangeEstimator keeps its default pyramid "
                                       "(skimage to the bit); every other GPU test gets the ideal-constants one")


def fuzz_settings():
    """(cases per family, seed) of tests/test_gpu_fuzz.py.  TDK_FUZZ_N / TDK_FUZZ_SEED win; by default the regular
    `pytest -m gpu` run IS a soak of its own: 2 000 cases per family (about a minute on the GPU box) on a seed that
    rotates with the code -- the first bytes of the SHA-256 of the built library (the snapshot on the GPU box has no
    .git to take a commit hash from; the library changes whenever a kernel does).  Both are printed in the header
    and in the last lines of the run, and every failure message carries the case number."""
    n = int(os.environ.get("TDK_FUZZ_N", "2000"))
    if "TDK_FUZZ_SEED" in os.environ:
        return n, int(os.environ["TDK_FUZZ_SEED"])
    import hashlib
    lib = os.path.join(REPO, "tadataka_amd", "lib", "libtadataka_hip.so")
    try:
        with open(lib, "rb") as f:
            digest = hashlib.sha256(f.read()).digest()
        return n, 1000 + int.from_bytes(digest[:4], "big") % 9000
    except OSError:
        return n, 0


def pytest_report_header(config):
    n, seed = fuzz_settings()
    return f"fuzz soak: TDK_FUZZ_N={n} cases per family, TDK_FUZZ_SEED={seed}"


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    ran = [r for r in terminalreporter.stats.get("passed", []) + terminalreporter.stats.get("failed", [])
           if "test_gpu_fuzz.py" in r.nodeid]
    if ran:
        n, seed = fuzz_settings()
        secs = sum(getattr(r, "duration", 0.0) for r in ran)
        terminalreporter.write_line(f"fuzz soak: {len(ran)} families x TDK_FUZZ_N={n} cases, TDK_FUZZ_SEED={seed}, "
                                    f"{secs:.0f} s in tests/test_gpu_fuzz.py")


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


@pytest.fixture(autouse=True)
def _ideal_pyramid_for_the_round_1_to_4_fixtures(request):
    """The fixtures of rounds 1-4 (dvo_pyramid, dvo_vga_pyramid, dvo_examples, dvo_real, dvo_holes, dvo_ill) were
    generated with a stand-in rescale at the IDEAL sample positions, and the oracle's default loop is the same
    reading.  The drop-in PoseChangeEstimator now defaults to skimage's rescale to the bit (level 0 included), which
    differs from them by the 1e-13 that moves poses by ~1e-5 -- so the tests that compare with those fixtures select
    PYRAMID = "ideal"; the tests of the real-skimage fixtures (tests/test_gpu_round5.py) are marked
    skimage_pyramid and keep the default."""
    if "gpu" not in request.keywords or "skimage_pyramid" in request.keywords or not _has_gpu():
        yield
        return
    import tadataka_amd  # noqa: F401
    import tadataka.vo.dvo as dvo
    saved = dvo.PYRAMID
    dvo.PYRAMID = "ideal"
    try:
        yield
    finally:
        dvo.PYRAMID = saved


_CANARY = {"tests": 0, "max_live": 0}


@pytest.fixture(autouse=True)
def _canary_check_after_every_gpu_test(request):
    """TDK_DEBUG_CANARY=1 pytest -m gpu: after every GPU test the red zones around every live device allocation
    are verified (tdk_debug_check_canaries); a damaged one fails the test that left it behind."""
    yield
    if "gpu" not in request.keywords or not _has_gpu():
        return
    import ctypes
    from tadataka_amd import _lib
    for option in (0, 1):                       # library-wide options a test may have switched: back to the defaults
        _lib.load().tdk_set_option(option, 1)
    if os.environ.get("TDK_DEBUG_CANARY") != "1":
        return
    n = ctypes.c_int()
    st = _lib.load().tdk_debug_check_canaries(ctypes.byref(n))
    assert st == 0, _lib.load().tdk_last_error().decode()
    _CANARY["tests"] += 1
    _CANARY["max_live"] = max(_CANARY["max_live"], n.value)


def pytest_sessionfinish(session, exitstatus):
    if os.environ.get("TDK_DEBUG_CANARY") == "1" and _CANARY["tests"]:
        print(f"\ncanaries: verified after {_CANARY['tests']} GPU tests, up to {_CANARY['max_live']} live device "
              f"allocations between red zones at a time, exit status {int(exitstatus)}")


@pytest.fixture
def device_maps():
    """tadataka_amd.enable_device_maps() for one test: rust_bindings.semi_dense returns DeviceMaps (the default is
    plain ndarrays, as the reference returns)."""
    import tadataka_amd
    previous = tadataka_amd.enable_device_maps(True)
    try:
        yield
    finally:
        tadataka_amd.enable_device_maps(previous)


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
