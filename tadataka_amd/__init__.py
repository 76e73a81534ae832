"""tadataka_amd -- MI355X-native (gfx950) implementation of Tadataka's per-pixel
DVO / semi-dense / bundle-adjustment hot path.

    tadataka_amd.ops       NumPy-facing wrappers over the C ABI (libtadataka_hip.so)
    tadataka_amd.compat    drop-in `tadataka` and `rust_bindings` packages that keep
                           the reference's Python API and run on the HIP kernels

Importing this package puts `tadataka_amd/compat` on sys.path, so that
`import tadataka` / `import rust_bindings` resolve to the MI355X build.  The
three-function stand-in for scikit-image that the reference's examples need
(`tadataka_amd/compat_thirdparty`) is made importable only if no scikit-image is
installed (it is appended BEHIND everything else on sys.path, with a warning) or on
request: `tadataka_amd.install(thirdparty=True)`.
"""
import os
import sys

COMPAT_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "compat")
THIRDPARTY_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "compat_thirdparty")

# The pyramid PoseChangeEstimator builds (tadataka/vo/dvo/__init__.py:144-148):
# skimage.transform.rescale anti-aliases by default when it shrinks (scikit-image
# >= 0.15; setup.py:117 pins 0.16.2), so the Gaussian-prefiltered pyramid is the
# reference-equivalent one.  ONE constant: the drop-in tadataka.vo.dvo, bench.py
# and the tests all derive their default from it.
PYRAMID_ANTI_ALIASING = True


def install(thirdparty=None):
    """Makes `tadataka` and `rust_bindings` importable (idempotent).  thirdparty: True = also the
    scikit-image stand-in, False = never, None = only when scikit-image is not installed."""
    if COMPAT_DIR not in sys.path:
        sys.path.insert(0, COMPAT_DIR)
    if thirdparty is None:
        import importlib.util
        try:
            missing = importlib.util.find_spec("skimage") is None
        except (ImportError, ValueError):
            missing = True
        thirdparty = missing
        if missing and THIRDPARTY_DIR not in sys.path:
            import warnings
            warnings.warn("scikit-image is not installed: `import skimage` resolves to tadataka_amd's "
                          "three-function stand-in (rgb2gray, rescale, resize on the GPU)", ImportWarning)
    if thirdparty and THIRDPARTY_DIR not in sys.path:
        sys.path.append(THIRDPARTY_DIR)


install()


def enable_device_maps(enabled=True):
    """rust_bindings.semi_dense.increment_age / propagate / update_depth (and Frame.image) return plain ndarrays, as
    the reference's do -- one download per returned map.  enable_device_maps() makes them return
    tadataka_amd.ops.DeviceMap instead: array-likes that stay in HBM and are downloaded when somebody looks at them,
    so that the loop of examples/semi_dense_vo.py:182-199, which hands every map straight into the next call, moves
    nothing over PCIe.  A DeviceMap is not an ndarray subclass (isinstance(m, np.ndarray) is False): opt in where
    that is acceptable.  Returns the previous setting."""
    import rust_bindings.semi_dense as sd
    previous = sd.LAZY_MAPS
    sd.LAZY_MAPS = bool(enabled)
    return previous
