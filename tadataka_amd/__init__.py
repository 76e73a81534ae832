"""tadataka_amd -- MI355X-native (gfx950) implementation of Tadataka's per-pixel
DVO / semi-dense / bundle-adjustment hot path.

    tadataka_amd.ops       NumPy-facing wrappers over the C ABI (libtadataka_hip.so)
    tadataka_amd.compat    drop-in `tadataka` and `rust_bindings` packages that keep
                           the reference's Python API and run on the HIP kernels

Importing this package puts `tadataka_amd/compat` on sys.path, so that
`import tadataka` / `import rust_bindings` resolve to the MI355X build, and
appends `tadataka_amd/compat_thirdparty` (a three-function stand-in for
scikit-image) BEHIND everything else, so a real scikit-image always wins.
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


def install():
    """Makes `tadataka` and `rust_bindings` importable (idempotent)."""
    if COMPAT_DIR not in sys.path:
        sys.path.insert(0, COMPAT_DIR)
    if THIRDPARTY_DIR not in sys.path:
        sys.path.append(THIRDPARTY_DIR)


install()
