"""tadataka_amd -- MI355X-native (gfx950) implementation of Tadataka's per-pixel
DVO / semi-dense / bundle-adjustment hot path.

    tadataka_amd.ops       NumPy-facing wrappers over the C ABI (libtadataka_hip.so)
    tadataka_amd.compat    drop-in `tadataka` and `rust_bindings` packages that keep
                           the reference's Python API and run on the HIP kernels

Importing this package puts `tadataka_amd/compat` on sys.path, so that
`import tadataka` / `import rust_bindings` resolve to the MI355X build.
"""
import os
import sys

COMPAT_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "compat")


def install():
    """Makes `tadataka` and `rust_bindings` importable (idempotent)."""
    if COMPAT_DIR not in sys.path:
        sys.path.insert(0, COMPAT_DIR)


install()
