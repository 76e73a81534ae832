"""tadataka.interpolation (reference tadataka/interpolation/__init__.py:8-29)."""
import numpy as np

from rust_bindings import interpolation as _interpolation


def interpolation_(image, C):
    C = np.ascontiguousarray(C, dtype=np.float64)
    image = np.ascontiguousarray(image, dtype=np.float64)
    if C.ndim == 1:
        return _interpolation.interpolation(image, C.reshape(1, 2))[0]
    if C.ndim != 2:
        raise ValueError("Argument number 1 has to be 1d or 2d array")
    return _interpolation.interpolation(image, C)


def interpolation(image, C):
    """Bilinear samples of a gray image at float coordinates (x, y).

    Raises ValueError for a non-2D image or any coordinate outside
    [0, W-1] x [0, H-1] (the range check runs on the device with the samples)."""
    if not np.ndim(image) == 2:
        raise ValueError("Image have to be a two dimensional array")
    return interpolation_(image, C)
