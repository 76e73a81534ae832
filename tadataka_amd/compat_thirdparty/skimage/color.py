import numpy as np

from tadataka_amd import ops


def rgb2gray(rgb):
    """Luminance 0.2125 R + 0.7154 G + 0.0721 B of an [H, W, 3 | 4] image (uint8 input is scaled to
    [0, 1] first, as img_as_float does); 2-D input is returned unchanged, as scikit-image 0.16.2 does."""
    a = np.asarray(rgb)
    if a.ndim == 2:
        return a
    return ops.rgb2gray(a)
