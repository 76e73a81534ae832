import numpy as np

from tadataka_amd import ops


def rescale(image, scale, anti_aliasing=True, **kwargs):
    """Bilinear (order=1) rescale of a 2-D image; anti_aliasing as in 0.15+ (Gaussian
    prefilter with sigma = (1/scale - 1) / 2 when shrinking)."""
    image = np.asarray(image, dtype=np.float64)
    if image.ndim != 2:
        raise NotImplementedError("only 2-D images are rescaled on the hot path")
    return ops.rescale(image, scale, anti_aliasing=bool(anti_aliasing) and scale < 1.0)


def resize(image, output_shape, anti_aliasing=True, **kwargs):
    image = np.asarray(image, dtype=np.float64)
    if image.ndim != 2:
        raise NotImplementedError("only 2-D images are resized on the hot path")
    return ops.resize(image, tuple(int(v) for v in output_shape[:2]),
                      anti_aliasing=bool(anti_aliasing) and
                      (output_shape[0] < image.shape[0] or output_shape[1] < image.shape[1]))
