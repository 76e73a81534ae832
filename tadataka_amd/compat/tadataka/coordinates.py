"""tadataka.coordinates (reference tadataka/coordinates.py:7-40): integer pixel
grids and fancy indexing helpers (host-side index bookkeeping)."""
import numpy as np


def image_coordinates(image_shape):
    """All pixel coordinates (x, y) of an image in raster order, x fastest:
    [[0,0],[1,0],...,[W-1,0],[0,1],...].  int64 [H*W, 2]."""
    height, width = image_shape[0:2]
    ys, xs = np.divmod(np.arange(height * width, dtype=np.int64), width)
    return np.stack([xs, ys], axis=1)


def yx_to_xy(coordinates):
    return coordinates[:, ::-1]


def xy_to_yx(coordinates):
    return coordinates[:, ::-1]


def substitute(array2d, us, values):
    assert(us.shape[0] == values.shape[0])
    array2d[us[:, 1], us[:, 0]] = values
    return array2d


def get(array2d, us):
    return array2d[us[:, 1], us[:, 0]]
