"""tadataka.camera.parameters (reference tadataka/camera/parameters.py:4-35)."""
import numpy as np


class CameraParameters(object):
    """Pinhole intrinsics: focal_length (fx, fy), offset (ox, oy)."""

    def __init__(self, focal_length, offset):
        assert(len(focal_length) == 2)
        assert(len(offset) == 2)
        self.focal_length = np.array(list(focal_length), dtype=np.float64)
        self.offset = np.array(list(offset), dtype=np.float64)

    @property
    def matrix(self):
        K = np.identity(3)
        K[0, 0], K[1, 1] = self.focal_length
        K[0, 2], K[1, 2] = self.offset
        return K

    @property
    def params(self):
        return list(self.focal_length) + list(self.offset)

    @staticmethod
    def from_params(params):
        return CameraParameters(focal_length=params[0:2], offset=params[2:4])

    def __eq__(self, another):
        return (np.array_equal(self.focal_length, another.focal_length) and
                np.array_equal(self.offset, another.offset))
