"""rust_bindings.camera (src/py/camera.rs:6-41)."""
import numpy as np


class CameraParameters(object):
    """CameraParameters((fx, fy), (ox, oy)) with ndarray getters."""

    def __init__(self, focal_length, offset):
        fx, fy = focal_length
        ox, oy = offset
        self._focal_length = np.array([float(fx), float(fy)])
        self._offset = np.array([float(ox), float(oy)])

    @property
    def focal_length(self):
        return self._focal_length.copy()

    @property
    def offset(self):
        return self._offset.copy()
