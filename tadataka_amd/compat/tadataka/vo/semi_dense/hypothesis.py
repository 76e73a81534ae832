"""tadataka.vo.semi_dense.hypothesis -- the module examples/semi_dense_vo.py:14
imports.  It does not exist in the reference at this revision; the
class below is the value type the example's update_hypothesis (:82-88) passes
between `fusion` and `regularize`: an inverse-depth map with its variance map,
the array form of Hypothesis (src/semi_dense/hypothesis.rs:39-62)."""
import numpy as np


class HypothesisMap(object):
    def __init__(self, inv_depth_map, variance_map):
        inv_depth_map = np.asarray(inv_depth_map, dtype=np.float64)
        variance_map = np.asarray(variance_map, dtype=np.float64)
        if inv_depth_map.shape != variance_map.shape:
            raise ValueError("inv_depth_map and variance_map must have the same shape")
        self.inv_depth_map = inv_depth_map
        self.variance_map = variance_map

    @property
    def shape(self):
        return self.inv_depth_map.shape

    @property
    def depth_map(self):
        from tadataka.numeric import safe_invert
        return safe_invert(self.inv_depth_map)
