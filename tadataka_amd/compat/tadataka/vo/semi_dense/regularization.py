"""tadataka.vo.semi_dense.regularization (imported by examples/semi_dense_vo.py:13,
called at :86): 3x3 inverse-variance smoothing of an inverse-depth map on the
MI355X through tdk_regularize (src/semi_dense/regularization.rs:5-64).

The Rust function works on DEPTH maps and takes a flag map; the example passes a
hypothesis map only and expects an INVERSE-depth map back, so this wrapper
inverts on the way in and out (tadataka.numeric.safe_invert, as the Rust `Inv`
type does) and treats every pixel as Success unless a flag map is given."""
import numpy as np

from tadataka.numeric import safe_invert
from tadataka_amd import ops


def regularize(hypothesis, flag_map=None):
    if flag_map is None:
        flag_map = np.zeros(hypothesis.shape, dtype=np.int64)
    depth = ops.regularize(safe_invert(hypothesis.inv_depth_map), hypothesis.variance_map, flag_map)
    return safe_invert(depth)
