"""Stand-in for the pybind11/Eigen module tadataka.camera._normalizer
(reference tadataka/camera/_normalizer.cpp:12-34) on the MI355X."""
import numpy as np

from tadataka_amd import ops


def _cam(focal_length, offset):
    return np.concatenate([np.asarray(focal_length, dtype=np.float64),
                           np.asarray(offset, dtype=np.float64)])


def normalize(keypoints, focal_length, offset):
    """(keypoints - offset) / focal_length, rowwise; integer input is converted
    to float64 as pybind11/Eigen does."""
    return ops.normalize(np.asarray(keypoints, dtype=np.float64), _cam(focal_length, offset))


def unnormalize(keypoints, focal_length, offset):
    """keypoints * focal_length + offset, rowwise."""
    return ops.unnormalize(np.asarray(keypoints, dtype=np.float64), _cam(focal_length, offset))
