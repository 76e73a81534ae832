"""tadataka.utils (reference tadataka/utils.py:35-54): the inclusive float
range test used as the DVO masks."""
import numpy as np

from tadataka.decorator import allow_1d


def _is_in_image_range(keypoints, image_shape):
    height, width = image_shape
    xs, ys = keypoints[:, 0], keypoints[:, 1]
    return (0 <= xs) & (xs <= width - 1) & (0 <= ys) & (ys <= height - 1)


@allow_1d(which_argument=0)
def is_in_image_range(keypoints, image_shape):
    """x in [0, width-1] and y in [0, height-1], float coordinates accepted."""
    return _is_in_image_range(np.asarray(keypoints), image_shape[0:2])


def round_int(X):
    return np.round(X, 0).astype(np.int64)


def merge_dicts(*dicts):
    merged = dict()
    for d in dicts:
        merged.update(d)
    return merged


def value_list(dict_, keys):
    return [dict_[k] for k in keys]


def radian_to_degree(radian):
    return radian / np.pi * 180
