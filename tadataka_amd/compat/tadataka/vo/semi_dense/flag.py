"""tadataka.vo.semi_dense.flag (reference tadataka/vo/semi_dense/flag.py:1-14):
per-pixel outcome codes of update_depth, identical to src/semi_dense/flag.rs:3-14."""
from enum import IntEnum


class ResultFlag(IntEnum):
    SUCCESS = 0
    HYPOTHESIS_OUT_OF_SERCH_RANGE = -1
    KEY_OUT_OF_RANGE = -2
    REF_CLOSE_OUT_OF_RANGE = -3
    REF_FAR_OUT_OF_RANGE = -4
    REF_EPIPOLAR_TOO_SHORT = -5
    INSUFFICIENT_GRADIENT = -6
    NEGATIVE_PRIOR_DEPTH = -7
    NEGATIVE_REF_DEPTH = -8
    NOT_PROCESSED = -9
