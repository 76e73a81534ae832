"""tadataka.decorator (reference tadataka/decorator.py:4-21)."""
import functools

import numpy as np


def allow_1d(which_argument):
    """Lets a function written for [N, d] arrays also take one d-vector at
    position `which_argument` (the result's first row is returned)."""
    def wrap(function):
        @functools.wraps(function)
        def inner(*args, **kwargs):
            dims = np.ndim(args[which_argument])
            if dims == 2:
                return function(*args, **kwargs)
            if dims != 1:
                raise ValueError(
                    f"Argument number {which_argument} has to be 1d or 2d array")
            promoted = list(args)
            promoted[which_argument] = np.atleast_2d(promoted[which_argument])
            return function(*promoted, **kwargs)[0]
        return inner
    return wrap
