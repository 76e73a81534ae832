"""tadataka.numeric (reference tadataka/numeric.py:1-2)."""


def safe_invert(v, epsilon=1e-16):
    """1 / (v + epsilon); examples/semi_dense_vo.py:52 turns a variance map into
    DVO weights with it."""
    return 1 / (v + epsilon)
