"""tadataka.numeric (reference tadataka/numeric.py:1-2)."""


def safe_invert(v, epsilon=1e-16):
    """1 / (v + epsilon); examples/semi_dense_vo.py:52 turns a variance map into
    DVO weights with it.  A map that lives on the device (what rust_bindings.semi_dense
    returns) is inverted there and stays there."""
    from tadataka_amd.ops import DeviceMap
    if isinstance(v, DeviceMap) and v._owner is None:
        return v.safe_invert(epsilon)
    return 1 / (v + epsilon)
