"""rust_bindings.triangulation (src/py/triangulation.rs:7-26)."""
from rust_bindings._check import f64
from tadataka_amd import ops


def calc_depth0(transform10, x0, x1):
    """Depth of x0 in frame 0 from the pair (x0, x1) (src/triangulation.rs:8-39)."""
    return ops.calc_depth0(f64(transform10, 2, "transform10"), f64(x0, 1, "x0"), f64(x1, 1, "x1"))
