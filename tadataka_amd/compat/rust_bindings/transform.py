"""rust_bindings.transform (src/py/transform.rs:6-22)."""
from rust_bindings._check import f64
from tadataka_amd import ops


def transform(transform10, points0):
    """(T10 [p;1])[0:3] for every row of points0 [N,3] (src/transform.rs:9-28)."""
    return ops.transform(f64(transform10, 2, "transform10"), f64(points0, 2, "points0"))
