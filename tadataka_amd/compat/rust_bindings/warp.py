"""rust_bindings.warp (src/py/warp.rs:23-75)."""
from rust_bindings._check import f64
from tadataka_amd import ops


def warp_vecs(transform10, xs, depths):
    """(xs1 [N,2], depths1 [N]) = project(T10 inv_project(xs, depths)) (src/warp.rs:31-50)."""
    f64(transform10, 2, "transform10"); f64(xs, 2, "xs"); f64(depths, 1, "depths")
    return ops.warp_vecs(transform10, xs, depths)


def warp_vec(transform10, x0, depth0):
    """Single point (src/warp.rs:11-29): returns (x1 [2], depth1 float)."""
    f64(transform10, 2, "transform10"); f64(x0, 1, "x0")
    xs1, d1 = ops.warp_vecs(transform10, x0.reshape(1, 2), [float(depth0)])
    return xs1[0], float(d1[0])
