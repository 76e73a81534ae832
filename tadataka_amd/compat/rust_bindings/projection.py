"""rust_bindings.projection (src/py/projection.rs:7-51)."""
from rust_bindings._check import f64
from tadataka_amd import ops


def project_vecs(points):
    return ops.project_vecs(f64(points, 2, "points"))


def project_vec(point):
    return ops.project_vecs(f64(point, 1, "point").reshape(1, 3))[0]


def inv_project_vecs(xs, depths):
    return ops.inv_project_vecs(f64(xs, 2, "xs"), f64(depths, 1, "depths"))


def inv_project_vec(x, depth):
    return ops.inv_project_vecs(f64(x, 1, "x").reshape(1, 2), [float(depth)])[0]
