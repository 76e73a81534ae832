"""MI355X stand-in for the reference's pyo3 crate `rust_bindings`
(Cargo.toml:7-9, src/py/*.rs): same submodules, functions and classes, backed by
libtadataka_hip.so.  Array arguments must be float64 (uint64 for age maps,
int64 for pixel indices) exactly as rust-numpy demands."""
from rust_bindings import (camera, homogeneous, interpolation, projection,  # noqa: F401
                           semi_dense, transform, triangulation, warp)
