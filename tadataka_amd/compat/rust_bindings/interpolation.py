"""rust_bindings.interpolation (src/py/interpolation.rs:6-22)."""
from rust_bindings._check import f64
from tadataka_amd import ops


def interpolation(image, coordinates):
    """Bilinear samples of `image` [H,W] at `coordinates` [M,2] = (x, y)
    (src/interpolation.rs:9-43).  The Rust code panics on coordinates outside
    the image; here that is a ValueError."""
    f64(image, 2, "image"); f64(coordinates, 2, "coordinates")
    return ops.interpolation(image, coordinates)
