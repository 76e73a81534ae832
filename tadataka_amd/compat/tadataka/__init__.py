"""Drop-in `tadataka` package for the MI355X build: keeps the reference's Python
API on the DVO / semi-dense / BA hot path and runs the
per-pixel work on libtadataka_hip.so.  Sub-systems that are out of scope for the
hot path (feature-based VO, datasets, plotting) are import-compatible shells."""
