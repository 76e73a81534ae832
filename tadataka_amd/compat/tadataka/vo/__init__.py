"""tadataka.vo: only the direct front end (dvo, semi_dense) is part of the
MI355X hot-path build; `FeatureBasedVO` (reference tadataka/vo/__init__.py:1) is
the sparse OpenCV-bound front end and is not provided."""
