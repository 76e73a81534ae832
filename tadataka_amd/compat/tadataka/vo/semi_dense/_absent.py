def absent(module, name):
    """examples/semi_dense_vo.py:13-21 imports these names, but the reference's
    tadataka/vo/semi_dense/ holds only flag.py at this revision and
    main() never calls them.  They import; calling raises."""
    def stub(*args, **kwargs):
        raise NotImplementedError(
            f"tadataka.vo.semi_dense.{module}.{name} does not exist in the reference at this "
            "revision; the semi-dense path runs through rust_bindings.semi_dense")
    stub.__name__ = name
    return stub
