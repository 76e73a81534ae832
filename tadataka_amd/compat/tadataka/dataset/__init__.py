"""tadataka.dataset: dataset readers are I/O outside the hot path;
the names the two examples import exist and raise when used."""


def _out_of_scope(name):
    class Dataset(object):
        def __init__(self, *args, **kwargs):
            raise NotImplementedError(
                f"tadataka.dataset.{name} is not part of the MI355X hot-path build; "
                "feed frames as float64 arrays (see tadataka_amd.synthetic for test data)")
    Dataset.__name__ = name
    return Dataset


NewTsukubaDataset = _out_of_scope("NewTsukubaDataset")
TumRgbdDataset = _out_of_scope("TumRgbdDataset")
EurocDataset = _out_of_scope("EurocDataset")
