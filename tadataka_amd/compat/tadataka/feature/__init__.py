"""tadataka.feature: sparse feature extraction / matching (OpenCV, scikit-image)
belongs to the feature-based front end, outside the hot path."""


def _out_of_scope(name):
    def stub(*args, **kwargs):
        raise NotImplementedError(f"tadataka.feature.{name} is not part of the MI355X hot-path build")
    stub.__name__ = name
    return stub


extract_features = _out_of_scope("extract_features")
empty_match = _out_of_scope("empty_match")


class Matcher(object):
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("tadataka.feature.Matcher is not part of the MI355X hot-path build")
