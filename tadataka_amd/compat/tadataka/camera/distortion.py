"""tadataka.camera.distortion: only the identity model is part of the DVO /
semi-dense hot path (both examples strip distortion, examples/dvo_pose_change.py:
17-19, examples/semi_dense_vo.py:47-50).  FOV / RadTan are importable names that
raise when constructed (out of scope)."""


class NoDistortion(object):
    params = []

    def distort(self, keypoints):
        return keypoints

    def undistort(self, keypoints):
        return keypoints

    def __eq__(self, another):
        return isinstance(another, NoDistortion)


def _out_of_scope(name):
    class Model(object):
        def __init__(self, *args, **kwargs):
            raise NotImplementedError(
                f"distortion model {name} is outside the MI355X hot-path build; "
                "use CameraModel(camera_parameters, distortion_model=None)")

        @classmethod
        def from_params(cls, params):
            return cls(params)
    Model.__name__ = name
    return Model


FOV = _out_of_scope("FOV")
RadTan = _out_of_scope("RadTan")
