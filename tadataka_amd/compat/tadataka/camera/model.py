"""tadataka.camera.model (reference tadataka/camera/model.py:26-74)."""
from tadataka.camera.distortion import NoDistortion
from tadataka.camera.normalizer import Normalizer
from tadataka.camera.parameters import CameraParameters
from tadataka.decorator import allow_1d


class CameraModel(object):
    def __init__(self, camera_parameters, distortion_model):
        self.normalizer = Normalizer(camera_parameters)
        self.camera_parameters = camera_parameters
        self.distortion_model = (NoDistortion() if distortion_model is None
                                 else distortion_model)

    @allow_1d(which_argument=1)
    def normalize(self, keypoints):
        """Image coordinates -> (undistorted) normalized image plane."""
        return self.distortion_model.undistort(self.normalizer.normalize(keypoints))

    @allow_1d(which_argument=1)
    def unnormalize(self, normalized_keypoints):
        """Normalized image plane -> image coordinates."""
        return self.normalizer.unnormalize(
            self.distortion_model.distort(normalized_keypoints))

    def __str__(self):
        params = self.camera_parameters.params + list(self.distortion_model.params)
        return ' '.join([type(self.distortion_model).__name__] + [str(v) for v in params])

    def __eq__(self, another):
        return (self.camera_parameters == another.camera_parameters and
                self.distortion_model == another.distortion_model)


def resize(cm, scale):
    """Intrinsics of the image rescaled by `scale`: f and o both multiply."""
    p = cm.camera_parameters
    return CameraModel(CameraParameters(p.focal_length * scale, p.offset * scale),
                       cm.distortion_model)
