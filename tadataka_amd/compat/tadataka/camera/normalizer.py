"""tadataka.camera.normalizer (reference tadataka/camera/normalizer.py:8-33)."""
from tadataka.camera._normalizer import normalize, unnormalize


class Normalizer(object):
    def __init__(self, camera_parameters):
        self.focal_length = camera_parameters.focal_length
        self.offset = camera_parameters.offset

    def normalize(self, keypoints):
        """Image coordinates -> normalized plane: (u - o) / f."""
        return normalize(keypoints, self.focal_length, self.offset)

    def unnormalize(self, keypoints):
        """Normalized plane -> image coordinates: x * f + o."""
        return unnormalize(keypoints, self.focal_length, self.offset)
