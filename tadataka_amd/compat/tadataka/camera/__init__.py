"""tadataka.camera (reference tadataka/camera/__init__.py:1-3)."""
from tadataka.camera.parameters import CameraParameters
from tadataka.camera.model import CameraModel, resize
from tadataka.camera.distortion import FOV, RadTan, NoDistortion
