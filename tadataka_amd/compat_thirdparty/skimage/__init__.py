"""Stand-in for the three scikit-image functions the reference's DVO / semi-dense
path and its two examples import (skimage.color.rgb2gray,
skimage.transform.rescale / resize: tadataka/vo/dvo/__init__.py:8-9,
examples/dvo_pose_change.py:1, examples/semi_dense_vo.py:2-3), running on the
MI355X through libtadataka_hip.so.  `import tadataka_amd` appends this directory
at the END of sys.path: with a real scikit-image installed it is never imported.

Third-party behaviour restated from the published algorithms of
scikit-image 0.16.2 (the version setup.py:117 pins): parity unpinned."""
__version__ = "0.16.2+tadataka_amd"
