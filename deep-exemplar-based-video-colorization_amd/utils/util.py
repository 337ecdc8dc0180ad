"""Drop-in for the hot-path tensor helpers of /root/reference/utils/util.py (test.py:20-21 imports
`tensor_lab2rgb, uncenter_l`; FrameColor uses gray2rgb_batch / feature_normalize)."""
from dvc_amd.util import (center_ab, center_l, feature_normalize, gray2rgb_batch,  # noqa: F401
                          tensor_lab2rgb, uncenter_l, vgg_preprocess)
