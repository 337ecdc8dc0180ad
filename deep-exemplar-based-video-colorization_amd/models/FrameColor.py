"""Drop-in for /root/reference/models/FrameColor.py (test.py:18)."""
from dvc_amd.frame import frame_colorization, warp_color  # noqa: F401
