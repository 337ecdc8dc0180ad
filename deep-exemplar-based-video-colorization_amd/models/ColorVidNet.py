"""Drop-in for /root/reference/models/ColorVidNet.py (test.py:17)."""
from dvc_amd.nets import ColorVidNet  # noqa: F401
