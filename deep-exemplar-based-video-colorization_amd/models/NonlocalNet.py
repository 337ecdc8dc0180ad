"""Drop-in for /root/reference/models/NonlocalNet.py (hot-path classes only: test.py:19 imports
`VGG19_pytorch, WarpNet`).  MI355X HIP implementation lives in dvc_amd.nets."""
from dvc_amd.nets import VGG19_pytorch, WarpNet  # noqa: F401
