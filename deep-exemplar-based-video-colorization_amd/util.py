"""Drop-in for /root/reference/utils/util.py on the hot path.

test.py:20-21 does
    from utils.util import (batch_lab2rgb_transpose_mc, folder2vid, mkdir_if_not,
                            save_frames, tensor_lab2rgb, uncenter_l)
The tensor helpers on the hot path (`tensor_lab2rgb`, `uncenter_l`, `gray2rgb_batch`,
`feature_normalize`, `vgg_preprocess`, `center_l`, `center_ab`) are this package's HIP-backed
versions (dvc_amd/util.py).  Every other name (`save_frames`, `folder2vid`, `mkdir_if_not`,
`batch_lab2rgb_transpose_mc`, the plotting / training helpers ...) is forwarded, on first use, to the
next `utils/util.py` found on `utils.__path__` — i.e. the reference's own file, loaded unmodified
under the module name `utils._reference_util`.  Its import errors (cv2 / skimage / torchvision
missing) surface exactly as they would from the reference.
"""
import importlib.util as _ilu
import os as _os
import sys as _sys

from dvc_amd.util import (ab_mean, ab_norm, center_ab, center_l, feature_normalize,  # noqa: F401
                          gray2rgb_batch, l_mean, l_norm, tensor_lab2rgb, uncenter_l, vgg_preprocess)

_HERE = _os.path.dirname(_os.path.abspath(__file__))
_REF_NAME = "utils._reference_util"


def _reference_util():
    """The reference's utils/util.py (the first one on utils.__path__ that is not this file), or None."""
    mod = _sys.modules.get(_REF_NAME)
    if mod is not None:
        return mod
    import utils as _pkg
    for d in list(getattr(_pkg, "__path__", [])):
        cand = _os.path.join(d, "util.py")
        if _os.path.abspath(d) == _HERE or not _os.path.isfile(cand):
            continue
        spec = _ilu.spec_from_file_location(_REF_NAME, cand)
        mod = _ilu.module_from_spec(spec)
        _sys.modules[_REF_NAME] = mod
        try:
            spec.loader.exec_module(mod)
        except BaseException:
            del _sys.modules[_REF_NAME]
            raise
        return mod
    return None


def __getattr__(name):   # PEP 562: only reached for names this module does not define
    if name.startswith("__") and name.endswith("__"):
        raise AttributeError(name)
    ref = _reference_util()
    if ref is None:
        raise AttributeError(
            f"module 'utils.util' has no attribute '{name}': it is not one of the hot-path helpers this drop-in "
            "provides, and no reference utils/util.py is on sys.path behind it to forward to")
    return getattr(ref, name)
