"""Drop-in for /root/reference/models/ContextualLoss.py (train.py:22 imports `ContextualLoss, ContextualLoss_forward`).
The two loss modules are the HIP-backed ones (dvc_amd.contextual); every other name of the reference's file
(`ContextualLoss_complex`, the Chamfer losses, `post_processing`) is forwarded, on first use, to the next
`models/ContextualLoss.py` on `models.__path__` — the reference's own file, loaded unmodified."""
import importlib.util as _ilu
import os as _os
import sys as _sys

from dvc_amd.contextual import ContextualLoss, ContextualLoss_forward  # noqa: F401

_HERE = _os.path.dirname(_os.path.abspath(__file__))
_REF_NAME = "models._reference_ContextualLoss"


def _reference_module():
    mod = _sys.modules.get(_REF_NAME)
    if mod is not None:
        return mod
    import models as _pkg
    for d in list(getattr(_pkg, "__path__", [])):
        cand = _os.path.join(d, "ContextualLoss.py")
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
    ref = _reference_module()
    if ref is None:
        raise AttributeError(f"module 'models.ContextualLoss' has no attribute '{name}' and no reference "
                             "models/ContextualLoss.py is on sys.path behind it to forward to")
    return getattr(ref, name)
