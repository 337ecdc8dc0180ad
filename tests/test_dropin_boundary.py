"""CPU: the drop-in boundary next to the unmodified reference.

The package directory goes FIRST on sys.path, /root/reference behind it, exactly as INTEGRATION.md §1
documents; then the reference's own import block (test.py:16-22) is executed verbatim.  Third-party
modules the reference imports but this image lacks (cv2, skimage, torchvision, numba) are stubbed the way
SURVEY.md §8c describes — they are host I/O dependencies, not part of the path.  Runs in a subprocess so
that the stubs and the reference's modules never leak into the pytest process.
"""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "deep-exemplar-based-video-colorization_amd")
REF = "/root/reference"

_STUBS = textwrap.dedent('''
    import sys, types
    sys.dont_write_bytecode = True          # never write __pycache__ into /root/reference
    def _stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        if "." in name:
            parent, child = name.rsplit(".", 1)
            setattr(sys.modules[parent], child, m)
        return m
    _ident = lambda *a, **k: (lambda f: f)
    _stub("cv2", setNumThreads=lambda n: None)
    _stub("torchvision"); _stub("torchvision.models"); _stub("torchvision.utils")
    _stub("torchvision.transforms", CenterCrop=object)
    _stub("skimage"); _stub("skimage.color"); _stub("skimage.io")
    _stub("skimage.draw", random_shapes=None); _stub("skimage.filters", gaussian=None)
    _stub("skimage.transform", resize=None)
    _stub("numba", jit=_ident, u1=None, u2=None)
''')

# the reference's import block, test.py:16-22, verbatim
_IMPORT_BLOCK = textwrap.dedent('''
    import lib.TestTransforms as transforms
    from models.ColorVidNet import ColorVidNet
    from models.FrameColor import frame_colorization
    from models.NonlocalNet import VGG19_pytorch, WarpNet
    from utils.util import (batch_lab2rgb_transpose_mc, folder2vid, mkdir_if_not,
                            save_frames, tensor_lab2rgb, uncenter_l)
    from utils.util_distortion import CenterPad, Normalize, RGB2Lab, ToTensor
''')


def _run(code):
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + "\n" + r.stderr
    return r.stdout


@pytest.mark.skipif(not os.path.isdir(REF), reason="/root/reference only exists in the build container")
def test_reference_import_block_resolves_with_package_in_front():
    code = _STUBS + f"sys.path[:0] = [{PKG!r}, {REF!r}]\n" + _IMPORT_BLOCK + textwrap.dedent(f'''
        import os, dvc_amd.nets, dvc_amd.frame, dvc_amd.util, utils, utils.util
        pkg, ref = {PKG!r}, {REF!r}
        # hot-path classes and helpers come from this package ...
        assert ColorVidNet is dvc_amd.nets.ColorVidNet and WarpNet is dvc_amd.nets.WarpNet
        assert VGG19_pytorch is dvc_amd.nets.VGG19_pytorch
        assert frame_colorization is dvc_amd.frame.frame_colorization
        assert tensor_lab2rgb is dvc_amd.util.tensor_lab2rgb and uncenter_l is dvc_amd.util.uncenter_l
        assert os.path.dirname(utils.util.__file__) == os.path.join(pkg, "utils")
        # ... everything else keeps resolving to the reference's own files
        for fn in (batch_lab2rgb_transpose_mc, folder2vid, mkdir_if_not, save_frames):
            assert fn.__module__ == "utils._reference_util", fn
            assert fn.__code__.co_filename == os.path.join(ref, "utils", "util.py"), fn.__code__.co_filename
        for cls in (CenterPad, Normalize, RGB2Lab, ToTensor):
            assert sys.modules[cls.__module__].__file__ == os.path.join(ref, "utils", "util_distortion.py")
        assert transforms.__file__ == os.path.join(ref, "lib", "TestTransforms.py")
        assert utils.__path__[0] == os.path.join(pkg, "utils") and os.path.join(ref, "utils") in utils.__path__
        # a name neither side defines is an AttributeError / ImportError, not a silent None
        try:
            from utils.util import no_such_helper
        except ImportError:
            pass
        else:
            raise AssertionError("unknown name resolved")
        # models/ is a namespace package in both trees: the reference's other model files stay reachable
        import importlib.util
        assert importlib.util.find_spec("models.GAN_models").origin == os.path.join(ref, "models", "GAN_models.py")
        assert importlib.util.find_spec("models.spectral_normalization").origin == os.path.join(ref, "models", "spectral_normalization.py")
        # train.py:22 `from models.ContextualLoss import ContextualLoss, ContextualLoss_forward`: the two HIP-backed losses from
        # this package, every other name of that file forwarded to the reference's own (loaded unmodified)
        import dvc_amd.contextual
        from models.ContextualLoss import ContextualLoss, ContextualLoss_forward
        assert ContextualLoss is dvc_amd.contextual.ContextualLoss and ContextualLoss_forward is dvc_amd.contextual.ContextualLoss_forward
        sys.modules["torchvision.transforms"].__dict__.update(Compose=lambda fs: fs, Lambda=lambda f: f,
                                                              Normalize=lambda mean=None, std=None: None, ToPILImage=lambda: None)
        from models.ContextualLoss import ContextualLoss_complex, ChamferDistance_loss
        for cls in (ContextualLoss_complex, ChamferDistance_loss):
            assert sys.modules[cls.__module__].__file__ == os.path.join(ref, "models", "ContextualLoss.py"), cls
        print("OK")
    ''')
    assert "OK" in _run(code)


def test_package_alone_gives_clear_error_for_reference_only_names():
    """Without the reference behind it, hot-path helpers import; host I/O helpers fail with a clear message."""
    code = textwrap.dedent(f'''
        import sys
        sys.path.insert(0, {PKG!r})
        from utils.util import tensor_lab2rgb, uncenter_l, gray2rgb_batch, feature_normalize, vgg_preprocess
        assert uncenter_l(-50.0) == 0.0
        try:
            from utils.util import save_frames
        except ImportError as e:
            assert "save_frames" in str(e)
        else:
            raise AssertionError("save_frames resolved without the reference")
        print("OK")
    ''')
    assert "OK" in _run(code)
