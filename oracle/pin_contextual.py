"""Pin oracle/contextual_oracle.py against the UNMODIFIED reference module and write tests/golden/contextual_*.npz.

TEST INFRASTRUCTURE ONLY.  Run in the build container (the only place /root/reference exists):

    python oracle/pin_contextual.py            # checks + (re)writes the fixtures
    python oracle/pin_contextual.py --check    # checks only

/root/reference/models/ContextualLoss.py imports torchvision at module level for an image post-processing helper the
losses never touch; torchvision is absent here, so it is stubbed with the four constructor names that line uses
(SURVEY.md §8(c) recipe).  The reference's `ContextualLoss_forward` / `ContextualLoss` then run as they are, in float32,
single-threaded, values and autograd gradients w.r.t. X; the oracle must reproduce both bit for bit.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")
CASES = [("contextual_5x8_c64", 11, 2, 64, 5, 8, 0.1, True), ("contextual_9x14_c128", 12, 1, 128, 9, 14, 0.1, True),
         ("contextual_6x7_c32_nocentre_h05", 13, 2, 32, 6, 7, 0.5, False)]


def import_reference():
    tv = types.ModuleType("torchvision")
    tr = types.ModuleType("torchvision.transforms")
    tr.Compose = lambda fs: fs
    tr.Lambda = lambda f: f
    tr.Normalize = lambda mean=None, std=None: None
    tr.ToPILImage = lambda: None
    tv.transforms = tr
    sys.modules["torchvision"], sys.modules["torchvision.transforms"] = tv, tr
    for sub in ("models", "utils"):                      # utils/util.py:9 imports torchvision.utils (plotting helpers)
        m = types.ModuleType("torchvision." + sub)
        setattr(tv, sub, m)
        sys.modules["torchvision." + sub] = m
    for name in ("cv2", "skimage", "skimage.color", "skimage.io"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["cv2"].setNumThreads = lambda n: None
    sys.path.insert(0, REF)
    import models.ContextualLoss as rcl
    assert rcl.__file__.startswith(REF), rcl.__file__
    sys.path.remove(REF)
    return rcl


def main(write=True):
    pkg = os.path.join(ROOT, "deep-exemplar-based-video-colorization_amd")
    while pkg in sys.path:
        sys.path.remove(pkg)
    rcl = import_reference()
    sys.path.insert(0, ROOT)
    from oracle import contextual_oracle as O
    torch.set_num_threads(1)
    ok = True
    for name, seed, B, C, H, W, h, centre in CASES:
        X, Y = O.synth_features(seed, B, C, H, W)
        out = {}
        for tag, ref_mod, fn in (("fwd", rcl.ContextualLoss_forward(), O.contextual_loss_forward),
                                 ("bwd", rcl.ContextualLoss(), O.contextual_loss)):
            xr = X.clone().requires_grad_(True)
            lr = ref_mod(xr, Y, h=h, feature_centering=centre)
            lr.sum().backward()
            xo = X.clone().requires_grad_(True)
            lo = fn(xo, Y, h=h, feature_centering=centre)
            lo.sum().backward()
            same = torch.equal(lr, lo) and torch.equal(xr.grad, xo.grad)
            print(f"{name} {tag}: reference loss {lr.tolist()}  oracle == reference (values and dX, bit for bit): {same}")
            ok &= same
            out[f"loss_{tag}"] = lr.detach().numpy()
            out[f"dx_{tag}"] = xr.grad.numpy()
        if write:
            np.savez_compressed(os.path.join(GOLD, name + ".npz"), seed=seed, shape=np.array([B, C, H, W]), h=h,
                                centre=int(centre), **out)
    if not ok:
        sys.exit("oracle != reference")
    print("contextual oracle pinned" + (" and fixtures written" if write else ""))


if __name__ == "__main__":
    main(write="--check" not in sys.argv)
