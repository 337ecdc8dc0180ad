"""GPU parity of the HIP contextual losses (dvc_amd.contextual) against the oracle restatement of
/root/reference/models/ContextualLoss.py:29-126 — values against the reference fixtures and the float64 truth, gradients
w.r.t. X against float64 autograd, at the sizes train.py:649-668 uses (relu5_1 13x24 and relu4_1 27x48 at 216x384; the
downsampled relu3_1 is 27x48 too) and at a size with several row blocks."""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import contextual_oracle as O  # noqa: E402


def _mods():
    from models.ContextualLoss import ContextualLoss, ContextualLoss_forward
    return {"fwd": (ContextualLoss_forward(), O.contextual_loss_forward), "bwd": (ContextualLoss(), O.contextual_loss)}


def test_contextual_loss_vs_reference_fixtures(golden_dir):
    """Forward values and dX against what the unmodified reference produced (float32 CPU): the HIP path is float32 with a
    different summation order, so the comparison is at fp32-rounding level, not bit for bit."""
    for f in sorted(glob.glob(os.path.join(golden_dir, "contextual_*.npz"))):
        g = np.load(f)
        B, C, H, W = (int(v) for v in g["shape"])
        X, Y = O.synth_features(int(g["seed"]), B, C, H, W)
        for tag, (mod, _) in _mods().items():
            x = X.cuda().requires_grad_(True)
            loss = mod(x, Y.cuda(), h=float(g["h"]), feature_centering=bool(int(g["centre"])))
            assert loss.shape == (B,) and loss.requires_grad
            loss.sum().backward()
            ref_l, ref_g = torch.from_numpy(g[f"loss_{tag}"]), torch.from_numpy(g[f"dx_{tag}"])
            el = (loss.detach().cpu() - ref_l).abs().max().item()
            eg = (x.grad.cpu() - ref_g).abs().max().item() / ref_g.abs().max().item()
            print(f"{os.path.basename(f)} {tag}: loss err {el:.2e}, dX rel err {eg:.2e}")
            assert el < 2e-6 * max(1.0, ref_l.abs().max().item()), (f, tag, el)       # measured <= 2.4e-7
            assert eg < 5e-5, (f, tag, eg)                                             # measured <= 1.7e-6


@pytest.fixture(params=["vendor-gemm", "engine-gemm"])
def gemm_mode(request):
    """r06: the N x N products run through the vendor's batched GEMM by default (ops.bmm) or on this library's 1x1-convolution
    engine (DVC_GEMM_LIB=0): both against the same float64 autograd, the same tolerances."""
    from dvc_amd import ops
    before = ops.gemm_lib()
    ops.set_gemm_lib(request.param == "vendor-gemm")
    yield request.param
    ops.set_gemm_lib(before)


@pytest.mark.parametrize("C,H,W,B,h,centre", [(512, 13, 24, 2, 0.1, True), (512, 27, 48, 1, 0.1, True), (256, 27, 48, 2, 0.1, True),
                                              (64, 24, 40, 1, 0.2, False), (128, 7, 9, 3, 0.1, True)])
def test_contextual_loss_vs_float64_autograd(C, H, W, B, h, centre, gemm_mode):
    X, Y = O.synth_features(1000 + C + H, B, C, H, W)
    gout = torch.linspace(0.5, 1.5, B)
    for tag, (mod, fn) in _mods().items():
        x64 = X.double().requires_grad_(True)
        l64 = fn(x64, Y.double(), h=h, feature_centering=centre)
        (l64 * gout.double()).sum().backward()
        x = X.cuda().requires_grad_(True)
        loss = mod(x, Y.cuda(), h=h, feature_centering=centre)
        (loss * gout.cuda()).sum().backward()
        torch.cuda.synchronize()
        el = (loss.detach().cpu().double() - l64.detach()).abs().max().item()
        scale = x64.grad.abs().max().item()
        eg = (x.grad.cpu().double() - x64.grad).abs().max().item()
        print(f"contextual {tag} C={C} {H}x{W} B={B} h={h}: loss {l64.tolist()} err {el:.2e}; dX max err {eg:.2e} (max |dX| {scale:.2e})")
        assert l64.min().item() > 0.05                      # a non-degenerate affinity
        assert el < 2e-6 * max(1.0, l64.abs().max().item()), (tag, el)           # measured <= 3.4e-7
        assert eg <= 5e-5 * scale, (tag, eg, scale)                                # measured <= 1.6e-6 of the largest gradient
        # deterministic
        x2 = X.cuda().requires_grad_(True)
        l2 = mod(x2, Y.cuda(), h=h, feature_centering=centre)
        (l2 * gout.cuda()).sum().backward()
        assert torch.equal(l2, loss) and torch.equal(x2.grad, x.grad)


def test_contextual_loss_contract():
    """Y is data (train.py passes it detached): a Y that requires grad raises; no_grad calls work; the loss of X against
    itself is ~0 (every row's best match is its own column with d ~ 0)."""
    from models.ContextualLoss import ContextualLoss_forward
    X, Y = O.synth_features(5, 1, 64, 6, 8)
    m = ContextualLoss_forward()
    with pytest.raises(NotImplementedError):
        m(X.cuda(), Y.cuda().requires_grad_(True))
    with torch.no_grad():
        l = m(X.cuda(), Y.cuda())
        assert l.shape == (1,) and not l.requires_grad
        same = m(Y.cuda(), Y.cuda())
    assert same.item() < 1e-3 < l.item()
