"""GPU parity of the differentiable fused correlation (dvc_amd.corr_autograd) against autograd
through the oracle's `correlate` (models/NonlocalNet.py:477-500) — the path train.py:402-427 differentiates, at its
temperature 0.01 and a batch of 2.  The oracle runs in float64 (the truth), the HIP path in float32."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _inputs(B, h, w, seed):
    g = torch.Generator().manual_seed(seed)
    P = h * w
    base = torch.randn(B, 256, 8, generator=g)                       # a few shared directions: peaked, non-trivial softmax rows
    th = torch.randn(B, 256, P, generator=g) + 2.0 * base[:, :, torch.randint(0, 8, (P,), generator=g)]
    ph = torch.randn(B, 256, P, generator=g) + 2.0 * base[:, :, torch.randint(0, 8, (P,), generator=g)]
    th = th - th.mean(-1, keepdim=True)
    ph = ph - ph.mean(-1, keepdim=True)
    th = th / th.norm(dim=1, keepdim=True)
    ph = ph / ph.norm(dim=1, keepdim=True)
    lab = torch.randn(B, 3, 4 * h, 4 * w, generator=g) * 30
    gy = torch.randn(B, 3, h, w, generator=g)
    gs = torch.randn(B, 1, h, w, generator=g)
    return th, ph, lab, gy, gs


@pytest.mark.parametrize("h,w,B,T", [(12, 20, 2, 0.01), (10, 16, 2, 0.01), (27, 48, 1, 0.01), (12, 20, 1, 0.005), (9, 7, 2, 0.05)])
def test_fused_correlation_backward_vs_oracle_autograd(h, w, B, T):
    from dvc_amd import ops
    from dvc_amd.corr_autograd import fused_correlation
    from oracle import dvc_oracle as O
    th, ph, lab, gy, gs = _inputs(B, h, w, 100 * h + w)
    # truth: float64 autograd through the reference's op sequence
    th64, ph64 = th.double().requires_grad_(True), ph.double().requires_grad_(True)
    y64, sim64, f64 = O.correlate(th64, ph64, lab.double(), T)
    ((y64 * gy.double()).sum() + (sim64 * gs.double()).sum()).backward()
    # HIP
    thd, phd = th.cuda().requires_grad_(True), ph.cuda().requires_grad_(True)
    blab = ops.avgpool4x4(lab.cuda()).view(B, 3, -1)
    y, sim, amax = fused_correlation(thd, phd, blab, T, h, w)
    assert y.requires_grad and sim.requires_grad and not amax.requires_grad
    ((y * gy.cuda()).sum() + (sim * gs.cuda()).sum()).backward()
    torch.cuda.synchronize()
    assert (y.detach().cpu().double() - y64.detach()).abs().max().item() < 5e-3 * max(1.0, 0.01 / T)
    assert (sim.detach().cpu().double() - sim64.detach()).abs().max().item() < 2e-6
    for name, got, ref in (("theta", thd.grad, th64.grad), ("phi", phd.grad, ph64.grad)):
        err = (got.cpu().double() - ref).abs().max().item()
        scale = ref.abs().max().item()
        print(f"corr backward {h}x{w} B={B} T={T}: d{name} max err {err:.3e} (max |grad| {scale:.3e})")
        assert err <= 2e-3 * scale, (name, err, scale)
    # the similarity branch alone: the gradient of max_j f lands on the arg-max column only
    thd2, phd2 = th.cuda().requires_grad_(True), ph.cuda().requires_grad_(True)
    _, sim2, amax2 = fused_correlation(thd2, phd2, blab, T, h, w)
    sim2.sum().backward()
    expect = torch.gather(ph.cuda(), 2, amax2.long().unsqueeze(1).expand(B, 256, h * w))
    assert (thd2.grad - expect).abs().max().item() < 1e-5


def test_requires_grad_inputs_raise_on_the_inference_modules():
    """The drop-in modules are inference-only: an input that requires grad, with autograd enabled, raises instead of
    silently returning a tensor without history; under torch.no_grad() (test.py:83) it runs."""
    import contextlib
    import io
    from dvc_amd import synth
    from models.ColorVidNet import ColorVidNet
    with contextlib.redirect_stdout(io.StringIO()):
        col = ColorVidNet(7)
    col.load_state_dict(synth.colorvidnet_state_dict(0, contractive=True))
    col.eval().cuda()
    x = torch.randn(1, 7, 16, 24, device="cuda", requires_grad=True)
    with pytest.raises(NotImplementedError):
        col(x)
    with torch.no_grad():
        assert col(x).shape == (1, 2, 16, 24)
