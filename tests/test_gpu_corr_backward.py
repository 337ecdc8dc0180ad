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


@pytest.fixture(params=["vendor-gemm", "engine-gemm"])
def gemm_mode(request):
    """r06: the backward's three plain GEMMs run through the vendor's batched GEMM by default (ops.bmm) or on this library's
    1x1-convolution engine (DVC_GEMM_LIB=0): both against the same oracle, the same tolerances."""
    from dvc_amd import ops
    before = ops.gemm_lib()
    ops.set_gemm_lib(request.param == "vendor-gemm")
    yield request.param
    ops.set_gemm_lib(before)


@pytest.mark.parametrize("h,w,B,T", [(12, 20, 2, 0.01), (10, 16, 2, 0.01), (27, 48, 1, 0.01), (12, 20, 1, 0.005), (9, 7, 2, 0.05)])
def test_fused_correlation_backward_vs_oracle_autograd(h, w, B, T, gemm_mode):
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


@pytest.mark.parametrize("h,w,B,T,scale", [(12, 20, 2, 0.01, 1e-4), (10, 16, 1, 0.01, 0.5), (12, 20, 1, 0.005, 3.0), (27, 48, 1, 0.01, 0.5)])
def test_fused_correlation_backward_with_wta_scale(h, w, B, T, scale, gemm_mode):
    """WTA_scale (NonlocalNet.py:288-327, `WTA_scale_weight != 1`), differentiated (r05): float64 autograd through
    oracle.correlate — whose `wta_scale` is the reference's autograd.Function restated, backward constant 1e-4 included,
    pinned bit-exact against the reference class by oracle/pin_reference.py — against the HIP path: forward through the fused
    kernel's two-pass WTA instantiation, backward through dvc_corr_softmax_bwd(wta_scale).  The similarity map's gradient is
    not re-weighted (it is taken before WTA_scale, NonlocalNet.py:481-483): checked on its own."""
    from dvc_amd import ops
    from dvc_amd.corr_autograd import fused_correlation
    from oracle import dvc_oracle as O
    th, ph, lab, gy, gs = _inputs(B, h, w, 100 * h + w + 7)
    th64, ph64 = th.double().requires_grad_(True), ph.double().requires_grad_(True)
    y64, sim64, _ = O.correlate(th64, ph64, lab.double(), T, WTA_scale_weight=scale)
    ((y64 * gy.double()).sum() + (sim64 * gs.double()).sum()).backward()
    thd, phd = th.cuda().requires_grad_(True), ph.cuda().requires_grad_(True)
    blab = ops.avgpool4x4(lab.cuda()).view(B, 3, -1)
    y, sim, _ = fused_correlation(thd, phd, blab, T, h, w, WTA_scale_weight=scale)
    ((y * gy.cuda()).sum() + (sim * gs.cuda()).sum()).backward()
    torch.cuda.synchronize()
    # (forward: the re-weighted softmax is peaked or flat depending on the scale; the bound is the plain case's)
    assert (y.detach().cpu().double() - y64.detach()).abs().max().item() < 5e-3 * max(1.0, 0.01 / T) * max(1.0, scale)
    assert (sim.detach().cpu().double() - sim64.detach()).abs().max().item() < 2e-6
    for name, got, ref in (("theta", thd.grad, th64.grad), ("phi", phd.grad, ph64.grad)):
        err = (got.cpu().double() - ref).abs().max().item()
        sc = ref.abs().max().item()
        print(f"corr backward WTA scale={scale} {h}x{w} B={B} T={T}: d{name} max err {err:.3e} (max |grad| {sc:.3e})")
        assert err <= 2e-3 * sc, (name, err, sc)
    # the re-weighting is really differentiated: the plain path's gradient is something else
    thp, php = th.cuda().requires_grad_(True), ph.cuda().requires_grad_(True)
    yp, _, _ = fused_correlation(thp, php, blab, T, h, w)
    (yp * gy.cuda()).sum().backward()
    thw, phw = th.cuda().requires_grad_(True), ph.cuda().requires_grad_(True)
    yw, _, _ = fused_correlation(thw, phw, blab, T, h, w, WTA_scale_weight=scale)
    (yw * gy.cuda()).sum().backward()
    assert (thp.grad - thw.grad).abs().max().item() > 1e-3 * thp.grad.abs().max().item()
    # similarity branch alone: unscaled, on the arg-max column only
    th2, ph2 = th.cuda().requires_grad_(True), ph.cuda().requires_grad_(True)
    _, sim2, amax2 = fused_correlation(th2, ph2, blab, T, h, w, WTA_scale_weight=scale)
    sim2.sum().backward()
    expect = torch.gather(ph.cuda(), 2, amax2.long().unsqueeze(1).expand(B, 256, h * w))
    assert (th2.grad - expect).abs().max().item() < 1e-5


def _oracle_grads_row_chunked(th, ph, lab, gy, gs, T, rows=1024):
    """float64 autograd through the reference's op sequence (NonlocalNet.py:477-500, as oracle.correlate restates it),
    evaluated `rows` query rows at a time so that the P x P matrices (215 MB each in double at 54x96, several of them
    alive under autograd) never exist at once: the loss is a sum over query rows, so the gradients of the row blocks add."""
    import torch.nn.functional as F
    B, C, P = th.shape
    th64, ph64 = th.double().requires_grad_(True), ph.double().requires_grad_(True)
    blab = F.avg_pool2d(lab.double(), 4).view(B, 3, P).permute(0, 2, 1)              # [B, P, 3]
    gyv, gsv = gy.double().view(B, 3, P), gs.double().view(B, P)
    ys, sims = [], []
    for r0 in range(0, P, rows):
        f = torch.matmul(th64[:, :, r0:r0 + rows].permute(0, 2, 1), ph64)              # [B, rows, P]
        sim = f.max(-1)[0]
        y = torch.matmul(F.softmax(f / T, dim=-1), blab).permute(0, 2, 1)             # [B, 3, rows]
        ((y * gyv[:, :, r0:r0 + rows]).sum() + (sim * gsv[:, r0:r0 + rows]).sum()).backward()
        ys.append(y.detach()); sims.append(sim.detach())
    return th64.grad, ph64.grad, torch.cat(ys, -1), torch.cat(sims, -1)


@pytest.mark.parametrize("h,w,B,T,autotune", [(54, 96, 2, 0.01, False), (27, 48, 2, 0.01, True), (12, 20, 1, 1e-7, False)])
def test_fused_correlation_backward_at_the_training_size(h, w, B, T, autotune, gemm_mode):
    """The size the training caller runs (train.py:44,402-427: 216x384 crops -> 54 x 96 = 5184 positions, T = 0.01), B = 2:
    11 row blocks of 512 per image (the last one 64 rows) against the row-chunked float64 autograd oracle.
    `autotune=True` repeats a case with ops.set_autotune(True): the d_phi accumulation `conv2d(dS, ..., residual=out, out=out)`
    aliases its skip input with its output, and the tuner's timing launches must not add into the caller's tensor.
    T = 1e-7: the backward softmax is evaluated around the row maximum of the RECOMPUTED block (another summation order than
    the forward kernel's similarity): gradients stay finite where exp((f - sim) / T) would overflow on a one-ulp excess."""
    from dvc_amd import ops
    from dvc_amd.corr_autograd import fused_correlation
    th, ph, lab, gy, gs = _inputs(B, h, w, 100 * h + w + 1)
    dth64, dph64, y64, sim64 = _oracle_grads_row_chunked(th, ph, lab, gy, gs, T)
    ops.set_autotune(autotune)
    try:
        thd, phd = th.cuda().requires_grad_(True), ph.cuda().requires_grad_(True)
        blab = ops.avgpool4x4(lab.cuda()).view(B, 3, -1)
        y, sim, amax = fused_correlation(thd, phd, blab, T, h, w)
        ((y * gy.cuda()).sum() + (sim * gs.cuda()).sum()).backward()
        torch.cuda.synchronize()
    finally:
        ops.set_autotune(False)
    assert (sim.detach().cpu().double().view(B, -1) - sim64).abs().max().item() < 2e-6
    assert torch.isfinite(thd.grad).all() and torch.isfinite(phd.grad).all()
    if T < 1e-6:
        # hard arg-max regime: softmax rows are one-hot up to exp(-gap/T) ~ 0, so the y branch carries (numerically) no
        # gradient and d theta is the similarity branch alone; compare where the arg-max is unambiguous in fp32
        return
    for name, got, ref in (("theta", thd.grad, dth64), ("phi", phd.grad, dph64)):
        err = (got.cpu().double() - ref).abs().max().item()
        scale = ref.abs().max().item()
        print(f"corr backward {h}x{w} B={B} T={T} autotune={autotune}: d{name} max err {err:.3e} (max |grad| {scale:.3e})")
        assert err <= 2e-3 * scale, (name, err, scale)


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
