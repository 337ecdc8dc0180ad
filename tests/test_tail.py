"""Clip-driver tail (test.py:98-116).

CPU: the oracle's bilinear x2 restatement against the call the reference itself makes
(torch.nn.functional.interpolate), and properties of the WLS restatement (cv2.ximgproc is absent: parity of
the filter is unpinned, see oracle/tail_oracle.py).  GPU: the HIP kernels against the oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import tail_oracle as T


@pytest.mark.parametrize("shape", [(1, 2, 54, 96), (1, 2, 216, 384), (1, 2, 108, 192)])
def test_oracle_bilinear_matches_aten_bitwise(shape):
    """The shapes the path produces (batch 1, the two ab planes): bit-exact against the reference's own call."""
    torch.set_num_threads(1)
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(1)) * 40
    ref = (F.interpolate(x, scale_factor=2, mode="bilinear") * 1.25).numpy()
    assert np.array_equal(T.upsample_ab(x.numpy()), ref)


def test_oracle_bilinear_other_shapes_within_2ulp():
    """For other shapes ATen's TensorIterator may pick another inner loop (different fma contraction): <= 2 ulp."""
    for shape in [(1, 2, 13, 24), (2, 3, 7, 5), (1, 1, 1, 1), (2, 3, 40, 64)]:
        x = torch.randn(*shape, generator=torch.Generator().manual_seed(2)) * 40
        ref = (F.interpolate(x, scale_factor=2, mode="bilinear") * 1.25).numpy()
        got = T.upsample_ab(x.numpy())
        assert np.abs(got - ref).max() <= 2 * np.spacing(np.abs(ref).max().astype(np.float32))


@pytest.mark.parametrize("shape", [(1, 3, 432, 768), (1, 3, 216, 384), (2, 3, 41, 65), (1, 1, 2, 2)])
def test_oracle_downsample_half_matches_aten_bitwise(shape):
    torch.set_num_threads(1)
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(4)) * 40
    ref = F.interpolate(x, scale_factor=0.5, mode="bilinear").numpy()
    assert np.array_equal(T.downsample_half(x.numpy()), ref)
    assert np.array_equal(F.avg_pool2d(x, 2).numpy(), ref)


def test_oracle_guide_and_fgs_properties():
    rng = np.random.default_rng(0)
    L = (rng.random((40, 60)) * 100 - 50).astype(np.float32)
    g = T.luminance_guide_u8(L)
    assert g.dtype == np.uint8 and np.array_equal(g, ((L + 50) * 255 / 100).astype(np.uint8))
    f = rng.standard_normal((40, 60)).astype(np.float32) * 20
    u = T.fgs_filter(g, f)
    # (I + lambda A) with A a graph Laplacian: constants are fixed points, the mean is preserved, energy drops
    assert np.abs(T.fgs_filter(g, np.full_like(f, 3.25)) - 3.25).max() < 1e-3
    assert abs(float(u.mean()) - float(f.mean())) < 1e-3
    assert u.var() < f.var()
    # lambda = 0 is the identity; an edge in the guide stops the smoothing across it
    assert np.abs(T.fgs_filter(g, f, lambda_value=0.0) - f).max() < 1e-6
    gs = np.zeros((16, 32), np.uint8)
    gs[:, 16:] = 255
    step = np.where(np.arange(32) < 16, -10.0, 10.0).astype(np.float32)[None].repeat(16, 0)
    noisy = step + rng.standard_normal(step.shape).astype(np.float32)
    sm = T.fgs_filter(gs, noisy)
    assert sm[:, :16].max() < -8 and sm[:, 16:].min() > 8 and sm[:, :16].std() < 0.2


def test_oracle_lab_to_rgb8_known_values():
    L = np.array([[50.0, -50.0, 50.0, 3.24]], np.float32)      # centred: L* = 100, 0, 100, 53.24
    ab = np.array([[[0.0, 0.0, 0.0, 80.09]], [[0.0, 0.0, 0.0, 67.20]]], np.float32)
    rgb = T.lab_to_rgb8(L, ab)
    assert rgb.shape == (1, 4, 3) and rgb.dtype == np.uint8
    assert (rgb[0, 0] >= 254).all() and (rgb[0, 1] == 0).all()
    assert rgb[0, 3, 0] >= 253 and rgb[0, 3, 1] <= 2 and rgb[0, 3, 2] <= 2      # sRGB red = Lab(53.24, 80.09, 67.20)


def test_oracle_rgb_lab_round_trip():
    """rgb8 -> Lab -> rgb8 through the two (unpinned, but mutually inverse) restatements: within one level
    (the final astype(uint8) truncates), and the CIELAB anchors come out right."""
    rng = np.random.default_rng(1)
    rgb = rng.integers(0, 256, (64, 96, 3), dtype=np.uint8)
    rgb[0, 0], rgb[0, 1], rgb[0, 2] = (255, 255, 255), (0, 0, 0), (255, 0, 0)
    lab = T.rgb8_to_lab(rgb)
    assert abs(lab[0, 0, 0] - 50.0) < 1e-3 and abs(lab[1, 0, 0]) < 1e-2 and abs(lab[2, 0, 0]) < 1e-2     # white
    assert abs(lab[0, 0, 1] + 50.0) < 1e-4                                                          # black
    assert np.abs(lab[:, 0, 2] - np.array([53.2408 - 50, 80.0925, 67.2032])).max() < 2e-2              # sRGB red
    back = T.lab_to_rgb8(lab[0], lab[1:])
    assert np.abs(back.astype(np.int32) - rgb.astype(np.int32)).max() <= 1


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 2, 54, 96), (1, 2, 216, 384), (2, 3, 13, 24), (1, 1, 1, 7)])
def test_gpu_bilinear_matches_oracle_bitwise(shape):
    from dvc_amd import tail
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(3)) * 40
    got = tail.upsample_ab(x.cuda()).cpu().numpy()
    assert np.array_equal(got, T.upsample_ab(x.numpy()))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 3, 432, 768), (2, 3, 41, 65)])
def test_gpu_downsample_half_matches_oracle_bitwise(shape):
    from dvc_amd import tail
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(5)) * 40
    assert np.array_equal(tail.downsample_half(x.cuda()).cpu().numpy(), T.downsample_half(x.numpy()))


@pytest.mark.gpu
@pytest.mark.parametrize("hw", [(40, 60), (108, 192), (33, 70), (432, 768)])
def test_gpu_tail_matches_oracle(hw, report=print):
    from dvc_amd import tail
    H, W = hw
    g = torch.Generator().manual_seed(H)
    base = F.interpolate(torch.rand(1, 1, max(H // 8, 1), max(W // 8, 1), generator=g), (H, W), mode="bilinear")
    L = (base + 0.05 * torch.rand(1, 1, H, W, generator=g)).clamp(0, 1) * 100 - 50
    src = torch.randn(2, H, W, generator=g) * 30
    guide = tail.luminance_guide_u8(L[0, 0].cuda())
    assert np.array_equal(guide.cpu().numpy(), T.luminance_guide_u8(L[0, 0].numpy()))
    got = tail.fgs_filter(guide, src.cuda()).cpu().numpy()
    ref = np.stack([T.fgs_filter(guide.cpu().numpy(), src[k].numpy()) for k in range(2)])
    err = np.abs(got - ref).max()
    print(f"fgs {H}x{W}: max abs err vs oracle {err:.2e} (values ~{np.abs(ref).max():.1f})")
    import os
    rp = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "test_report.txt")
    os.makedirs(os.path.dirname(rp), exist_ok=True)
    open(rp, "a").write(f"fgs {H}x{W}: kernel vs oracle max abs err {err:.2e} (values ~{np.abs(ref).max():.1f})\n")
    assert err < 3e-4          # two float32 evaluations of the same recurrences on values of ~100 (measured ~6e-5)
    rgb = tail.lab_to_rgb8(L[0, 0].cuda(), torch.from_numpy(ref).cuda()).cpu().numpy()
    ref8 = T.lab_to_rgb8(L[0, 0].numpy(), ref)
    d = np.abs(rgb.astype(np.int32) - ref8.astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() < 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("hw,num_iter", [((1, 1), 3), ((1, 37), 3), ((5, 1), 2), ((2, 64), 3), ((3, 65), 1), ((64, 129), 3), ((7, 1024), 2),
                                         ((6, 1100), 2), ((1030, 9), 3), ((40, 60), 8)])
def test_gpu_fgs_edge_geometries(hw, num_iter):
    """The scan solver's edges (r05): single rows / columns / pixels, line lengths on and around multiples of the wave size, the
    longest line it takes (1024), and beyond it in either direction — where dvc_fgs_filter falls back to the thread-per-line
    sweeps — and the maximum number of iterations; several frames with their own guides in one call.  Against the oracle."""
    from dvc_amd import tail
    H, W = hw
    g = torch.Generator().manual_seed(1000 * H + W)
    G = 2
    guide = torch.randint(0, 256, (G, H, W), generator=g, dtype=torch.uint8)
    guide[1] = (torch.arange(H * W).view(H, W) % 7 * 30).to(torch.uint8)          # (a structured guide next to the noisy one)
    src = torch.randn(G, 2, H, W, generator=g) * 30
    got = tail.fgs_filter(guide.cuda(), src.cuda(), num_iter=num_iter).cpu().numpy()
    for f in range(G):
        for k in range(2):
            ref = T.fgs_filter(guide[f].numpy(), src[f, k].numpy(), num_iter=num_iter)
            err = np.abs(got[f, k] - ref).max()
            assert err < 1e-3, (hw, num_iter, f, k, err)          # white-noise guide: the worst conditioned systems (cf. frame_tail test)
    one = tail.fgs_filter(guide[0].cuda(), src[0].cuda(), num_iter=num_iter).cpu().numpy()
    assert np.array_equal(one, got[0])                            # a frame's result does not depend on the batch it came in
    # a lambda so large that the windowed coefficient recurrence would need more than its maximum warm-up (the coefficients then
    # come from the thread-per-line chains, transposed for the scan solves), and a tiny one (warm-up of one block)
    # (r06: with the workspace the library asks for, that case keeps ONE coefficient copy and runs the thread-per-line solver;
    # `second_coeff_copy` gives it room for the transposed copy the scan solver reads)
    for lam, second in ((20000.0, False), (20000.0, True), (0.05, False)):
        got_l = tail.fgs_filter(guide[1].cuda(), src[1].cuda(), lambda_value=lam, num_iter=num_iter, second_coeff_copy=second).cpu().numpy()
        for k in range(2):
            ref = T.fgs_filter(guide[1].numpy(), src[1, k].numpy(), lambda_value=lam, num_iter=num_iter)
            err = np.abs(got_l[k] - ref).max()
            assert err < (2e-2 if lam > 1e3 else 1e-3), (hw, num_iter, lam, second, k, err)      # (lambda = 2e4: systems conditioned ~1e5)
    # lambda = 1e7: the largest decade at which the fp32 recurrences of the algorithm are still finite (the oracle — and the
    # reference's fp32 filter — divide by zero from ~1e8 on); far beyond the windowed kernel's warm-up budget.  At 1e18 the
    # warm-up length is not even representable (rho -> 1, log(rho) -> 0: ADVICE r05): the library must take the fall-back,
    # i.e. do what the oracle does (NaN / inf included), not run the windowed kernel without a warm-up
    for second in (False, True):
        got_l = tail.fgs_filter(guide[1].cuda(), src[1].cuda(), lambda_value=1e7, num_iter=num_iter, second_coeff_copy=second).cpu().numpy()
        with np.errstate(all="ignore"):
            ref = np.stack([T.fgs_filter(guide[1].numpy(), src[1, k].numpy(), lambda_value=1e7, num_iter=num_iter) for k in range(2)])
        if np.isfinite(ref).all():
            scale = np.abs(src[1].numpy()).max()
            assert np.isfinite(got_l).all() and np.abs(got_l - ref).max() < 0.2 * scale, (hw, num_iter, second, np.abs(got_l - ref).max())
        big = tail.fgs_filter(guide[1].cuda(), src[1].cuda(), lambda_value=1e18, num_iter=num_iter, second_coeff_copy=second).cpu().numpy()
        with np.errstate(all="ignore"):
            ref = np.stack([T.fgs_filter(guide[1].numpy(), src[1, k].numpy(), lambda_value=1e18, num_iter=num_iter) for k in range(2)])
        assert big.shape == ref.shape
        if np.isfinite(ref).all():
            assert np.isfinite(big).all()


@pytest.mark.gpu
def test_gpu_frame_tail_end_to_end():
    from dvc_amd import tail
    H, W = 54, 96
    g = torch.Generator().manual_seed(9)
    L = torch.rand(1, 1, 2 * H, 2 * W, generator=g) * 100 - 50
    lab_large = torch.cat((L, torch.zeros(1, 2, 2 * H, 2 * W)), 1)
    ab = torch.randn(1, 2, H, W, generator=g) * 25
    rgb, cur = tail.frame_tail(lab_large.cuda(), ab.cuda())
    rgb_o, cur_o = T.frame_tail(L.numpy(), ab.numpy())
    assert np.abs(cur.cpu().numpy() - cur_o).max() < 1e-3      # white-noise guide: the worst conditioned systems (measured 3.1e-4)
    d = np.abs(rgb.cpu().numpy().astype(np.int32) - rgb_o.astype(np.int32))
    assert d.max() <= 1
    rgb2, cur2 = tail.frame_tail(lab_large.cuda(), ab.cuda(), wls_filter_on=False)
    assert np.array_equal(cur2.cpu().numpy(), T.upsample_ab(ab.numpy()))
    with pytest.raises(RuntimeError):
        tail.frame_tail(lab_large, ab)          # CPU tensors: no fallback
    # several frames in one call == frame by frame
    labs = [lab_large.cuda(), (lab_large * 0.5).cuda(), (lab_large * -0.3).cuda()]
    abs_ = [ab.cuda(), (ab * 0.7).cuda(), (ab + 3).cuda()]
    many, _ = tail.frames_tail(labs, abs_)
    for lg, a, r in zip(labs, abs_, many):
        one, _ = tail.frame_tail(lg, a)
        assert torch.equal(one, r)


@pytest.mark.gpu
def test_gpu_clip_rgb_equals_stagewise():
    """ClipColorizer.clip_rgb (x0.5 -> pipelined recurrence -> tail on its own stream) == the same stages called
    one after the other on one stream."""
    import contextlib
    import io
    from dvc_amd import synth, tail
    from dvc_amd.frame import ClipColorizer
    from models.ColorVidNet import ColorVidNet
    from models.NonlocalNet import VGG19_pytorch, WarpNet
    dev = torch.device("cuda")
    with contextlib.redirect_stdout(io.StringIO()):
        nets = (VGG19_pytorch(), WarpNet(1), ColorVidNet(7))
    for m, sd in zip(nets, (synth.vgg19_state_dict(0), synth.warpnet_state_dict(0), synth.colorvidnet_state_dict(0))):
        m.load_state_dict(sd)
        m.eval().to(dev)
    H, W = 96, 160                                  # full resolution; the networks run at 48 x 80
    large = [synth.synth_lab(synth.FRAME_SEED0 + i, H, W).to(dev) for i in range(5)]
    cc = ClipColorizer(*nets, temperature=1e-10)
    cc.set_exemplar(tail.downsample_half(synth.synth_lab(synth.EXEMPLAR_SEED, H, W).to(dev)))
    got = cc.clip_rgb(large)
    torch.cuda.synchronize()
    abs_ = cc.clip([tail.downsample_half(f) for f in large], lookahead=0)
    for t, (f, ab) in enumerate(zip(large, abs_)):
        rgb, _ = tail.frame_tail(f, ab)
        assert got[t].shape == (H, W, 3) and got[t].dtype == torch.uint8
        assert torch.equal(got[t], rgb), t



@pytest.mark.gpu
def test_gpu_lab_to_rgb8_two_tier_is_the_float64_chain_byte_for_byte():
    """dvc_lab2rgb_u8 (r06): a float32 evaluation decides the bytes that are further than its error bound from a rounding
    boundary, the rest of the pixels go through the float64 chain.  Byte-for-byte the oracle's float64 result on 3.1 million
    pixels: in-gamut colours, far out-of-gamut ones (both clipping points, the linear pieces of both curves), a dense grey
    ramp (every channel crosses all 255 boundaries), and exact zeros."""
    from dvc_amd import tail
    rng = np.random.default_rng(11)
    H, W = 1024, 1024
    L = rng.uniform(-50, 50, (H, W)).astype(np.float32)
    ab = rng.uniform(-110, 110, (2, H, W)).astype(np.float32)
    L2 = rng.uniform(-80, 90, (H, W)).astype(np.float32)            # beyond the gamut: negative and > 1 linear values
    ab2 = (rng.standard_normal((2, H, W)) * 150).astype(np.float32)
    L3 = np.linspace(-50, 50, H * W, dtype=np.float32).reshape(H, W)
    ab3 = np.zeros((2, H, W), np.float32)
    ab3[:, H // 2:] = rng.uniform(-3, 3, (2, H - H // 2, W)).astype(np.float32)
    for Lc, abc in ((L, ab), (L2, ab2), (L3, ab3)):
        got = tail.lab_to_rgb8(torch.from_numpy(Lc).cuda(), torch.from_numpy(abc).cuda()).cpu().numpy()
        ref = T.lab_to_rgb8(Lc, abc)
        assert got.shape == ref.shape == (H, W, 3)
        assert np.array_equal(got, ref), int((got != ref).sum())
    # ragged size (not a multiple of the 1024 pixels a workgroup takes)
    got = tail.lab_to_rgb8(torch.from_numpy(L[:37, :53].copy()).cuda(), torch.from_numpy(ab[:, :37, :53].copy()).cuda()).cpu().numpy()
    assert np.array_equal(got, T.lab_to_rgb8(L[:37, :53], ab[:, :37, :53]))


@pytest.mark.gpu
def test_gpu_rgb8_to_lab_is_the_float64_chain_rounded_once():
    """dvc_rgb8_to_lab: skimage's float64 chain, rounded to float32 at the end (r06: the 256 possible sRGB -> linear values come from
    a per-workgroup table of the same expression).  Against the oracle's numpy float64 chain on 2 M random colours + every grey
    level: the same float32 in all but a vanishing fraction (two float64 libm implementations can differ in the last bit, which
    survives the rounding to float32 once in ~1e8), never more than one float32 step apart."""
    from dvc_amd import tail
    rng = np.random.default_rng(5)
    rgb = rng.integers(0, 256, (1000, 2000, 3), dtype=np.uint8)
    rgb[:, :256] = np.arange(256, dtype=np.uint8)[None, :, None]
    got = tail.rgb8_to_lab(torch.from_numpy(rgb).cuda())[0].cpu().numpy()
    ref = T.rgb8_to_lab(rgb)
    same = got == ref
    assert same.mean() > 1.0 - 1e-5, 1.0 - same.mean()
    assert np.abs(got - ref).max() <= 2e-5          # (one float32 step at |value| <= 128 is 7.6e-6 .. 1.5e-5)


@pytest.mark.gpu
def test_gpu_rgb8_to_lab_matches_oracle_and_round_trips():
    from dvc_amd import tail
    rng = np.random.default_rng(2)
    rgb = rng.integers(0, 256, (70, 130, 3), dtype=np.uint8)
    lab = tail.rgb8_to_lab(torch.from_numpy(rgb).cuda())
    ref = T.rgb8_to_lab(rgb)
    assert tuple(lab.shape) == (1, 3, 70, 130)
    assert np.abs(lab[0].cpu().numpy() - ref).max() < 1e-4
    back = tail.lab_to_rgb8(lab[0, 0].contiguous(), lab[0, 1:].contiguous()).cpu().numpy()
    assert np.abs(back.astype(np.int32) - rgb.astype(np.int32)).max() <= 1


# ---------------------------------------------------------------------------------------------------------------
# Pinning what can be pinned without OpenCV / skimage
def _fgs_fp64_direct(guide_u8, src, lambda_value=500.0, sigma_color=4.0, num_iter=3, lambda_attenuation=0.25):
    """Min et al. 2014, Alg. 1 with every 1-D system (I + lambda_t A) u = f solved DIRECTLY in float64 by LAPACK's
    banded solver (scipy.linalg.solve_banded) — no Thomas recurrence, no float32: the published linear systems."""
    from scipy.linalg import solve_banded
    g = np.asarray(guide_u8).astype(np.int64)
    wh = np.exp(-np.abs(g[:, 1:] - g[:, :-1]) / float(sigma_color))
    wv = np.exp(-np.abs(g[1:, :] - g[:-1, :]) / float(sigma_color))
    u = np.asarray(src, dtype=np.float64).copy()
    lam = 1.5 * lambda_value * 4.0 ** (num_iter - 1) / (4.0 ** num_iter - 1.0)

    def solve_lines(f, w):      # f [L, n]; w [L, n-1]: one tridiagonal system per line
        out = np.empty_like(f)
        n = f.shape[1]
        for i in range(f.shape[0]):
            off = -lam * w[i]
            ab = np.zeros((3, n))
            ab[0, 1:] = off
            ab[2, :-1] = off
            ab[1] = 1.0
            ab[1, :-1] -= off
            ab[1, 1:] -= off
            out[i] = solve_banded((1, 1), ab, f[i])
        return out

    for _ in range(num_iter):
        u = solve_lines(u, wh)
        u = solve_lines(u.T.copy(), wv.T.copy()).T.copy()
        lam *= lambda_attenuation
    return u


def test_fgs_oracle_solver_pinned_by_direct_fp64_solve():
    """The oracle's float32 Thomas recurrences against a direct float64 banded solve of the same linear systems:
    pins the SOLVER (what remains unpinned without cv2.ximgproc is only that OpenCV's filter is Alg. 1 with these
    weights and lambda_t — the published definition)."""
    rng = np.random.default_rng(3)
    for (H, W) in [(24, 40), (37, 53)]:
        L = (rng.random((H, W)) * 100 - 50).astype(np.float32)
        L[:, W // 2:] += 20                                     # a luminance edge
        g = T.luminance_guide_u8(np.clip(L, -50, 50))
        f = (rng.standard_normal((H, W)) * 20).astype(np.float32)
        for lam, sig in [(500.0, 4.0), (50.0, 10.0)]:
            u32 = T.fgs_filter(g, f, lam, sig)
            u64 = _fgs_fp64_direct(g, f, lam, sig)
            err = np.abs(u32 - u64).max()
            assert err < 2e-4 * max(1.0, np.abs(u64).max()), (H, W, lam, err)


def _lab_grid():
    Ls = np.linspace(0.0, 100.0, 21)
    As = np.linspace(-100.0, 100.0, 21)
    L, a, b = np.meshgrid(Ls, As, As, indexing="ij")
    return L.reshape(1, -1), a.reshape(1, -1), b.reshape(1, -1)


def test_lab2rgb_oracle_pinned_by_reference_tensor_lab2rgb():
    """oracle.tail_oracle.lab_to_rgb8 (restating skimage.color.lab2rgb, absent here) against the reference's OWN
    Lab->RGB, utils/util.py:379-414 `tensor_lab2rgb` (restated in oracle/dvc_oracle.py and pinned bit-exact to the
    reference by oracle/pin_reference.py), on a dense Lab grid: the two differ only by the digits of the XYZ->RGB
    matrix, i.e. never by more than one 8-bit level, and on < 0.5 % of the grid."""
    from oracle import dvc_oracle as O
    L, a, b = _lab_grid()
    lab = torch.from_numpy(np.stack([L, a, b])[None].astype(np.float32))            # [1,3,1,K], L in [0,100]
    with torch.no_grad():
        ref = (O.tensor_lab2rgb(lab.double())[0, :, 0].numpy().T * 255.0)             # [K,3] in [0,255]
    got = T.lab_to_rgb8((L - 50.0).astype(np.float32), np.stack([a, b]).astype(np.float32))[0]   # [K,3] uint8
    ref8 = ref.astype(np.uint8)
    diff = np.abs(got.astype(np.int64) - ref8.astype(np.int64))
    assert diff.max() <= 1
    assert (diff > 0).mean() < 5e-3
    # where they differ the reference value sits on an integer boundary (truncation decides)
    frac = np.abs(ref - np.round(ref))
    assert (frac[diff > 0] < 2e-2).all()


def test_rgb2lab_oracle_inverts_the_reference_lab2rgb():
    """oracle.tail_oracle.rgb8_to_lab (restating skimage.color.rgb2lab) must be the inverse of the reference-held
    Lab->RGB (`tensor_lab2rgb`): 8-bit RGB -> Lab -> RGB returns the same 8-bit colour on a 17^3 grid."""
    from oracle import dvc_oracle as O
    v = np.linspace(0, 255, 17).round().astype(np.uint8)
    r, g, b = np.meshgrid(v, v, v, indexing="ij")
    rgb = np.stack([r, g, b], axis=-1).reshape(1, -1, 3)
    lab = T.rgb8_to_lab(rgb)                                                        # [3,1,K], L centred
    lab_t = torch.from_numpy(lab)[None].double()
    lab_t[:, 0] += 50.0
    with torch.no_grad():
        back = O.tensor_lab2rgb(lab_t)[0, :, 0].numpy().T * 255.0
    assert np.abs(back - rgb[0].astype(np.float64)).max() < 0.02
