"""Frame ingest (test.py:44-46, utils/util_distortion.py:217-258).

CPU: the index-by-index restatement of skimage's anti-aliased resize equals the SciPy calls skimage makes (skimage
itself is absent: parity unpinned, see oracle/ingest_oracle.py), and CenterPad's three branches.  GPU: kernels vs
the oracle (<= 1 level: astype(uint8) truncates float64 values that sit on integers)."""
import numpy as np
import pytest
import torch

from oracle import ingest_oracle as G
from oracle import tail_oracle as T

CASES = [((37, 53), (16, 24)), ((108, 200), (54, 96)), ((120, 160), (54, 96)), ((54, 96), (54, 96)),
         ((30, 40), (54, 96)), ((270, 480), (216, 384)), ((90, 250), (54, 96)), ((200, 120), (54, 96))]


@pytest.mark.parametrize("src,dst", CASES)
def test_oracle_resize_restatement_equals_scipy(src, dst):
    rng = np.random.default_rng(src[0] * 1000 + src[1])
    img = rng.integers(0, 256, src + (3,), dtype=np.uint8)
    a = G.center_pad(img, dst)
    b = G.center_pad(img, dst, resize=G.resize_numpy)
    assert a.shape == dst + (3,) and a.dtype == np.uint8
    d = np.abs(a.astype(np.int32) - b.astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() < 0.02


def test_oracle_center_pad_branches_and_properties():
    rng = np.random.default_rng(3)
    same = rng.integers(0, 256, (54, 96, 3), dtype=np.uint8)
    assert np.array_equal(G.center_pad(same, (54, 96)), same)                  # same size: untouched
    flat = np.full((200, 120, 3), 77, np.uint8)
    out = G.center_pad(flat, (54, 96))
    assert np.abs(out.astype(np.int32) - 77).max() <= 1                       # (76.99999 truncates to 76)
    # a vertical edge stays a vertical edge at the scaled position; anti-aliasing makes it a ramp, not a step
    edge = np.zeros((108, 192, 3), np.uint8)
    edge[:, 96:] = 200
    o = G.center_pad(edge, (54, 96)).astype(np.int32)
    assert (o[:, :46] == 0).all() and (o[:, 50:] >= 199).all() and 0 < o[10, 47, 0] < 200
    lab = G.frame_ingest(edge, (54, 96))
    assert lab.shape == (3, 54, 96) and abs(lab[0, 0, 0] + 50.0) < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("src,dst", CASES + [((1080, 1920), (432, 768)), ((480, 640), (216, 384))])
def test_gpu_center_pad_matches_oracle(src, dst):
    from dvc_amd import tail
    rng = np.random.default_rng(src[0] + 7 * src[1])
    # smooth image + noise (so that the test is not only about white noise)
    yy, xx = np.mgrid[0:src[0], 0:src[1]]
    img = (127 + 100 * np.sin(yy / 17.0)[..., None] * np.cos(xx / 23.0)[..., None]
           + rng.normal(0, 12, src + (3,))).clip(0, 255).astype(np.uint8)
    got = tail.center_pad(torch.from_numpy(img).cuda(), dst).cpu().numpy()
    ref = G.center_pad(img, dst)
    d = np.abs(got.astype(np.int32) - ref.astype(np.int32))
    print(f"center_pad {src}->{dst}: max level diff {d.max()}, fraction differing {(d > 0).mean():.2e}")
    assert got.shape == ref.shape and d.max() <= 1 and (d > 0).mean() < 0.02
    lab = tail.frame_ingest(torch.from_numpy(img).cuda(), dst)
    assert tuple(lab.shape) == (1, 3) + dst
    assert np.abs(lab[0].cpu().numpy() - T.rgb8_to_lab(got)).max() < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("src,dst", [((1080, 1920), (432, 768)), ((480, 640), (216, 384)), ((37, 53), (16, 24)), ((108, 200), (54, 96)),
                                     ((120, 160), (54, 96)), ((270, 480), (216, 384)), ((90, 250), (54, 96)), ((200, 120), (54, 96)),
                                     ((720, 1280), (216, 384)), ((433, 770), (216, 384)), ((216, 384), (216, 384)), ((700, 1245), (216, 384)),
                                     ((30, 40), (54, 96)), ((1000, 1700), (216, 384)), ((100, 150), (40, 60)), ((20, 30), (8, 12)), ((65, 97), (26, 39)),
                                     ((17, 300), (16, 282))])
def test_gpu_center_pad_fused_kernel_is_the_three_pass_path_byte_for_byte(src, dst):
    """dvc_center_pad without a workspace (r06): one launch that filters only what the samples read, against the three
    full-frame float64 passes on the same frame — the same bytes, on every branch of CenterPad (crop rows / crop columns / same
    ratio / same size), Gaussian radii 0 .. 4, ragged tile edges; factors beyond 3.25 and up-scaling keep the three passes."""
    from dvc_amd import _lib, tail
    rng = np.random.default_rng(src[0] + 3 * src[1])
    yy, xx = np.mgrid[0:src[0], 0:src[1]]
    img = (127 + 100 * np.sin(yy / 11.0)[..., None] * np.cos(xx / 19.0)[..., None]
           + rng.normal(0, 25, src + (3,))).clip(0, 255).astype(np.uint8)
    x = torch.from_numpy(img).cuda()
    a = tail.center_pad(x, dst)
    b = tail.center_pad(x, dst, three_pass=True)
    assert torch.equal(a, b), int((a != b).sum())
    # which cases the fused kernel takes: the reference's own sizes do; x4.6, up-scaling and unequal radii do not
    fused = {k: bool(_lib.load().dvc_center_pad_is_fused(*k)) for k in ((1080, 1920, 432, 768), (480, 640, 216, 384), (216, 384, 216, 384),
                                                                       (1000, 1700, 216, 384), (30, 40, 54, 96), (37, 53, 16, 24))}
    assert list(fused.values()) == [True, True, True, False, False, False], fused


@pytest.mark.gpu
def test_gpu_center_pad_errors_are_loud():
    from dvc_amd import tail
    img = torch.zeros(64, 64, 3, dtype=torch.uint8)
    with pytest.raises(RuntimeError):
        tail.center_pad(img, (54, 96))                                      # CPU tensor
    with pytest.raises(RuntimeError):
        tail.center_pad(torch.zeros(2200, 2200, 3, dtype=torch.uint8).cuda(), (54, 96))   # x23 down-scaling: radius > 40


@pytest.mark.gpu
def test_gpu_colorize_video_equals_stagewise():
    """8-bit RGB frames of another size in -> 8-bit RGB out, one call == ingest, x0.5, recurrence, tail by hand."""
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
    rng = np.random.default_rng(5)
    yy, xx = np.mgrid[0:150, 0:200]
    frames = [torch.from_numpy((127 + 90 * np.sin((yy + 3 * t) / 19.0)[..., None] * np.cos(xx / 13.0)[..., None]
                                + rng.normal(0, 10, (150, 200, 3))).clip(0, 255).astype(np.uint8)).to(dev)
              for t in range(4)]
    ref = frames[2].flip(1).contiguous()
    size = (96, 160)
    cc = ClipColorizer(*nets, temperature=1e-10)
    got = cc.colorize_video(frames, ref, image_size=size)
    torch.cuda.synchronize()
    large = [tail.frame_ingest(f, size) for f in frames]
    cc.set_exemplar(tail.downsample_half(tail.frame_ingest(ref, size)))
    abs_ = cc.clip([tail.downsample_half(f) for f in large], lookahead=0)
    for t in range(4):
        rgb, _ = tail.frame_tail(large[t], abs_[t])
        assert got[t].shape == size + (3,) and torch.equal(got[t], rgb), t
