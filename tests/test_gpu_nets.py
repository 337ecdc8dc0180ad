"""GPU parity, network level: the drop-in modules (models.NonlocalNet / models.ColorVidNet /
models.FrameColor) vs the CPU oracle and vs the golden vectors recorded from the unmodified reference.

Tolerance policy (SURVEY.md §7 hard part 1, §8c): stages are compared on IDENTICAL stage inputs with
tight tolerances; end-to-end ab is compared against an fp64 run of the oracle next to the oracle's own
fp32-vs-fp64 error (the reference's reproducibility floor), and against the reference golden at the
reference's measured thread-count noise level.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

REPORT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "test_report.txt")


def report(line):
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    with open(REPORT, "a") as f:
        f.write(line + "\n")


@pytest.fixture(scope="module")
def nets(weights):
    import contextlib
    import io
    from models.ColorVidNet import ColorVidNet
    from models.NonlocalNet import VGG19_pytorch, WarpNet
    sd_v, sd_w, sd_c = weights
    with contextlib.redirect_stdout(io.StringIO()):
        vgg, warp, col = VGG19_pytorch(), WarpNet(1), ColorVidNet(7)
    vgg.load_state_dict(sd_v)
    warp.load_state_dict(sd_w)
    col.load_state_dict(sd_c)
    for m in (vgg, warp, col):
        m.eval()
        m.cuda()
    return vgg, warp, col


def rel(got, ref):
    ref = ref.double()
    return ((got.double().cpu() - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize("H,W", [(48, 80), (216, 384)])
def test_vgg19_all_keys(nets, weights, H, W):
    from dvc_amd import arch, synth
    from oracle import dvc_oracle as O
    vgg = nets[0]
    x = O.gray2rgb_batch(synth.synth_lab(5, H, W)[:, 0:1])
    keys = list(arch.VGG_KEYS)
    with torch.no_grad():
        ref = O.vgg19_forward(O.to_dtype(weights[0], torch.float64), x.double(), keys)
    got = vgg(x.cuda(), keys, preprocess=True)
    for k, g, r in zip(keys, got, ref):
        e = rel(g, r)
        report(f"vgg {H}x{W} {k}: rel_err={e:.2e} shape={tuple(g.shape)}")
        assert tuple(g.shape) == tuple(r.shape)
        assert e < 1e-4, (k, e)      # 16 stacked fp32 convs vs fp64 truth
    # preprocess=False path and avg-pool variant
    got2 = vgg(x.cuda(), ["r22"], preprocess=False)[0]
    ref2 = O.vgg19_forward(O.to_dtype(weights[0], torch.float64), x.double(), ["r22"], preprocess=False)[0]
    assert rel(got2, ref2) < 1e-4


@pytest.mark.parametrize("H,W", [(216, 384), (54, 90)])
def test_vgg19_pool_fused_into_the_convolution(nets, H, W):
    """relu1_2 / relu2_2 / relu3_4 / relu4_4 -> pool come out of the convolution's own launch (ops.conv2d_winograd_pool): every
    requested activation is bit-identical to the convolution -> maxpool2x2 sequence, whether the pre-pool tensor is itself
    requested (r12, r22: the front end's taps; r34) or not."""
    from dvc_amd import ops, synth
    from oracle import dvc_oracle as O
    vgg = nets[0]
    x = O.gray2rgb_batch(synth.synth_lab(7, H, W)[:, 0:1]).cuda()
    for keys in (["r12", "r22", "r32", "r42", "r52"], ["r34", "p3", "r44", "p4"], ["p1"]):
        assert ops.pool_fusion()
        fused = vgg(x, keys)
        try:
            ops.set_pool_fusion(False)
            plain = vgg(x, keys)
        finally:
            ops.set_pool_fusion(True)
        for k, a, b in zip(keys, fused, plain):
            assert torch.equal(a, b), k


def test_vgg_avgpool_variant(weights):
    from models.NonlocalNet import VGG19_pytorch
    from oracle import dvc_oracle as O
    v = VGG19_pytorch(pool="avg")
    v.load_state_dict(weights[0])
    v.cuda()
    x = torch.rand(1, 3, 32, 48)
    ref = O.vgg19_forward(O.to_dtype(weights[0], torch.float64), x.double(), ["r32"], pool="avg")[0]
    assert rel(v(x.cuda(), ["r32"])[0], ref) < 1e-4


def _norm_feats(sd_v, lab, dtype=torch.float32):
    from oracle import dvc_oracle as O
    x = O.gray2rgb_batch(lab[:, 0:1]).to(dtype)
    f = O.vgg19_forward(O.to_dtype(sd_v, dtype), x, O.VGG_OUT)
    return [O.feature_normalize(t) for t in f[1:]]


@pytest.mark.parametrize("H,W", [(48, 80), (40, 64), (216, 384)])
def test_warpnet_stages_identical_inputs(nets, weights, H, W):
    """heads+trunk, theta/phi projection and the fused correlation, each on identical inputs."""
    from dvc_amd import synth
    from oracle import dvc_oracle as O
    warp = nets[1]
    sd_v, sd_w, _ = weights
    with torch.no_grad():
        nA = _norm_feats(sd_v, synth.synth_lab(1000, H, W))
        nB = _norm_feats(sd_v, synth.synth_lab(2, H, W))
        sd64 = O.to_dtype(sd_w, torch.float64)
        fA64 = O.warp_features(sd64, *[t.double() for t in nA])
    fA = warp.features(*[t.cuda() for t in nA])
    e = rel(fA, fA64)
    report(f"warp.features {H}x{W}: rel_err_vs_fp64={e:.2e}")
    assert e < 2e-4
    with torch.no_grad():
        th64 = O.corr_project(sd64, "theta", fA.double().cpu())
    th = warp.project("theta", fA)
    e = (th.double().cpu() - th64).abs().max().item()
    report(f"warp.project {H}x{W}: abs_err={e:.2e}")
    assert e < 2e-6          # unit-norm columns
    # full forward vs fp32 oracle (argmax flips possible only on near-ties)
    IB = synth.synth_lab(2, H, W)
    for T in (1e-10, 0.01):
        taps = {}
        with torch.no_grad():
            y_ref, sim_ref = O.warpnet_forward(sd_w, IB, *nA, *nB, temperature=T, taps=taps)
        y, sim, tp = warp(IB.cuda(), *[t.cuda() for t in nA], *[t.cuda() for t in nB], temperature=T,
                          return_taps=True)
        gap = taps["top2"][0, :, 0] - taps["top2"][0, :, 1]
        safe = gap > 1e-4
        agree = (tp["argmax"][0].cpu().long() == taps["argmax"][0])
        sim_err = (sim.cpu() - sim_ref).abs().max().item()
        ys = (tp["y_small"].cpu() - taps["y_small"]).abs().view(3, -1)
        report(f"warp.forward {H}x{W} T={T}: sim_err={sim_err:.2e} argmax_agree_all={agree.float().mean():.4f} "
               f"agree_safe={agree[safe].float().mean():.4f} y_err_safe={ys[:, safe].max():.2e} y_err_all={ys.max():.2e}")
        assert y.shape == y_ref.shape and sim.shape == sim_ref.shape
        assert sim_err < 1e-4
        assert agree[safe].float().mean().item() > 0.995


@pytest.mark.parametrize("H,W,N", [(216, 384, 1), (48, 80, 3), (37, 53, 2)])
def test_vgg_gray_input_folded_into_conv1_1_is_bit_identical(nets, H, W, N):
    """r06 (DVC_CONV_GRAY_INPUT): FrameColor.py:8-10's gray2rgb_batch(IA_l) -> vggnet(...) with the replication folded into
    conv1_1's load: every tap bit-identical to the two-step form, for a contiguous luminance tensor and for the channel-0 slice
    of a Lab batch (what warp_color hands over: images 3*H*W apart); and warp_color gives the same warped colours either way."""
    from dvc_amd import ops, synth
    from dvc_amd.frame import VGG_OUT, warp_color
    from utils.util import gray2rgb_batch
    vgg, warp, col = nets
    lab = torch.cat([synth.synth_lab(1000 + i, H, W) for i in range(N)]).cuda()
    for l in (lab[:, 0:1], lab[:, 0:1].contiguous()):
        want = vgg(gray2rgb_batch(l), VGG_OUT, preprocess=True)
        got = vgg.forward_gray(l, VGG_OUT)
        for a, b in zip(got, want):
            assert torch.equal(a, b)
    if H % 16 == 0 and W % 16 == 0:
        IB = torch.cat([synth.synth_lab(2 + i, H, W) for i in range(N)]).cuda()
        fB = vgg(ops.lab2rgb(IB, l_offset=50.0), VGG_OUT, preprocess=True)
        outs = []
        for flag in (True, False):
            ops.set_gray_fusion(flag)
            try:
                w_, s_, fA = warp_color(lab[:, 0:1], IB, fB, vgg, warp, col, 0, temperature=1e-10)
            finally:
                ops.set_gray_fusion(True)
            outs.append((w_, s_, fA))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        for a, b in zip(outs[0][2], outs[1][2]):
            assert torch.equal(a, b)


@pytest.mark.parametrize("H,W,N", [(216, 384, 1), (48, 80, 2), (40, 64, 1), (432, 768, 1)])
def test_warpnet_heads_grouped_launches_are_bit_identical(nets, weights, H, W, N):
    """r06: the four heads advance stage by stage, each stage ONE launch over the independent layers
    (dvc_conv2d_winograd_group / dvc_instnorm_apply_group).  Every item keeps the plan and the kernel body it has as a launch
    of its own, so heads + trunk must come out bit-identical with DVC_GROUP_HEADS on and off — at the path's size, a batch, the
    replicate-pad geometry (40x64) and configs[3]'s size; and the grouped form really is fewer launches."""
    from dvc_amd import _lib, ops, synth
    warp = nets[1]
    sd_v = weights[0]
    with torch.no_grad():
        nA = [torch.cat([t] + [_norm_feats(sd_v, synth.synth_lab(1000 + k, H, W))[j] for k in range(1, N)])
              for j, t in enumerate(_norm_feats(sd_v, synth.synth_lab(1000, H, W)))]
    nA = [t.cuda() for t in nA]
    lib = _lib.load()
    calls = {}

    class Spy:
        def __init__(self, name):
            self.fn, self.name = getattr(lib, name), name

        def __call__(self, *a):
            calls[self.name] = calls.get(self.name, 0) + 1
            return self.fn(*a)
    names = ["dvc_conv2d_winograd_group", "dvc_instnorm_apply_group", "dvc_conv2d_winograd", "dvc_instnorm_apply",
             "dvc_instnorm_apply_partials", "dvc_conv2d"]

    def run(flag):
        calls.clear()
        ops.set_group_heads(flag)
        spies = {}
        try:
            for nme in names:
                spies[nme] = getattr(lib, nme)
                setattr(lib, nme, Spy(nme))
            out = warp.features(*nA)
            torch.cuda.synchronize()
        finally:
            for nme, fn in spies.items():
                setattr(lib, nme, fn)
            ops.set_group_heads(True)
        return out, dict(calls)
    warp.prepare()
    grouped, cg = run(True)
    single, cs = run(False)
    assert torch.equal(grouped, single), (grouped - single).abs().max().item()
    assert torch.isfinite(grouped).all()
    n_g, n_s = sum(cg.values()), sum(cs.values())
    report(f"warp.features {H}x{W} N={N}: host calls grouped {cg} = {n_g}, per layer {cs} = {n_s}")
    assert cg.get("dvc_instnorm_apply_group", 0) == 2 and n_g < n_s
    if H >= 216:        # (smaller maps: the heads' layers are below the Winograd rule's 13x24 and keep their direct launches)
        assert cg.get("dvc_conv2d_winograd_group", 0) == 2 and n_g <= n_s - 9


@pytest.mark.parametrize("seed", [1000, 1001, 1002, 1003])
def test_correlation_on_real_features_self_consistent(nets, weights, seed):
    """For several frames: (a) the kernel's argmax / sim / gathered colour agree with an fp64 evaluation
    of the SAME theta/phi it consumed (isolates the correlation kernel), and (b) theta/phi/argmax agree
    with the oracle run end-to-end from the same Lab frame (isolates everything upstream)."""
    from dvc_amd import ops, synth
    from dvc_amd.frame import VGG_OUT
    from oracle import dvc_oracle as O
    from utils.util import feature_normalize, gray2rgb_batch
    vgg, warp, _ = nets
    H, W, T = 216, 384, 1e-10
    IB = synth.synth_lab(synth.EXEMPLAR_SEED, H, W)
    fr = synth.synth_lab(seed, H, W)
    fB = vgg(ops.lab2rgb(IB.cuda(), l_offset=50.0), VGG_OUT)
    fA = vgg(gray2rgb_batch(fr.cuda()[:, 0:1]), VGG_OUT)
    y, sim, tp = warp(IB.cuda(), *[feature_normalize(t) for t in fA[1:]], *[feature_normalize(t) for t in fB[1:]],
                      temperature=T, return_taps=True)
    th, ph = tp["theta"][0].double(), tp["phi"][0].double()
    f = th.t() @ ph                                   # 5184 x 5184 fp64 on the device (test-only torch op)
    top2 = torch.topk(f, 2, dim=-1)
    gap = top2[0][:, 0] - top2[0][:, 1]
    amax_k = tp["argmax"][0].long()
    dis = amax_k != top2[1][:, 0]
    blab = ops.avgpool4x4(IB.cuda()).view(3, -1)
    y_k = tp["y_small"][0].view(3, -1)
    y_from_amax = blab[:, amax_k]
    colour_bad = ((y_k - y_from_amax).abs().max(0)[0] > 1e-3)
    sim_err = (tp["sim_small"].view(-1).double() - top2[0][:, 0]).abs().max().item()
    report(f"corr self-consistency seed={seed}: argmax!=fp64 on {int(dis.sum())} rows (max gap among them "
           f"{gap[dis].max().item() if dis.any() else 0:.2e}); colour!=blab[argmax] on {int(colour_bad.sum())} rows "
           f"(gaps {gap[colour_bad][:5].tolist()}); sim_err={sim_err:.2e}")
    assert sim_err < 2e-6
    assert (gap[dis] < 1e-5).all()
    assert (gap[colour_bad] < 1e-5).all()            # softmax one-hot must sit on the kernel's own argmax
    taps = {}
    with torch.no_grad():
        fB_o = O.exemplar_features(IB, weights[0])
        O.warp_color(fr[:, 0:1], IB, fB_o, weights[0], weights[1], temperature=T, taps=taps)
    th_err = (tp["theta"].cpu() - taps["theta"]).abs().max().item()
    ph_err = (tp["phi"].cpu() - taps["phi"]).abs().max().item()
    ogap = taps["top2"][0, :, 0] - taps["top2"][0, :, 1]
    odis = tp["argmax"][0].cpu().long() != taps["argmax"][0]
    report(f"corr vs oracle seed={seed}: theta_err={th_err:.2e} phi_err={ph_err:.2e} argmax disagreements "
           f"{int(odis.sum())} (max gap {ogap[odis].max().item() if odis.any() else 0:.2e})")
    assert th_err < 2e-5 and ph_err < 2e-5
    assert (ogap[odis] < 1e-4).all()


def test_warpnet_exemplar_cache_is_bit_identical(nets, weights):
    from dvc_amd import synth
    warp = nets[1]
    with torch.no_grad():
        nA = [t.cuda() for t in _norm_feats(weights[0], synth.synth_lab(1000, 48, 80))]
        nB = [t.cuda() for t in _norm_feats(weights[0], synth.synth_lab(2, 48, 80))]
    IB = synth.synth_lab(2, 48, 80).cuda()
    y0, s0 = warp(IB, *nA, *nB, temperature=0.01)
    cache = warp.exemplar_side(IB, *nB)
    y1, s1 = warp(IB, *nA, *nB, temperature=0.01, exemplar_cache=cache)
    assert torch.equal(y0, y1) and torch.equal(s0, s1)


@pytest.fixture
def conv_algo(request):
    """Run a test under a given convolution algorithm choice (ops.set_conv_algo) and restore the default afterwards."""
    from dvc_amd import ops
    old = ops.conv_algo()
    ops.set_conv_algo(request.param)
    yield request.param
    ops.set_conv_algo(old)


# With plain random weights the network amplifies rounding noise ~70x (dvc_amd/synth.py), so these tests bound the GPU's
# distance from the fp64 truth by a multiple of the reference-equivalent CPU fp32 run's distance.  The direct engine sits
# at or below the CPU run; Winograd F(2x2,3x3) carries 1.1-2.2x the rounding error of a direct fp32 sum per layer
# (profiles/r05_engine_layer_error.txt) — the same trade cuDNN makes for the reference under cudnn.benchmark (test.py:140).
# r05: the default choice ("auto") keeps the layers where that matters on the direct engine (arch.DIRECT_LAYERS, measured with
# tools/engine_sensitivity.py) and is held to the SAME factor as the direct engine; "speed" (the geometry rule alone, what "auto"
# meant up to r04) keeps the looser factor and is reported next to it.  The literal 1e-3 claim is asserted on the well-conditioned
# weights in tests/test_gpu_e2e.py, with the default algorithm.
WORST_CASE_FACTOR = {"direct": 1.5, "auto": 1.5, "speed": 2.5}


@pytest.mark.parametrize("conv_algo", ["auto", "direct", "speed"], indirect=True)
@pytest.mark.parametrize("H,W", [(48, 80), (216, 384)])
def test_colorvidnet(nets, weights, H, W, conv_algo):
    from oracle import dvc_oracle as O
    col = nets[2]
    g = torch.Generator().manual_seed(9)
    x = torch.randn(1, 7, H, W, generator=g) * torch.tensor([30, 40, 40, 0.3, 30, 40, 40.]).view(1, 7, 1, 1)
    with torch.no_grad():
        ref32 = O.colorvidnet_forward(weights[2], x)
        ref64 = O.colorvidnet_forward(O.to_dtype(weights[2], torch.float64), x.double())
    got = col(x.cuda())
    e_gpu = (got.double().cpu() - ref64).abs()
    e_cpu = (ref32.double() - ref64).abs()
    report(f"colorvidnet {H}x{W} conv={conv_algo}: gpu_vs_fp64 max={e_gpu.max():.2e} mean={e_gpu.mean():.2e} | "
           f"cpu32_vs_fp64 max={e_cpu.max():.2e} mean={e_cpu.mean():.2e}")
    assert got.shape == ref32.shape
    # fp32 tolerance on the +-128-range output: within 1e-3 max-abs of the fp64 truth, or — where the
    # reference's own fp32 run is not (it is not, with these random weights) — no further from the
    # truth than a small multiple of the reference's fp32 run on the same input (see WORST_CASE_FACTOR).
    assert e_gpu.max().item() < max(1e-3, WORST_CASE_FACTOR[conv_algo] * e_cpu.max().item())
    assert e_gpu.mean().item() < max(1e-4, 1.5 * e_cpu.mean().item())


def _run_clip(nets, H, W, nf, T, cache):
    from dvc_amd import synth
    from dvc_amd.frame import ClipColorizer
    vgg, warp, col = nets
    cc = ClipColorizer(vgg, warp, col, temperature=T, cache_exemplar=cache)
    cc.set_exemplar(synth.synth_lab(synth.EXEMPLAR_SEED, H, W).cuda())
    frames = [synth.synth_lab(synth.FRAME_SEED0 + i, H, W).cuda() for i in range(nf)]
    outs, warped = [], []
    last = torch.zeros_like(frames[0])
    for fr in frames:
        ab, nl = cc.frame(fr, last)
        last = torch.cat((fr[:, 0:1], ab), 1)
        outs.append(ab)
        warped.append(nl)
    return outs, warped


@pytest.mark.parametrize("conv_algo", ["auto", "direct", "speed"], indirect=True)
@pytest.mark.parametrize("name", ["small_48x80_T1e-10", "small_40x64_T0.01", "full_216x384_T1e-10"])
def test_frame_colorization_vs_reference_golden(nets, weights, golden_dir, name, conv_algo):
    """End-to-end vs outputs recorded from the UNMODIFIED reference (oracle/pin_reference.py), under both convolution
    engines.

    Every frame is compared on IDENTICAL inputs: frame i's `IA_last_lab` is built from the golden
    prediction of frame i-1 (with random weights ColorVidNet amplifies a 1e-3 difference in IA_last
    ~40x per frame, for the reference's own fp32-vs-fp64 runs just as much — see the free-running
    report line of the fp64 test — so a free-running comparison measures chaos, not the kernels).
    The golden is the reference's fp32 CPU run, itself 3e-3..6e-3 max-abs from the fp64 truth on ab,
    so this test pins the DISCRETE part exactly (which exemplar position every pixel picked; the
    similarity map) and bounds the continuous part at that noise level — ON EVERY FRAME (r04; it used to be "at least
    one clean frame"):
      * frames whose smallest recorded top-1/top-2 gap is >= 2e-6 (both frames of small_48x80, frame 0 of full_216x384:
        2.6e-5, 3.9e-5, 3.0e-6) must pick the reference's exemplar position on EVERY row;
      * on the other frames (full_216x384 frame 1: a row with gap 6.0e-7, below fp32 resolution of a 256-term dot product) a
        row may pick another position only if its recorded gap is < 1e-5, and `ab` is then compared with the reference
        arithmetic evaluated WITH THAT TIE-BREAK: the oracle's ColorVidNet (bit-identical to the reference module,
        oracle/pin_reference.py) on the golden warped colours / similarity / previous frame with the flipped rows' 4x4
        blocks carrying the colour the HIP path chose.  Without a flip that evaluation IS the golden `ab`;
      * the statistics every frame must meet: mean < 2e-3, p99 < 1e-2, and max below WORST_CASE_FACTOR x 1e-2 — 1.5e-2 for the
        direct engine and, since r05, for the default choice (the error-aware engine map); 2.5e-2 for "speed" = Winograd wherever
        the geometry allows (the reference's own fp32-vs-fp64 worst case on this network is 6.4e-3,
        test_colorvidnet, and the golden carries that error too; the MAX over 166k values of a chaotic network's error field
        moves by 50 % when one front-end layer rounds differently — measured r03 / r04: 1.13e-2 / 1.69e-2 with Winograd,
        7.9e-3 / 8.4e-3 direct — so the factor the less accurate engine costs is asserted on the worst case loosely and on
        p99 / mean tightly)."""
    import torch.nn.functional as F
    from dvc_amd import synth
    from dvc_amd.frame import VGG_OUT, frame_colorization
    from dvc_amd import ops
    from oracle import dvc_oracle as O
    vgg, warp, col = nets
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    H, W, nf, T = int(g["H"]), int(g["W"]), int(g["n_frames"]), float(g["temperature"])
    h, w = H // 4, W // 4
    IB = synth.synth_lab(synth.EXEMPLAR_SEED, H, W).cuda()
    fB = vgg(ops.lab2rgb(IB, l_offset=50.0), VGG_OUT)
    for i in range(nf):
        fr = synth.synth_lab(synth.FRAME_SEED0 + i, H, W).cuda()
        if i == 0:
            last = torch.zeros_like(fr)
        else:
            prev = synth.synth_lab(synth.FRAME_SEED0 + i - 1, H, W)
            last = torch.cat((prev[:, 0:1], torch.from_numpy(g["ab"][i - 1])[None]), 1).cuda()
        ab, nl, _ = frame_colorization(fr, IB, last, fB, vgg, warp, col, joint_training=False, temperature=T)
        nl_small = nl[0, :, ::4, ::4].cpu()
        wl = np.abs(nl_small.numpy() - g["warped_lab_small"][i])
        gap = g["top2gap"][i].reshape(h, w)
        flips = (wl.max(0) > 1e-3)
        if T < 1e-6:
            assert not (flips & (gap >= 1e-5)).any(), (name, i)     # a different exemplar pixel only on near-ties
            if gap.min() >= 2e-6:
                assert not flips.any(), (name, i, "this frame has no row below fp32 resolution: every row must agree")
            assert wl[:, ~flips].max() < 1e-4
            want = g["ab"][i]
            if flips.any():
                y_alt = torch.from_numpy(g["warped_lab_small"][i]).clone()
                fm = torch.from_numpy(flips)
                y_alt[:, fm] = nl_small[:, fm]
                nthreads = torch.get_num_threads()
                torch.set_num_threads(int(g["num_threads"]))        # the thread count the golden was recorded with
                try:
                    with torch.no_grad():
                        cin = torch.cat((fr.cpu()[:, 0:1], F.interpolate(y_alt[None], scale_factor=4, mode="nearest")[:, 1:3],
                                         F.interpolate(torch.from_numpy(g["sim"][i])[None, None], scale_factor=4, mode="nearest"),
                                         last.cpu()), dim=1)
                        want = O.colorvidnet_forward(weights[2], cin)[0].numpy()
                finally:
                    torch.set_num_threads(nthreads)
            d = np.abs(ab[0].cpu().numpy() - want)
            report(f"e2e golden {name} conv={conv_algo} frame{i}: ab max={d.max():.2e} mean={d.mean():.2e} p99={np.quantile(d, 0.99):.2e} "
                   f"warped max on agreeing rows={wl[:, ~flips].max():.2e} flipped_rows={int(flips.sum())} (their gaps "
                   f"{gap[flips].tolist()}; rows with gap<1e-5: {int((gap < 1e-5).sum())}; tie-break-matched reference: {bool(flips.any())})")
            assert d.mean() < 2e-3 and np.quantile(d, 0.99) < 1e-2, (name, i)
            assert d.max() < WORST_CASE_FACTOR[conv_algo] * 1e-2, (name, i, d.max())
        else:   # soft temperature: d(y)/d(f) = |B_lab|/T ~ 1e4 and fp32 affinities differ by ~1e-6
            # The bound is the reference's OWN fp32 noise at this temperature, measured here: the golden (= the reference's fp32
            # run) against the fp64 truth on the same inputs.  A GPU result no further from the truth than the golden is lies
            # within 2x that figure of the golden (triangle inequality); the maximum of the error field gets 2.5x.
            d = np.abs(ab[0].cpu().numpy() - g["ab"][i])
            sd64 = tuple(O.to_dtype(s_, torch.float64) for s_ in weights)
            with torch.no_grad():
                fB64 = O.exemplar_features(IB.cpu().double(), sd64[0])
                ab64, nl64, _ = O.frame_colorization(fr.cpu().double(), IB.cpu().double(), last.cpu().double(), fB64, *sd64, temperature=T)
            e_ref = np.abs(g["ab"][i] - ab64[0].numpy())
            w_ref = np.abs(g["warped_lab_small"][i] - nl64[0, :, ::4, ::4].numpy())
            report(f"e2e golden {name} conv={conv_algo} frame{i}: ab max={d.max():.2e} mean={d.mean():.2e} warped max={wl.max():.2e} | "
                   f"golden (reference fp32) vs fp64 on the same inputs: ab max={e_ref.max():.2e} mean={e_ref.mean():.2e} "
                   f"warped max={w_ref.max():.2e}")
            assert d.mean() <= 2.0 * e_ref.mean(), (name, i, d.mean(), e_ref.mean())
            assert d.max() <= 2.5 * e_ref.max(), (name, i, d.max(), e_ref.max())
            assert wl.max() <= 2.0 * w_ref.max() + 1e-4, (name, i, wl.max(), w_ref.max())


@pytest.mark.parametrize("H,W,T", [(48, 80, 1e-10), (40, 64, 0.01), (216, 384, 1e-10)])
def test_e2e_error_vs_fp64_oracle_next_to_cpu_fp32(nets, weights, H, W, T):
    """The honest form of the "ab within 1e-3" claim with the chaotic random weights (SURVEY.md §7 hard part 1, §8(c): "GPU-fp32
    vs fp64 oracle reported next to CPU-fp32 vs fp64 oracle (must be <= the CPU figure)"): GPU-fp32 vs the fp64 truth, next to the
    reference-equivalent CPU-fp32 vs the same truth, under the default engine choice ("auto" = what bench.py times), the direct
    engine and the geometry-only Winograd rule ("speed") in one test.

    r05 bar (r04 review, item 1): at test.py's temperature the TIMED engine must be no further from the truth than the CPU fp32
    run — mean <= 1.0x, q999 <= 1.0x, max <= 1.25x — and so must the direct engine; "speed" is reported, with the price it
    pays (r04: q999 1.8x, max 2.8x at 216x384) bounded at WORST_CASE_FACTOR["speed"].  r06: "the CPU fp32 run" is evaluated at two
    thread counts — its own tail statistics on this frame differ by 1.5x / 2.0x between them — and the single frame's q999 / max
    are held against that range, mean and rms against the better run; the 1.0x on q999 is held by the pooled test below.
    At the soft temperature the correlation's 1/T dominates both engines' errors (bounded against the CPU run at 1.5x)."""
    from dvc_amd import ops, synth
    from oracle import dvc_oracle as O
    sd32 = weights
    sd64 = tuple(O.to_dtype(s, torch.float64) for s in weights)
    IB = synth.synth_lab(synth.EXEMPLAR_SEED, H, W)
    fr = synth.synth_lab(synth.FRAME_SEED0, H, W)
    last = torch.zeros_like(fr)
    with torch.no_grad():
        fB32 = O.exemplar_features(IB, sd32[0])
        ab32, nl32, _ = O.frame_colorization(fr, IB, last, fB32, *sd32, temperature=T)
        fB64 = O.exemplar_features(IB.double(), sd64[0])
        ab64, nl64, _ = O.frame_colorization(fr.double(), IB.double(), last.double(), fB64, *sd64, temperature=T)
        # r06: the reference's fp32 run is not ONE number.  ATen's CPU convolutions block differently for another thread count,
        # and with the chaotic random weights one region of frame 1000 (rows 10-30, columns 218-233) amplifies whatever rounding
        # noise reaches it: the CPU run ITSELF moves from q99.9 3.9e-3 / max 5.9e-3 (16 threads) to 5.7e-3 / 1.2e-2 (4 threads)
        # on this frame while its mean moves by 5 % (profiles/r06_parity_hotspot.txt) — and so does any GPU arithmetic when only
        # a summation ORDER changes (profiles/r06_parity_pool_probe.txt: the same frame's tail ratio 0.84 ... 2.2, mean 0.81 ...
        # 0.89, across ten engine configurations).  The tail statistics are held against the reference's own RANGE, the mean
        # against its better run; the pooled test below holds q99.9 / mean / rms over several frames at 1.0x.
        n_thr = torch.get_num_threads()
        torch.set_num_threads(4 if n_thr != 4 else 2)
        try:
            ab32b = O.frame_colorization(fr, IB, last, O.exemplar_features(IB, sd32[0]), *sd32, temperature=T)[0]
        finally:
            torch.set_num_threads(n_thr)
    e_cpu = (ab32.double() - ab64).abs()
    e_cpu_b = (ab32b.double() - ab64).abs()
    w_cpu = (nl32.double() - nl64).abs()
    q = lambda t: np.quantile(t.numpy(), 0.999)
    rms = lambda t: float((t ** 2).mean().sqrt())      # noqa: E731
    report(f"e2e vs fp64 {H}x{W} T={T}: the reference's own spread — CPU32 at {n_thr} threads max={e_cpu.max():.2e} q999={q(e_cpu):.2e} "
           f"mean={e_cpu.mean():.2e}; at {4 if n_thr != 4 else 2} threads max={e_cpu_b.max():.2e} q999={q(e_cpu_b):.2e} mean={e_cpu_b.mean():.2e}")
    errs = {}
    old = ops.conv_algo()
    try:
        for algo in ("auto", "direct", "speed"):
            ops.set_conv_algo(algo)
            outs, warped = _run_clip(nets, H, W, 1, T, cache=True)
            e_gpu = (outs[0].double().cpu() - ab64).abs()
            w_gpu = (warped[0].double().cpu() - nl64).abs()
            errs[algo] = e_gpu
            report(f"e2e vs fp64 {H}x{W} T={T} conv={algo}: GPU ab max={e_gpu.max():.2e} q999={q(e_gpu):.2e} mean={e_gpu.mean():.2e} | "
                   f"CPU32 ab max={e_cpu.max():.2e} q999={q(e_cpu):.2e} mean={e_cpu.mean():.2e} | "
                   f"warped GPU max={w_gpu.max():.2e} CPU32 max={w_cpu.max():.2e}")
            assert e_gpu.mean().item() < max(1e-3, 1.5 * e_cpu.mean().item())
            assert q(e_gpu) < max(1e-3, WORST_CASE_FACTOR[algo] * q(e_cpu))
            assert w_gpu.max().item() < max(1e-3, 2.0 * w_cpu.max().item())
            if T < 1e-6 and algo != "speed":
                # the bar: the engine bench.py times (and the direct engine) at or below the reference's own fp32 error —
                # mean and rms against its better run, q99.9 / max against its own range (see above)
                assert e_gpu.mean().item() <= 1.0 * min(e_cpu.mean().item(), e_cpu_b.mean().item()), (algo, e_gpu.mean().item(), e_cpu.mean().item())
                assert rms(e_gpu) <= 1.0 * min(rms(e_cpu), rms(e_cpu_b)), (algo, rms(e_gpu), rms(e_cpu), rms(e_cpu_b))
                assert q(e_gpu) <= 1.0 * max(q(e_cpu), q(e_cpu_b)), (algo, q(e_gpu), q(e_cpu), q(e_cpu_b))
                assert e_gpu.max().item() <= 1.25 * max(e_cpu.max().item(), e_cpu_b.max().item()), (algo, e_gpu.max().item(), e_cpu.max().item())
            if H <= 64 and algo == "auto":
                # report only: how a FREE-RUNNING second frame (IA_last = own previous prediction) diverges
                fr1 = synth.synth_lab(synth.FRAME_SEED0 + 1, H, W)
                with torch.no_grad():
                    a32, _, _ = O.frame_colorization(fr1, IB, torch.cat((fr[:, 0:1], ab32), 1), fB32, *sd32, temperature=T)
                    a64, _, _ = O.frame_colorization(fr1.double(), IB.double(), torch.cat((fr[:, 0:1].double(), ab64), 1),
                                                     fB64, *sd64, temperature=T)
                o2, _ = _run_clip(nets, H, W, 2, T, cache=True)
                report(f"free-running frame1 {H}x{W} T={T}: GPU-vs-fp64 mean={(o2[1].double().cpu() - a64).abs().mean():.2e} "
                       f"| CPU32-vs-fp64 mean={(a32.double() - a64).abs().mean():.2e}")
    finally:
        ops.set_conv_algo(old)
    ratio = lambda a, b: (errs[a].max().item() / b.max().item(), q(errs[a]) / q(b), errs[a].mean().item() / b.mean().item())   # noqa: E731
    for a in ("auto", "direct", "speed"):
        r = ratio(a, e_cpu)
        report(f"e2e vs fp64 {H}x{W} T={T}: {a} / CPU32 error ratio max {r[0]:.2f} q999 {r[1]:.2f} mean {r[2]:.2f}")
    r = ratio("speed", errs["direct"])
    report(f"e2e vs fp64 {H}x{W} T={T}: speed (Winograd wherever the geometry allows) / direct error ratio max {r[0]:.2f} q999 {r[1]:.2f} "
           f"mean {r[2]:.2f}")
    if T < 1e-6:
        assert r[1] <= 2.5 and r[2] <= 1.6, r
        assert errs["speed"].max().item() <= 4.0 * e_cpu.max().item()


def test_timed_engine_no_further_from_fp64_than_cpu_fp32_pooled_216x384(nets, weights):
    """The same comparison pooled over several frames (one frame's error maximum is a lottery: over 10 frames the per-frame
    max ratio of the SAME engine ranges 0.5 ... 1.1): frames 1000..1005 as first frames of a clip at 216x384, T = 1e-10, the
    default engine against the fp64 oracle next to CPU fp32.  Frames on which an arg-max differs from the truth's (a near-tie
    row: the golden tests deal with those) are left out for both sides; at least three must remain.  Pooled rms, mean and q999
    <= 1.0x the CPU run's, pooled max <= 1.25x."""
    from dvc_amd import synth
    from dvc_amd.frame import ClipColorizer
    from oracle import dvc_oracle as O
    H, W, T = 216, 384, 1e-10
    sd64 = tuple(O.to_dtype(s, torch.float64) for s in weights)
    IB = synth.synth_lab(synth.EXEMPLAR_SEED, H, W)
    cc = ClipColorizer(*nets, temperature=T)
    cc.set_exemplar(IB.cuda())
    eg, ec, used = [], [], []
    with torch.no_grad():
        fB32 = O.exemplar_features(IB, weights[0])
        fB64 = O.exemplar_features(IB.double(), sd64[0])
    for seed in range(synth.FRAME_SEED0, synth.FRAME_SEED0 + 6):
        fr = synth.synth_lab(seed, H, W)
        z = torch.zeros_like(fr)
        with torch.no_grad():
            ab32, nl32, _ = O.frame_colorization(fr, IB, z, fB32, *weights, temperature=T)
            ab64, nl64, _ = O.frame_colorization(fr.double(), IB.double(), z.double(), fB64, *sd64, temperature=T)
        ab, nl = cc.frame(fr.cuda(), z.cuda())
        if (nl.double().cpu() - nl64).abs().max().item() > 1e-3 or (nl32.double() - nl64).abs().max().item() > 1e-3:
            report(f"pooled e2e vs fp64: frame seed {seed} left out (an arg-max differs from the fp64 truth's)")
            continue
        used.append(seed)
        eg.append((ab.double().cpu() - ab64).abs())
        ec.append((ab32.double() - ab64).abs())
    assert len(used) >= 3, used
    eg, ec = torch.cat(eg), torch.cat(ec)
    q = lambda t: np.quantile(t.numpy(), 0.999)       # noqa: E731
    rms = lambda t: t.pow(2).mean().sqrt().item()     # noqa: E731
    report(f"pooled e2e vs fp64 216x384 T=1e-10 over frames {used}: GPU(auto) max={eg.max():.2e} q999={q(eg):.2e} mean={eg.mean():.2e} "
           f"rms={rms(eg):.2e} | CPU32 max={ec.max():.2e} q999={q(ec):.2e} mean={ec.mean():.2e} rms={rms(ec):.2e} | ratios max "
           f"{eg.max().item() / ec.max().item():.2f} q999 {q(eg) / q(ec):.2f} mean {eg.mean().item() / ec.mean().item():.2f} "
           f"rms {rms(eg) / rms(ec):.2f}")
    assert rms(eg) <= rms(ec) and eg.mean().item() <= ec.mean().item() and q(eg) <= q(ec)
    assert eg.max().item() <= 1.25 * ec.max().item()


def test_clip_recurrence_cached_equals_uncached_and_deterministic(nets):
    a, _ = _run_clip(nets, 48, 80, 3, 1e-10, cache=True)
    b, _ = _run_clip(nets, 48, 80, 3, 1e-10, cache=False)
    c, _ = _run_clip(nets, 48, 80, 3, 1e-10, cache=True)
    for x, y, z in zip(a, b, c):
        assert torch.equal(x, y) and torch.equal(x, z)


def test_batch_of_two_frames_matches_single_frame_runs(nets, weights):
    """The API is batched (train.py:402 calls it with B=16): a B=2 call must equal two B=1 calls, including the exemplar
    batch — BIT FOR BIT: the library plans every launch per image (tile configuration, split over input channels,
    stream-K ranges, one correlation decomposition per image), so an image's result never depends on the batch it came in.
    Both temperatures, the whole frame (VGG taps, warped colours, ab)."""
    import contextlib
    import io
    from dvc_amd import ops, synth
    from dvc_amd.frame import VGG_OUT, frame_colorization
    from models.ColorVidNet import ColorVidNet
    vgg, warp, _ = nets
    with contextlib.redirect_stdout(io.StringIO()):
        col = ColorVidNet(7)
    col.load_state_dict(synth.colorvidnet_state_dict(0, contractive=True))
    col.eval().cuda()
    H, W = 48, 80
    IB = torch.cat([synth.synth_lab(2, H, W), synth.synth_lab(3, H, W)]).cuda()
    IA = torch.cat([synth.synth_lab(1000, H, W), synth.synth_lab(1001, H, W)]).cuda()
    last = torch.cat([synth.synth_lab(7, H, W), synth.synth_lab(8, H, W)]).cuda()
    fB = vgg(ops.lab2rgb(IB, l_offset=50.0), VGG_OUT)
    for T in (1e-10, 0.01):
        ab2, nl2, fA2 = frame_colorization(IA, IB, last, fB, vgg, warp, col, joint_training=False, temperature=T)
        assert ab2.shape == (2, 2, H, W) and nl2.shape == (2, 3, H, W) and fA2[0].shape[0] == 2
        for i in range(2):
            fBi = vgg(ops.lab2rgb(IB[i:i + 1].contiguous(), l_offset=50.0), VGG_OUT)
            ab1, nl1, _ = frame_colorization(IA[i:i + 1].contiguous(), IB[i:i + 1].contiguous(), last[i:i + 1].contiguous(),
                                             fBi, vgg, warp, col, joint_training=False, temperature=T)
            dn = (nl2[i:i + 1] - nl1).abs().max().item()
            d = (ab2[i:i + 1] - ab1).abs()
            report(f"batch-of-2 vs single T={T} image {i}: warped max diff {dn:.2e}, ab max diff {d.max().item():.2e} mean {d.mean().item():.2e}")
            assert torch.equal(nl2[i:i + 1], nl1), (T, i, dn)
            assert torch.equal(ab2[i:i + 1], ab1), (T, i, d.max().item())
            for k, (f2, f1) in enumerate(zip(fA2, vgg(ops.gray2rgb(IA[i:i + 1, 0:1]), VGG_OUT))):
                assert torch.equal(f2[i:i + 1], f1), (T, i, k)


def test_drop_in_signature_and_loud_cpu_failure(nets):
    import inspect
    from models.FrameColor import frame_colorization, warp_color
    sig = list(inspect.signature(frame_colorization).parameters)
    assert sig[:11] == ["IA_lab", "IB_lab", "IA_last_lab", "features_B", "vggnet", "nonlocal_net", "colornet",
                        "joint_training", "feature_noise", "luminance_noise", "temperature"]
    assert list(inspect.signature(warp_color).parameters)[:8] == [
        "IA_l", "IB_lab", "features_B", "vggnet", "nonlocal_net", "colornet", "feature_noise", "temperature"]
    with pytest.raises(RuntimeError):
        nets[2](torch.zeros(1, 7, 16, 16))      # CPU tensor must not silently fall back


def test_config5_bf16_correlation_path(nets, weights):
    """BASELINE configs[4] end to end at 216x384: WarpNet with corr_precision='bf16' (bf16 MFMA candidate
    filter + fp32 re-scoring) against the fp32 path.  Restated tolerance: identical exemplar pixel on
    every row whose fp32 top-1/top-2 gap exceeds 1e-5, similarity map within 2e-6, hence identical warped
    colours there; `ab` is then compared like any two fp32 runs (mean < 1e-3).  A soft temperature (0.01)
    silently uses the fp32 kernel (bit-identical)."""
    from dvc_amd import synth
    from dvc_amd.frame import ClipColorizer
    vgg, warp, col = nets
    H, W = 216, 384
    IB = synth.synth_lab(synth.EXEMPLAR_SEED, H, W).cuda()
    fr = synth.synth_lab(synth.FRAME_SEED0 + 1, H, W).cuda()
    last = torch.zeros_like(fr)
    outs = {}
    try:
        for prec in ("fp32", "bf16"):
            warp.corr_precision = prec
            cc = ClipColorizer(vgg, warp, col, temperature=1e-10)
            cc.set_exemplar(IB)
            outs[prec] = cc.frame(fr, last)
        warp.corr_precision = "bf16"
        cs = ClipColorizer(vgg, warp, col, temperature=0.01)
        cs.set_exemplar(IB)
        soft_b = cs.frame(fr, last)
        warp.corr_precision = "fp32"
        cs = ClipColorizer(vgg, warp, col, temperature=0.01)
        cs.set_exemplar(IB)
        soft_f = cs.frame(fr, last)
    finally:
        warp.corr_precision = "fp32"
    wl = (outs["bf16"][1] - outs["fp32"][1]).abs()[0, :, ::4, ::4].max(0)[0]
    d = (outs["bf16"][0] - outs["fp32"][0]).abs()
    report(f"config5 bf16 corr: warped rows differing {int((wl > 1e-3).sum())}/5184, ab diff max={d.max():.2e} mean={d.mean():.2e}")
    assert int((wl > 1e-3).sum()) <= 2          # only exact/near ties may differ
    if int((wl > 1e-3).sum()) == 0:
        assert d.mean().item() < 1e-3
    assert torch.equal(soft_b[0], soft_f[0]) and torch.equal(soft_b[1], soft_f[1])


def test_config4_432x768_end_to_end(nets, weights):
    """BASELINE configs[3]: 432x768 frames, N = 20736 correlation positions (the oracle's N x N path would
    need ~7 GB per temporary).  VGG is compared with the fp64 oracle; the correlation is checked for
    self-consistency against an fp64 evaluation of the theta/phi it consumed (argmax, similarity, exact
    one-hot gather); the whole frame is checked for determinism and finiteness."""
    from dvc_amd import ops, synth
    from dvc_amd.frame import VGG_OUT, ClipColorizer
    from oracle import dvc_oracle as O
    from utils.util import feature_normalize, gray2rgb_batch
    vgg, warp, col = nets
    H, W, T = 432, 768, 1e-10
    IB = synth.synth_lab(synth.EXEMPLAR_SEED, H, W)
    fr = synth.synth_lab(synth.FRAME_SEED0, H, W)
    # VGG stage parity (fp64 truth)
    x = O.gray2rgb_batch(fr[:, 0:1])
    with torch.no_grad():
        ref = O.vgg19_forward(O.to_dtype(weights[0], torch.float64), x.double(), ["r12", "r32", "r52"])
    got = vgg(x.cuda(), ["r12", "r32", "r52"])
    for k, g, r in zip(("r12", "r32", "r52"), got, ref):
        assert rel(g, r) < 1e-4, k
    # correlation self-consistency on the real features
    fB = vgg(ops.lab2rgb(IB.cuda(), l_offset=50.0), VGG_OUT)
    fA = vgg(gray2rgb_batch(fr.cuda()[:, 0:1]), VGG_OUT)
    y, sim, tp = warp(IB.cuda(), *[feature_normalize(t) for t in fA[1:]], *[feature_normalize(t) for t in fB[1:]],
                      temperature=T, return_taps=True)
    assert y.shape == (1, 3, H, W) and sim.shape == (1, 1, H, W)
    th, ph = tp["theta"][0].double(), tp["phi"][0].double()
    P = th.shape[1]
    assert P == 108 * 192
    best_v = torch.empty(P, dtype=torch.float64, device="cuda")
    best_i = torch.empty(P, dtype=torch.long, device="cuda")
    gap = torch.empty(P, dtype=torch.float64, device="cuda")
    for s0 in range(0, P, 4096):                        # chunked fp64 affinity rows (test-only torch ops)
        f = th[:, s0:s0 + 4096].t() @ ph
        t2 = torch.topk(f, 2, dim=-1)
        best_v[s0:s0 + 4096], best_i[s0:s0 + 4096] = t2[0][:, 0], t2[1][:, 0]
        gap[s0:s0 + 4096] = t2[0][:, 0] - t2[0][:, 1]
    amax = tp["argmax"][0].long()
    dis = amax != best_i
    sim_err = (tp["sim_small"].view(-1).double() - best_v).abs().max().item()
    blab = ops.avgpool4x4(IB.cuda()).view(3, -1)
    colour_bad = ((tp["y_small"][0].view(3, -1) - blab[:, amax]).abs().max(0)[0] > 1e-3)
    report(f"config4 432x768: argmax!=fp64 on {int(dis.sum())}/{P} rows (max gap {gap[dis].max().item() if dis.any() else 0:.2e}); "
           f"colour!=blab[argmax] on {int(colour_bad.sum())}; sim_err={sim_err:.2e}")
    assert sim_err < 2e-6 and (gap[dis] < 1e-5).all() and (gap[colour_bad] < 1e-5).all()
    # whole frame: deterministic and finite
    cc = ClipColorizer(vgg, warp, col, temperature=T)
    cc.set_exemplar(IB.cuda())
    a1, _ = cc.frame(fr.cuda(), torch.zeros_like(fr).cuda())
    a2, _ = cc.frame(fr.cuda(), torch.zeros_like(fr).cuda())
    assert a1.shape == (1, 2, H, W) and torch.isfinite(a1).all() and torch.equal(a1, a2)


def test_full_res_432x768_properties(nets):
    """BASELINE configs[3]: N = 20736 positions — size-independent properties only (the oracle would
    need 7 GB of N x N temporaries): one-hot gather identity, sim in [-1,1], determinism."""
    from dvc_amd import ops
    g = torch.Generator().manual_seed(4)
    h, w = 108, 192
    P = h * w
    th = ops.corr_prepare(torch.randn(1, 256, P, generator=g).cuda())
    ph = ops.corr_prepare(torch.randn(1, 256, P, generator=g).cuda())
    bl = torch.randn(1, 3, P, generator=g).cuda()
    o1 = ops.corr_fwd(th, ph, bl, 1e-10, h, w, want_small=True, want_argmax=True)
    o2 = ops.corr_fwd(th, ph, bl, 1e-10, h, w, want_small=True, want_argmax=True)
    assert torch.equal(o1["y_small"], o2["y_small"])
    assert o1["sim_small"].max() <= 1.0 + 1e-5 and o1["sim_small"].min() >= -1.0 - 1e-5
    gathered = torch.gather(bl, 2, o1["argmax"].long().unsqueeze(1).expand(1, 3, P))
    same = (o1["y_small"].view(1, 3, P) == gathered).all(1).float().mean().item()
    assert same > 0.9999
    # spot-check 64 rows against a direct fp64 evaluation
    idx = torch.randint(0, P, (64,), generator=g)
    f = th[0, :, idx.cuda()].double().t() @ ph[0].double()
    assert (f.max(1)[0].float() - o1["sim_small"].view(-1)[idx.cuda()]).abs().max().item() < 2e-6


@pytest.mark.gpu
def test_clip_pipelined_equals_sequential():
    """ClipColorizer.clip with look-ahead (front ends on side HIP streams, several frames per front-end batch) is
    bit-identical to the per-frame recurrence of test.py:68-96, for every look-ahead depth and batch size, and continues a
    recurrence via `last`."""
    import contextlib
    import io
    from dvc_amd import synth
    from dvc_amd.frame import ClipColorizer
    from models.ColorVidNet import ColorVidNet
    from models.NonlocalNet import VGG19_pytorch, WarpNet
    dev = torch.device("cuda")
    with contextlib.redirect_stdout(io.StringIO()):
        nets = (VGG19_pytorch(), WarpNet(1), ColorVidNet(7))
    for m, sd in zip(nets, (synth.vgg19_state_dict(0), synth.warpnet_state_dict(0), synth.colorvidnet_state_dict(0))):
        m.load_state_dict(sd)
        m.eval().to(dev)
    H, W = 48, 80
    frames = [synth.synth_lab(synth.FRAME_SEED0 + i, H, W).to(dev) for i in range(7)]
    cc = ClipColorizer(*nets, temperature=1e-10)
    cc.set_exemplar(synth.synth_lab(synth.EXEMPLAR_SEED, H, W).to(dev))
    ref = cc.clip(frames, lookahead=0)
    for la, fb in ((1, 1), (2, 1), (3, 1), (1, 2), (2, 2), (2, 3), (2, 4), (1, 7)):    # look-ahead depth x frames per front-end batch
        got = cc.clip(frames, lookahead=la, front_batch=fb)
        torch.cuda.synchronize()
        assert len(got) == len(ref)
        for a, b in zip(got, ref):
            assert torch.equal(a, b), (la, fb)
    # split the clip in two calls
    first = cc.clip(frames[:3], lookahead=2)
    rest = cc.clip(frames[3:], last=cc.last_lab, lookahead=2)
    for a, b in zip(first + rest, ref):
        assert torch.equal(a, b)


def test_graph_replay_equals_eager():
    """hipGraph replay (dvc_amd/graph.py) of the per-frame launch sequences of test.py:68-96: the
    captured front end and ColorVidNet chain, replayed through ClipColorizer.clip(graph=True) (pipelined, one graph per side
    stream) and through the per-frame call ClipColorizer.frame(graph=True), give predictions BIT-IDENTICAL to the eager
    launches — for every look-ahead depth, in both recurrence modes, when a call continues an earlier one, after the
    exemplar is replaced (the captured front ends read the refreshed cache in place) and after a weight is reloaded (the
    sequences are re-captured)."""
    import contextlib
    import io
    from dvc_amd import synth
    from dvc_amd.frame import ClipColorizer
    from models.ColorVidNet import ColorVidNet
    from models.NonlocalNet import VGG19_pytorch, WarpNet
    dev = torch.device("cuda")
    with contextlib.redirect_stdout(io.StringIO()):
        nets = (VGG19_pytorch(), WarpNet(1), ColorVidNet(7))
    sds = (synth.vgg19_state_dict(0), synth.warpnet_state_dict(0), synth.colorvidnet_state_dict(0))
    for m, sd in zip(nets, sds):
        m.load_state_dict(sd)
        m.eval().to(dev)
    H, W = 48, 80
    frames = [synth.synth_lab(synth.FRAME_SEED0 + i, H, W).to(dev) for i in range(7)]
    IB = synth.synth_lab(synth.EXEMPLAR_SEED, H, W).to(dev)
    eager = ClipColorizer(*nets, temperature=1e-10)
    eager.set_exemplar(IB)
    cc = ClipColorizer(*nets, temperature=1e-10, graph=True)
    cc.set_exemplar(IB)
    for fp in (False, True):
        ref = eager.clip(frames, lookahead=0, frame_propagate=fp)
        for la in (0, 1, 2, 3):
            got = cc.clip(frames, lookahead=la, frame_propagate=fp)
            torch.cuda.synchronize()
            assert len(got) == len(ref)
            for t, (a, b) in enumerate(zip(got, ref)):
                assert torch.equal(a, b), (fp, la, t)
            assert torch.equal(cc.last_lab, eager.last_lab)
    ref = eager.clip(frames, lookahead=0)
    # every choice of what the pipelined driver replays (default "front": see ClipColorizer.graph_parts)
    for parts in ("all", "color", "front"):
        cc.graph_parts = parts
        got = cc.clip(frames, lookahead=2)
        assert all(torch.equal(a, b) for a, b in zip(got, ref)), parts
    # replays do not alias their outputs: the predictions of one call survive the next call
    keep = cc.clip(frames, lookahead=2)
    cc.clip(list(reversed(frames)), lookahead=2)
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(keep, ref))
    # split the clip in two calls; per-frame API
    first = cc.clip(frames[:3], lookahead=2)
    rest = cc.clip(frames[3:], last=cc.last_lab, lookahead=2)
    assert all(torch.equal(a, b) for a, b in zip(first + rest, ref))
    last = torch.zeros_like(frames[0])
    for t, f in enumerate(frames[:4]):
        ab, nl = cc.frame(f, last)
        ab_e, nl_e = eager.frame(f, last)
        assert torch.equal(ab, ab_e) and torch.equal(nl, nl_e) and torch.equal(ab, ref[t])
        last = torch.cat((f[:, 0:1], ab), dim=1)
    n_graphs = len(cc._graphs)
    # another exemplar of the same geometry: the captured sequences stay, the cache they read is refreshed in place
    IB2 = synth.synth_lab(synth.EXEMPLAR_SEED + 5, H, W).to(dev)
    eager.set_exemplar(IB2)
    cc.set_exemplar(IB2)
    ref2 = eager.clip(frames, lookahead=0)
    got2 = cc.clip(frames, lookahead=2)
    assert len(cc._graphs) == n_graphs and all(torch.equal(a, b) for a, b in zip(got2, ref2))
    assert not torch.equal(ref2[0], ref[0])
    # reloaded weights: packs change, the sequences are re-captured on the next call
    sd2 = synth.colorvidnet_state_dict(3)
    nets[2].load_state_dict(sd2)
    ref3 = eager.clip(frames, lookahead=0)
    got3 = cc.clip(frames, lookahead=2)
    assert all(torch.equal(a, b) for a, b in zip(got3, ref3)) and not torch.equal(ref3[0], ref2[0])
    # the bf16 candidate-filter correlation (configs[4]) inside the captured front end: atomics, counters and all
    nets[1].corr_precision = "bf16"
    try:
        e16 = ClipColorizer(*nets, temperature=1e-10)
        e16.set_exemplar(IB)
        g16 = ClipColorizer(*nets, temperature=1e-10, graph=True)
        g16.set_exemplar(IB)
        ref16 = e16.clip(frames, lookahead=0)
        for la in (0, 2):
            got16 = g16.clip(frames, lookahead=la)
            assert all(torch.equal(a, b) for a, b in zip(got16, ref16)), la
        got16 = g16.clip(frames, lookahead=2)          # replayed again: the candidate counters are reset inside the sequence
        assert all(torch.equal(a, b) for a, b in zip(got16, ref16))
    finally:
        nets[1].corr_precision = "fp32"
    # a graph-mode driver without an exemplar cache says so
    with pytest.raises(RuntimeError, match="exemplar cache"):
        bad = ClipColorizer(*nets, temperature=1e-10, cache_exemplar=False, graph=True)
        bad.set_exemplar(IB)
        bad.clip(frames[:2], lookahead=0)


def test_graph_replay_notices_reloaded_weights_without_an_eager_call():
    """Advisor (r03): a replayed sequence never reaches nets._PackCache, so an in-place `load_state_dict` followed DIRECTLY by
    a graph-mode call — no eager forward, no prepare() in between — used to replay the graphs that read the OLD packed weights.
    ClipColorizer._sync_weights fingerprints the parameters before every replayed call: the per-frame call, the sequential
    clip and the pipelined clip must all see the new weights."""
    import contextlib
    import io
    from dvc_amd import synth
    from dvc_amd.frame import ClipColorizer
    from models.ColorVidNet import ColorVidNet
    from models.NonlocalNet import VGG19_pytorch, WarpNet
    dev = torch.device("cuda")
    with contextlib.redirect_stdout(io.StringIO()):
        nets = (VGG19_pytorch(), WarpNet(1), ColorVidNet(7))
    for m, sd in zip(nets, (synth.vgg19_state_dict(0), synth.warpnet_state_dict(0), synth.colorvidnet_state_dict(0))):
        m.load_state_dict(sd)
        m.eval().to(dev)
    H, W = 48, 80
    frames = [synth.synth_lab(synth.FRAME_SEED0 + i, H, W).to(dev) for i in range(3)]
    IB = synth.synth_lab(synth.EXEMPLAR_SEED, H, W).to(dev)
    zero = torch.zeros_like(frames[0])
    cc = ClipColorizer(*nets, temperature=1e-10, graph=True)
    cc.set_exemplar(IB)
    ab0, _ = cc.frame(frames[0], zero)                     # captures both sequences
    for step, seed in enumerate((5, 6, 7)):
        nets[2].load_state_dict(synth.colorvidnet_state_dict(seed))     # in place: same parameter objects, new versions
        if step == 0:
            got = cc.frame(frames[0], zero)[0]
        elif step == 1:
            got = cc.clip(frames[:1], lookahead=0)[0]
        else:
            got = cc.clip(frames, lookahead=2)[0]
        torch.cuda.synchronize()
        fresh = ClipColorizer(*nets, temperature=1e-10)    # eager, same modules: packs the CURRENT weights
        fresh.set_exemplar(IB)
        want = fresh.frame(frames[0], zero)[0]
        assert torch.equal(got, want), (step, (got - want).abs().max().item())
        assert not torch.equal(got, ab0)


@pytest.mark.parametrize("graph", [False, True])
def test_reloaded_front_end_weights_refresh_the_cached_exemplar_side(graph):
    """Advisor (r04): the cached exemplar side (phi, pooled Lab) was computed with the VGG19 / WarpNet weights of its time; after
    an in-place `load_state_dict` of either, a replayed graph (or an eager call with the cache) used to mix a NEW frame side
    with the STALE exemplar side.  ClipColorizer._sync_weights now recomputes it from the exemplar when the front end's
    fingerprint moves — on the per-frame call, the sequential and the pipelined clip."""
    import contextlib
    import io
    from dvc_amd import synth
    from dvc_amd.frame import ClipColorizer
    from models.ColorVidNet import ColorVidNet
    from models.NonlocalNet import VGG19_pytorch, WarpNet
    dev = torch.device("cuda")
    with contextlib.redirect_stdout(io.StringIO()):
        nets = (VGG19_pytorch(), WarpNet(1), ColorVidNet(7))
    for m, sd in zip(nets, (synth.vgg19_state_dict(0), synth.warpnet_state_dict(0), synth.colorvidnet_state_dict(0))):
        m.load_state_dict(sd)
        m.eval().to(dev)
    H, W = 48, 80
    frames = [synth.synth_lab(synth.FRAME_SEED0 + i, H, W).to(dev) for i in range(3)]
    IB = synth.synth_lab(synth.EXEMPLAR_SEED, H, W).to(dev)
    zero = torch.zeros_like(frames[0])
    cc = ClipColorizer(*nets, temperature=0.01, graph=graph)
    cc.set_exemplar(IB)
    ab0, _ = cc.frame(frames[0], zero)
    for step, (which, seed) in enumerate(((1, 5), (0, 6), (1, 7))):
        sd = synth.warpnet_state_dict(seed) if which == 1 else synth.vgg19_state_dict(seed)
        nets[which].load_state_dict(sd)                  # in place: same parameter objects, new versions
        if step == 0:
            got = cc.frame(frames[0], zero)[0]
        elif step == 1:
            got = cc.clip(frames[:1], lookahead=0)[0]
        else:
            got = cc.clip(frames, lookahead=2)[0]
        torch.cuda.synchronize()
        fresh = ClipColorizer(*nets, temperature=0.01)   # eager, same modules, exemplar side built with the CURRENT weights
        fresh.set_exemplar(IB)
        want = fresh.frame(frames[0], zero)[0]
        assert torch.equal(got, want), (step, (got - want).abs().max().item())
        assert not torch.equal(got, ab0)


def test_folded_merge_is_bit_identical_in_the_drivers(nets):
    """ops.set_fold_merge: with the correlation's merge folded into pack_color_input (default) and with the separate merge +
    pack launches, frame_colorization (incl. the warped Lab it returns), the sequential and the pipelined clip and the
    graph-replayed per-frame call give the same bits; one kernel launch less per frame on the recurrence's critical path and
    one less in the front end."""
    from dvc_amd import ops, synth
    from dvc_amd.frame import VGG_OUT, ClipColorizer, frame_colorization
    vgg, warp, col = nets
    H, W = 48, 80
    IB = synth.synth_lab(synth.EXEMPLAR_SEED, H, W).cuda()
    frames = [synth.synth_lab(synth.FRAME_SEED0 + i, H, W).cuda() for i in range(4)]
    res = {}
    old = ops.fold_merge()
    try:
        for fold in (True, False):
            ops.set_fold_merge(fold)
            fB = vgg(ops.lab2rgb(IB, l_offset=50.0), VGG_OUT)
            ab, nl, _ = frame_colorization(frames[0], IB, torch.zeros_like(frames[0]), fB, vgg, warp, col, joint_training=False,
                                           temperature=1e-10)
            ab_s, nl_s, _ = frame_colorization(frames[1], IB, frames[0], fB, vgg, warp, col, joint_training=False, temperature=0.01)
            cc = ClipColorizer(vgg, warp, col, temperature=1e-10)
            cc.set_exemplar(IB)
            seq = cc.clip(frames, lookahead=0)
            pipe = cc.clip(frames, lookahead=2)
            cg = ClipColorizer(vgg, warp, col, temperature=1e-10, graph=True)
            cg.set_exemplar(IB)
            gab, gnl = cg.frame(frames[0], torch.zeros_like(frames[0]))
            gclip = cg.clip(frames, lookahead=2)
            torch.cuda.synchronize()
            res[fold] = [ab, nl, ab_s, nl_s] + seq + pipe + [gab, gnl] + gclip
    finally:
        ops.set_fold_merge(old)
    assert len(res[True]) == len(res[False])
    for i, (a, b) in enumerate(zip(res[True], res[False])):
        assert torch.equal(a, b), i
    assert torch.equal(res[True][0], res[True][4]) and torch.equal(res[True][0], res[True][12])      # frame 0: all drivers agree


def test_luminance_noise_path(nets):
    """frame_colorization(luminance_noise=s) (models/FrameColor.py:55-57) adds s * randn to the L channel that feeds BOTH
    the VGG front end and ColorVidNet's first input channel: equal, bit for bit, to a call with the same noise already
    added to IA_lab (same device generator state), and different from the noise-free call."""
    from dvc_amd import ops, synth
    from dvc_amd.frame import VGG_OUT, frame_colorization
    vgg, warp, col = nets
    H, W, T, s = 48, 80, 0.01, 2.5
    IB = synth.synth_lab(synth.EXEMPLAR_SEED, H, W).cuda()
    fr = synth.synth_lab(synth.FRAME_SEED0, H, W).cuda()
    last = synth.synth_lab(synth.FRAME_SEED0 - 1, H, W).cuda()
    fB = vgg(ops.lab2rgb(IB, l_offset=50.0), VGG_OUT)
    torch.manual_seed(1234)
    ab_n, nl_n, _ = frame_colorization(fr, IB, last, fB, vgg, warp, col, joint_training=False, luminance_noise=s, temperature=T)
    torch.manual_seed(1234)
    noise = torch.randn_like(fr[:, 0:1]) * s
    fr2 = torch.cat((fr[:, 0:1] + noise, fr[:, 1:3]), dim=1).contiguous()
    ab_p, nl_p, _ = frame_colorization(fr2, IB, last, fB, vgg, warp, col, joint_training=False, temperature=T)
    assert torch.equal(ab_n, ab_p) and torch.equal(nl_n, nl_p)
    ab_0, _, _ = frame_colorization(fr, IB, last, fB, vgg, warp, col, joint_training=False, temperature=T)
    assert (ab_0 - ab_n).abs().max().item() > 1e-3


def test_fused_split_k_reduce_is_bit_identical_in_the_networks(nets, weights):
    """WarpNet trunk and ColorVidNet with the split-K reduce folded into the InstanceNorm launches (default) against the same
    networks with separate reduce launches (ops.set_fuse_reduce(False)): bit for bit."""
    from dvc_amd import ops
    vgg, warp, col = nets
    g = torch.Generator().manual_seed(21)
    feats = [torch.randn(1, c, h, w, generator=g).cuda() for c, h, w in ((128, 108, 192), (256, 54, 96), (512, 27, 48), (512, 13, 24))]
    x = (torch.randn(1, 7, 216, 384, generator=g) * torch.tensor([30, 40, 40, 0.3, 30, 40, 40.]).view(1, 7, 1, 1)).cuda()
    try:
        ops.set_fuse_reduce(True)
        ops.conv_record = []
        a_w, a_c = warp.features(*feats), col(x)
        rec, ops.conv_record = ops.conv_record, None
        ops.set_fuse_reduce(False)
        b_w, b_c = warp.features(*feats), col(x)
    finally:
        ops.set_fuse_reduce(True)
        ops.conv_record = None
    assert torch.equal(a_w, b_w) and torch.equal(a_c, b_c)
    assert len(rec) > 30
