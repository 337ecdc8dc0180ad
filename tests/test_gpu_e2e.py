"""GPU parity, end to end: the north-star tolerance asserted LITERALLY.

BASELINE.json: "ab-channel output within 1e-3 max-abs of the reference".  With the plain He-uniform synthetic
weights the reference's own CPU fp32 run does not agree with itself to that level (thread count alone moves
free-running frames by 6e-3 / 0.26 / 15.8, see tests/test_gpu_nets.py), so those weights are used for the
per-stage tests only.  Here the ColorVidNet weights are the well-conditioned set
`synth.colorvidnet_state_dict(contractive=True)` — under it the oracle agrees with itself across thread counts
to < 4e-5 on every frame of a free-running clip at 216x384 (measured: 1-vs-8 threads 1.3e-5 .. 3.8e-5, fp32-vs-
fp64 2.5e-5 .. 4.1e-5 over 6 frames; tests/test_oracle_golden.py asserts the property on CPU) — and the frames
are ones whose hard arg-max (test.py's temperature 1e-10) is well separated on every row
(`synth.WELL_SEPARATED_FRAME_SEEDS_216x384`).  VGG19 and WarpNet keep the plain random weights.
"""
import contextlib
import io
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

REPORT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "test_report.txt")
NORTH_STAR_TOL = 1e-3        # BASELINE.json north_star: ab within 1e-3 max-abs of the reference


def report(line):
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    with open(REPORT, "a") as f:
        f.write(line + "\n")


def _oracle_threads():
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(32, avail)))


def _state_dicts():
    from dvc_amd import synth
    return (synth.vgg19_state_dict(0), synth.warpnet_state_dict(0), synth.colorvidnet_state_dict(0, contractive=True))


def _fresh_nets(sd):
    """Newly constructed modules: nothing packed, no stream pools warm (the state smoke() starts from)."""
    from models.ColorVidNet import ColorVidNet
    from models.NonlocalNet import VGG19_pytorch, WarpNet
    with contextlib.redirect_stdout(io.StringIO()):
        nets = (VGG19_pytorch(), WarpNet(1), ColorVidNet(7))
    for m, s in zip(nets, sd):
        m.load_state_dict(s)
        m.eval().cuda()
    return nets


def _oracle_clip(frames, IB, sd, T, frame_propagate=False, want_gaps=True):
    """oracle.colorize_clip, additionally returning each frame's smallest top-1/top-2 affinity gap."""
    from oracle import dvc_oracle as O
    outs, gaps = [], []
    with torch.no_grad():
        fB = O.exemplar_features(IB, sd[0])
        last = IB if frame_propagate else torch.zeros_like(frames[0])
        for fr in frames:
            taps = {} if want_gaps else None
            ab, _, _ = O.frame_colorization(fr, IB, last, fB, *sd, temperature=T, taps=taps)
            last = torch.cat((fr[:, 0:1], ab), dim=1)
            outs.append(ab)
            if want_gaps:
                gaps.append((taps["top2"][0, :, 0] - taps["top2"][0, :, 1]).min().item())
    return outs, gaps


def test_free_running_clip_within_1e3_of_oracle_216x384():
    """configs[1] geometry, free-running recurrence (frame t consumes the HIP path's own prediction of frame
    t-1, the oracle its own), 6 frames: |ab_gpu - ab_oracle| <= 1e-3 on EVERY frame.  The clip goes through
    ClipColorizer.clip with look-ahead on freshly built modules first (cold weight caches and cold side
    streams: the ordering bug class the advisor flagged) and must equal the per-frame loop bit for bit."""
    from dvc_amd import synth
    from dvc_amd.frame import ClipColorizer
    H, W, T = 216, 384, 1e-10
    _oracle_threads()
    sd = _state_dicts()
    seeds = list(synth.WELL_SEPARATED_FRAME_SEEDS_216x384) + list(synth.WELL_SEPARATED_FRAME_SEEDS_216x384[:2])
    IB = synth.synth_lab(synth.EXEMPLAR_SEED, H, W)
    frames = [synth.synth_lab(s, H, W) for s in seeds]
    nets = _fresh_nets(sd)
    cc = ClipColorizer(*nets, temperature=T)
    cc.set_exemplar(IB.cuda())
    got = cc.clip([f.cuda() for f in frames], lookahead=2)            # cold start, pipelined
    torch.cuda.synchronize()
    seq = cc.clip([f.cuda() for f in frames], lookahead=0)
    for a, b in zip(got, seq):
        assert torch.equal(a, b), "cold pipelined clip != per-frame loop"
    ref, gaps = _oracle_clip(frames, IB, sd, T)
    for i, (g, r) in enumerate(zip(got, ref)):
        d = (g.cpu() - r).abs()
        report(f"e2e literal 216x384 frame{i} (seed {seeds[i]}): |ab| max={r.abs().max():.2f} mean={r.abs().mean():.3f} "
               f"gpu-vs-oracle max={d.max():.2e} mean={d.mean():.2e}; oracle min top-1/top-2 gap {gaps[i]:.2e}")
        assert gaps[i] > 1e-6, (i, gaps[i])                 # the precondition the seeds were chosen for
        assert r.abs().max().item() > 1.0                  # a real colour signal
        assert d.max().item() <= NORTH_STAR_TOL, (i, d.max().item())
        assert d.max().item() <= 2.5e-4, (i, d.max().item())   # ~6x the oracle's own thread-count noise


@pytest.mark.parametrize("frame_propagate", [False, True])
def test_free_running_clip_small_and_frame_propagate(frame_propagate):
    """48x80, 4 frames, both recurrence modes of test.py:50,76-80 (`frame_propagate`: the exemplar's own Lab is
    the first frame's `I_last_lab_predict`) against oracle.colorize_clip, literal tolerance."""
    from dvc_amd import synth
    from dvc_amd.frame import ClipColorizer
    from oracle import dvc_oracle as O
    H, W, T = 48, 80, 1e-10
    _oracle_threads()
    sd = _state_dicts()
    IB = synth.synth_lab(synth.EXEMPLAR_SEED, H, W)
    frames = [synth.synth_lab(synth.FRAME_SEED0 + i, H, W) for i in range(4)]
    cc = ClipColorizer(*_fresh_nets(sd), temperature=T)
    cc.set_exemplar(IB.cuda())
    got = cc.clip([f.cuda() for f in frames], frame_propagate=frame_propagate, lookahead=2)
    with torch.no_grad():
        ref = O.colorize_clip(frames, IB, *sd, temperature=T, frame_propagate=frame_propagate)
    for i, (g, r) in enumerate(zip(got, ref)):
        d = (g.cpu() - r).abs().max().item()
        report(f"e2e literal 48x80 frame_propagate={frame_propagate} frame{i}: gpu-vs-oracle max={d:.2e}")
        assert d <= NORTH_STAR_TOL and d <= 2.5e-4, (i, d)
    if frame_propagate:     # the two modes differ from frame 0 on (I_last = exemplar Lab instead of zeros)
        other = cc.clip([f.cuda() for f in frames], frame_propagate=False, lookahead=0)
        assert (other[0] - got[0]).abs().max().item() > 1e-3


def test_long_free_running_clip_replayed_as_hipgraphs_48x80():
    """32 free-running frames (test.py:68-96 runs whole clips; the recurrence feeds every prediction back) through the clip
    driver with the per-frame launch sequences REPLAYED as hipGraphs (ClipColorizer(graph=True), look-ahead 2), against
    oracle.colorize_clip: the literal 1e-3 on every frame, no growth along the clip, and bit-identical to the eager clip.
    (All 32 frames have an oracle top-1/top-2 affinity gap >= 8e-6 at this size, checked once on the CPU.)"""
    from dvc_amd import synth
    from dvc_amd.frame import ClipColorizer
    from oracle import dvc_oracle as O
    H, W, T, NF = 48, 80, 1e-10, 32
    _oracle_threads()
    sd = _state_dicts()
    IB = synth.synth_lab(synth.EXEMPLAR_SEED, H, W)
    frames = [synth.synth_lab(synth.FRAME_SEED0 + i, H, W) for i in range(NF)]
    cc = ClipColorizer(*_fresh_nets(sd), temperature=T, graph=True)
    cc.set_exemplar(IB.cuda())
    got = cc.clip([f.cuda() for f in frames], lookahead=2)
    torch.cuda.synchronize()
    eager = cc.clip([f.cuda() for f in frames], lookahead=2, graph=False)
    for a, b in zip(got, eager):
        assert torch.equal(a, b), "replayed clip != eager clip"
    with torch.no_grad():
        ref = O.colorize_clip(frames, IB, *sd, temperature=T)
    errs = [(g.cpu() - r).abs().max().item() for g, r in zip(got, ref)]
    report(f"e2e literal 48x80, {NF} free-running frames, hipGraph replay: gpu-vs-oracle max per frame "
           f"first 8 {[f'{e:.1e}' for e in errs[:8]]} ... last 8 {[f'{e:.1e}' for e in errs[-8:]]}; max {max(errs):.2e}")
    assert max(errs) <= NORTH_STAR_TOL and max(errs) <= 2.5e-4, errs
    assert max(errs[-8:]) <= 4 * max(max(errs[:8]), 1e-5), errs         # no drift along the recurrence


def test_config4_432x768_against_oracle():
    """BASELINE configs[3] (432x768, N = 20736 correlation positions) against ORACLE TENSORS, stage by stage on
    identical stage inputs and end to end: VGG taps, WarpNet trunk, theta/phi, similarity map, arg-max, warped
    colours, ColorVidNet, ab.  The oracle's correlation runs row-chunked (oracle.correlate_chunked: the N x N
    matrix is 1.72 GB in fp32 and `correlate` holds several copies)."""
    from dvc_amd import ops, synth
    from dvc_amd.frame import VGG_OUT, ClipColorizer
    from oracle import dvc_oracle as O
    from utils.util import feature_normalize, gray2rgb_batch
    H, W, T = 432, 768, 1e-10
    _oracle_threads()
    sd = _state_dicts()
    vgg, warp, col = _fresh_nets(sd)
    IB = synth.synth_lab(synth.EXEMPLAR_SEED, H, W)
    fr = synth.synth_lab(synth.WELL_SEPARATED_FRAME_SEED_432x768, H, W)   # hard arg-max well separated on every row
    prev = synth.synth_lab(synth.FRAME_SEED0 - 1, H, W)       # a non-trivial I_last for the ColorVidNet stage

    def rel(g, r):
        return ((g.double().cpu() - r.double()).abs().max() / r.double().abs().max().clamp_min(1e-30)).item()

    with torch.no_grad():
        # ---- oracle, fp32, the reference's op order
        fB_o = O.exemplar_features(IB, sd[0])
        fA_o = O.vgg19_forward(sd[0], O.gray2rgb_batch(fr[:, 0:1]), O.VGG_OUT)
        nA_o = [O.feature_normalize(t) for t in fA_o[1:]]
        nB_o = [O.feature_normalize(t) for t in fB_o[1:]]
        A_feat_o = O.warp_features(sd[1], *nA_o)
        B_feat_o = O.warp_features(sd[1], *nB_o)
        th_o = O.corr_project(sd[1], "theta", A_feat_o)
        ph_o = O.corr_project(sd[1], "phi", B_feat_o)
        y_o, sim_o, amax_o, gap_o = O.correlate_chunked(th_o, ph_o, IB, T)
        y_up_o = torch.nn.functional.interpolate(y_o, scale_factor=4, mode="nearest")
        sim_up_o = torch.nn.functional.interpolate(sim_o, scale_factor=4, mode="nearest")
        cin_o = torch.cat((fr[:, 0:1], y_up_o[:, 1:3], sim_up_o, prev), dim=1)
        ab_o = O.colorvidnet_forward(sd[2], cin_o)
    # ---- stage by stage on the ORACLE's stage inputs
    fA = vgg(gray2rgb_batch(fr.cuda()[:, 0:1]), VGG_OUT)
    for k, g, r in zip(VGG_OUT, fA, fA_o):
        e = rel(g, r)
        report(f"config4 432x768 vgg {k}: rel_err={e:.2e}")
        assert e < 1e-4, (k, e)
    A_feat = warp.features(*[t.cuda() for t in nA_o])
    e = rel(A_feat, A_feat_o)
    report(f"config4 432x768 warp trunk (identical inputs): rel_err={e:.2e}")
    assert e < 2e-4
    th = warp.project("theta", A_feat_o.cuda())
    ph = warp.project("phi", B_feat_o.cuda())
    e_th, e_ph = (th.cpu() - th_o).abs().max().item(), (ph.cpu() - ph_o).abs().max().item()
    report(f"config4 432x768 theta/phi (identical inputs): abs_err={e_th:.2e} / {e_ph:.2e}")
    assert e_th < 2e-6 and e_ph < 2e-6
    res = ops.corr_fwd(th_o.cuda(), ph_o.cuda(), ops.avgpool4x4(IB.cuda()).view(1, 3, -1), T, H // 4, W // 4,
                       want_small=True, want_argmax=True)
    dis = res["argmax"][0].cpu().long() != amax_o[0]
    sim_err = (res["sim_small"].cpu() - sim_o).abs().max().item()
    yerr = (res["y_small"].cpu() - y_o).abs().view(3, -1).max(0)[0]
    report(f"config4 432x768 correlation (identical theta/phi): argmax differs on {int(dis.sum())}/20736 rows (max gap "
           f"among them {gap_o[0][dis].max().item() if dis.any() else 0:.2e}), sim_err={sim_err:.2e}, warped colour max err on "
           f"agreeing rows {yerr[~dis].max().item():.2e}; oracle min gap {gap_o.min().item():.2e}")
    assert sim_err < 2e-6
    assert (gap_o[0][dis] < 1e-5).all()                     # a different exemplar position only on near-ties
    assert yerr[~dis].max().item() < 1e-4
    assert torch.equal(res["y_up"][:, :, ::4, ::4], res["y_small"]) and torch.equal(res["y_up"][:, :, 3::4, 3::4], res["y_small"])
    # ---- configs[4] at this size: the bf16 candidate filter + fp32 re-scoring against the SAME oracle tensors (theta / phi
    # projected by the HIP 1x1 convolution from the oracle's trunk features: 3e-7 from the oracle's, asserted above)
    res16 = ops.corr_fwd_bf16(warp.project("theta", A_feat_o.cuda(), bf16=True), warp.project("phi", B_feat_o.cuda(), bf16=True),
                              ops.avgpool4x4(IB.cuda()).view(1, 3, -1), T, H // 4, W // 4, want_small=True, want_argmax=True)
    dis16 = res16["argmax"][0].cpu().long() != amax_o[0]
    sim_err16 = (res16["sim_small"].cpu() - sim_o).abs().max().item()
    yerr16 = (res16["y_small"].cpu() - y_o).abs().view(3, -1).max(0)[0]
    report(f"config4 432x768 bf16 correlation vs oracle: argmax differs on {int(dis16.sum())}/20736 rows (max gap among them "
           f"{gap_o[0][dis16].max().item() if dis16.any() else 0:.2e}), sim_err={sim_err16:.2e}, warped colour max err on agreeing rows "
           f"{yerr16[~dis16].max().item():.2e}")
    assert sim_err16 < 2e-6
    assert (gap_o[0][dis16] < 1e-5).all()
    assert yerr16[~dis16].max().item() < 1e-4
    ab_stage = col(cin_o.cuda())
    d = (ab_stage.cpu() - ab_o).abs()
    report(f"config4 432x768 ColorVidNet (identical input): max={d.max():.2e} mean={d.mean():.2e} (|ab| max {ab_o.abs().max():.2f})")
    assert d.max().item() <= NORTH_STAR_TOL and d.max().item() <= 2.5e-4
    # ---- end to end from the Lab frame (one frame, I_last = prev), the HIP path's own intermediates
    cc = ClipColorizer(vgg, warp, col, temperature=T)
    cc.set_exemplar(IB.cuda())
    ab, nl = cc.frame(fr.cuda(), prev.cuda())
    flips = ((nl[0, :, ::4, ::4].cpu() - y_o[0]).abs().max(0)[0] > 1e-3)
    d = (ab.cpu() - ab_o).abs()
    report(f"config4 432x768 end to end: rows whose warped colour differs {int(flips.sum())}/20736 "
           f"(oracle gaps there {gap_o[0][flips.view(-1)][:4].tolist()}); ab max={d.max():.2e} mean={d.mean():.2e}")
    assert (gap_o[0][flips.view(-1)] < 1e-5).all()
    if not flips.any():
        assert d.max().item() <= NORTH_STAR_TOL, d.max().item()
        assert d.max().item() <= 2.5e-4, d.max().item()
    else:   # a near-tie (gap < 1e-5, asserted above) picked the other exemplar position: the 4x4 block it feeds
        # changes and the rest of the frame does not (r02 run with seed 1000: 1 row, gap 8e-7 -> max 3.25, mean 1.2e-3)
        assert torch.quantile(d.flatten()[::7], 0.5).item() <= 1e-4


@pytest.mark.parametrize("seed", [1001, 1002, 1004])
def test_near_tie_frames_216x384_bounded_behaviour(seed):
    """Frames REJECTED for the literal test (synth.WELL_SEPARATED_FRAME_SEEDS_216x384 leaves out the seeds with a query row
    whose top-1/top-2 affinity gap is below 2e-6: fp32 rounding, ~3e-7 per affinity, decides that row's hard arg-max and the
    reference flips it with its own thread count).  What must hold there, at test.py's temperature 1e-10:
      (i)   the similarity map agrees everywhere (1e-5: end to end, theta comes from the HIP path's own VGG19 / WarpNet
            features, ~1e-6 from the oracle's) and the arg-max agrees on every row whose oracle gap is >= 1e-5;
      (ii)  a row that picks another exemplar position picks one whose float64 affinity is within 1e-5 of the row maximum
            (an admissible tie-break), and the warped colours differ from the oracle's ONLY in the 4x4 blocks of those rows
            (bit-equal to the pooled exemplar colour of the chosen position elsewhere and there);
      (iii) the ab prediction is within the north-star 1e-3 of the reference arithmetic evaluated WITH THAT TIE-BREAK (the
            oracle's ColorVidNet on the oracle's warped colours, the flipped rows' blocks replaced by the colour of the position
            the HIP path chose) — and within 1e-3 of the plain oracle when no row flipped.
    A flipped block is NOT local in ab (InstanceNorm couples the frame: a 25-Lab-unit change of one block moves the median
    of |d ab| by 2e-3 in the oracle itself), which is why (iii) is stated against the tie-break-matched reference."""
    import torch.nn.functional as F
    from dvc_amd import ops, synth
    from dvc_amd.frame import VGG_OUT, ClipColorizer
    from oracle import dvc_oracle as O
    from utils.util import feature_normalize, gray2rgb_batch
    H, W, T = 216, 384, 1e-10
    h, w = H // 4, W // 4
    P = h * w
    _oracle_threads()
    sd = _state_dicts()
    IB = synth.synth_lab(synth.EXEMPLAR_SEED, H, W)
    fr = synth.synth_lab(seed, H, W)
    prev = synth.synth_lab(synth.FRAME_SEED0 - 1, H, W)
    with torch.no_grad():
        taps = {}
        ab_o, nl_o, _ = O.frame_colorization(fr, IB, prev, O.exemplar_features(IB, sd[0]), *sd, temperature=T, taps=taps)
    gap = (taps["top2"][0, :, 0] - taps["top2"][0, :, 1])
    vgg, warp, col = _fresh_nets(sd)
    cc = ClipColorizer(vgg, warp, col, temperature=T)
    cc.set_exemplar(IB.cuda())
    fA = vgg(gray2rgb_batch(fr.cuda()[:, 0:1]), VGG_OUT)
    nA = [feature_normalize(t) for t in fA[1:]]
    y_up, sim_up, tp = warp(cc.IB_lab, *nA, *nA, temperature=T, exemplar_cache=cc.ex_cache, return_taps=True)
    ab, nl = cc.frame(fr.cuda(), prev.cuda())
    assert torch.equal(nl, y_up)
    amax = tp["argmax"][0].cpu().long()
    flipped = amax != taps["argmax"][0]
    sim_err = (tp["sim_small"].cpu() - taps["sim_small"]).abs().max().item()
    # (ii) admissible tie-break: float64 affinity of the chosen position against the row maximum, on the oracle's theta / phi
    th64, ph64 = taps["theta"][0].double(), taps["phi"][0].double()
    rows = flipped.nonzero().flatten()
    deficit = torch.zeros(0, dtype=torch.float64)
    if rows.numel():
        frow = th64[:, rows].t() @ ph64                                              # [n_flipped, P]
        deficit = frow.max(-1)[0] - frow.gather(1, amax[rows].unsqueeze(1)).squeeze(1)
    blab = F.avg_pool2d(IB, 4).view(3, P)
    y_alt = taps["y_small"].clone().view(3, P)
    y_alt[:, rows] = blab[:, amax[rows]]
    y_alt_up = F.interpolate(y_alt.view(1, 3, h, w), scale_factor=4, mode="nearest")
    with torch.no_grad():
        sim_up_o = F.interpolate(taps["sim_small"], scale_factor=4, mode="nearest")
        cin_alt = torch.cat((fr[:, 0:1], y_alt_up[:, 1:3], sim_up_o, prev), dim=1)
        ab_alt = O.colorvidnet_forward(sd[2], cin_alt) if rows.numel() else ab_o
    d_plain = (ab.cpu() - ab_o).abs()
    d_alt = (ab.cpu() - ab_alt).abs()
    report(f"near-tie frame 216x384 seed {seed}: oracle rows with gap<1e-5: {int((gap < 1e-5).sum())} (min {gap.min():.2e}); HIP picks another "
           f"position on {int(flipped.sum())} rows (gaps {gap[flipped].tolist()}, fp64 deficit of the chosen key {deficit.tolist()}); "
           f"sim_err={sim_err:.2e}; ab vs plain oracle max={d_plain.max():.2e} median={d_plain.median():.2e}; "
           f"ab vs tie-break-matched oracle max={d_alt.max():.2e}")
    assert (gap < 2e-6).any(), "this seed was chosen for a near-tie row"
    assert sim_err < 1e-5
    assert (gap[flipped] < 1e-5).all()                                   # (i)
    assert (deficit < 1e-5).all()                                        # (ii) admissible tie-break
    # (ii) one-hot colours, flipped rows included (2e-5: the pooled exemplar colours are fp32 means of 16 values ~100)
    assert (nl.cpu() - y_alt_up).abs().max().item() <= 2e-5
    assert torch.equal(nl, F.interpolate(nl[:, :, ::4, ::4], scale_factor=4, mode="nearest"))
    assert d_alt.max().item() <= NORTH_STAR_TOL, d_alt.max().item()      # (iii)
    if not flipped.any():
        assert d_plain.max().item() <= NORTH_STAR_TOL


_SOFT_ORACLE = {}


def _soft_oracle_clip(H, W, T, clip, NF, sd):
    """(fp32 oracle, fp64 oracle) predictions of free-running clip `clip` (exemplar seed 2 + clip, frames 1000 + 100 clip + i);
    cached across the parametrisations of the test below (the fp64 run of a 216x384 frame costs ~9 s of host time)."""
    from dvc_amd import synth
    from oracle import dvc_oracle as O
    key = (H, W, T, clip, NF)
    if key not in _SOFT_ORACLE:
        IB = synth.synth_lab(synth.EXEMPLAR_SEED + clip, H, W)
        frames = [synth.synth_lab(synth.FRAME_SEED0 + 100 * clip + i, H, W) for i in range(NF)]
        sd64 = tuple(O.to_dtype(s_, torch.float64) for s_ in sd)
        with torch.no_grad():
            r32 = O.colorize_clip(frames, IB, *sd, temperature=T)
            r64 = O.colorize_clip([f.double() for f in frames], IB.double(), *sd64, temperature=T)
        _SOFT_ORACLE[key] = (r32, r64)
    return _SOFT_ORACLE[key]


@pytest.mark.parametrize("B", [1, 2])
@pytest.mark.parametrize("T", [0.01, 0.005])
@pytest.mark.parametrize("H,W", [(40, 64), (216, 384)])
def test_soft_temperature_free_running_clip(H, W, T, B):
    """The API's other regime, end to end (r04 review, item 4): `frame_colorization`'s default temperature 0.01
    (FrameColor.py:52) and WarpNet.forward's 0.005 (NonlocalNet.py:438), free-running frames, at 40x64 — where layer5_1's
    output is one row short and NonlocalNet.py:461-463 pads a replicated row top and bottom — and at 216x384, for one clip
    and for a batch of two independent clips (each with its own exemplar; train.py:402 calls the function with B = 16).

    At these temperatures d(colour)/d(affinity) = |B_lab| / T ~ 1e4: an fp32 affinity (3e-7 from its fp64 value on either
    side, tests/test_gpu_nets.py reports it) moves the warped colours by ~1e-3, so the reference's OWN fp32 run is 1e-3 ... 3e-3
    from the fp64 truth on ab at T = 0.005 — two fp32 implementations cannot agree with each other to the north-star 1e-3
    there, however exact.  What is asserted, per frame of every clip (SURVEY.md §7 hard part 1: "<= the CPU figure, and <= 1e-3
    wherever the oracle's own fp32 is"):
      * against the fp64 truth: max-abs <= 1e-3 where the CPU fp32 oracle is within 1e-3 of it, else <= 1.25x the CPU oracle's
        max-abs (one value's lottery); mean <= 1.1x the CPU oracle's mean;
      * against the fp32 oracle: within the two runs' distances from the truth (triangle inequality, a consistency check),
        and within 1e-3 wherever BOTH are within 5e-4 of the truth."""
    from dvc_amd import synth
    from dvc_amd.frame import ClipColorizer
    _oracle_threads()
    sd = _state_dicts()
    NF = 4 if H <= 64 else 2
    IBs = [synth.synth_lab(synth.EXEMPLAR_SEED + b, H, W) for b in range(B)]
    clips = [[synth.synth_lab(synth.FRAME_SEED0 + 100 * b + i, H, W) for i in range(NF)] for b in range(B)]
    cc = ClipColorizer(*_fresh_nets(sd), temperature=T)
    cc.set_exemplar(torch.cat(IBs).cuda())
    got = cc.clip([torch.cat([clips[b][i] for b in range(B)]).cuda() for i in range(NF)], lookahead=2)
    torch.cuda.synchronize()
    for b in range(B):
        r32, r64 = _soft_oracle_clip(H, W, T, b, NF, sd)
        for i in range(NF):
            g = got[i][b:b + 1].double().cpu()
            e_gpu, e_cpu, d = (g - r64[i]).abs(), (r32[i].double() - r64[i]).abs(), (g - r32[i].double()).abs()
            report(f"e2e soft temperature {H}x{W} T={T} B={B} clip{b} frame{i}: |ab| max={r64[i].abs().max():.2f} "
                   f"gpu-vs-fp64 max={e_gpu.max():.2e} mean={e_gpu.mean():.2e} | cpu32-vs-fp64 max={e_cpu.max():.2e} "
                   f"mean={e_cpu.mean():.2e} | gpu-vs-cpu32 max={d.max():.2e}")
            assert r64[i].abs().max().item() > 1.0
            if e_cpu.max().item() <= NORTH_STAR_TOL:
                assert e_gpu.max().item() <= NORTH_STAR_TOL, (b, i, e_gpu.max().item())
            else:
                assert e_gpu.max().item() <= 1.25 * e_cpu.max().item(), (b, i, e_gpu.max().item(), e_cpu.max().item())
            assert e_gpu.mean().item() <= 1.1 * e_cpu.mean().item() + 1e-6, (b, i, e_gpu.mean().item(), e_cpu.mean().item())
            assert d.max().item() <= e_gpu.max().item() + e_cpu.max().item() + 1e-7
            if max(e_gpu.max().item(), e_cpu.max().item()) <= 5e-4:
                assert d.max().item() <= NORTH_STAR_TOL
