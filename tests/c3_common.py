"""Shared by the BASELINE configs[2] / configs[4] parity tests (tests/test_gpu_configs.py, tests/test_gpu_parallel_nccl.py):
the ORACLE side of a chunk of a free-running clip, with the per-frame tie-break matching of
test_near_tie_frames_216x384_bounded_behaviour extended along the recurrence.

SURVEY.md §8(d) C3: 64 frames (seeds 1000..1063) at 216x384 in 8 contiguous chunks of 8, every chunk started from
I_last = 0 (/root/reference/test.py:76-80, loop :68-96).  8 of every 12 such seeds have a query row whose top-1/top-2
affinity gap is below fp32 resolution of a 256-term dot product (~4e-7 per affinity): there the reference's hard arg-max
(temperature 1e-10) flips with its own thread count, and a flipped 4x4 block is not local in `ab` (InstanceNorm couples the
frame).  The statement that CAN be asserted on every frame, and is here: the HIP path may pick another exemplar position
only on rows whose oracle gap is < 1e-5, and its `ab` is within the north-star 1e-3 of the reference arithmetic evaluated
WITH THAT TIE-BREAK — the oracle's ColorVidNet recurrence (its own previous prediction fed back, test.py:96) on the oracle's
warped colours with the flipped rows' blocks carrying the pooled colour of the position the HIP path chose.  Without a flip
anywhere in a chunk that recurrence IS oracle.colorize_clip on the chunk.

Test infrastructure only (imports the oracle)."""
import torch
import torch.nn.functional as F

from oracle import dvc_oracle as O


def oracle_exemplar(sd, IB):
    """Everything of the oracle that depends on the exemplar alone (what O.frame_colorization recomputes per frame with
    identical results): phi.  Same functions, same inputs -> the same bits as inside O.warpnet_forward."""
    with torch.no_grad():
        fB = O.exemplar_features(IB, sd[0])
        nB = [O.feature_normalize(t) for t in fB[1:]]
        return O.corr_project(sd[1], "phi", O.warp_features(sd[1], *nB))


def oracle_front(sd, IB, phi, frame, T, keep_theta=False):
    """Front end of one frame through the oracle (O.warp_color's op sequence): the 1/4-resolution warped colours and
    similarity map, the arg-max and the top-1/top-2 gap of every query row (keep_theta: also theta, for the admissibility
    check of matched_oracle_chunk)."""
    with torch.no_grad():
        fA = O.vgg19_forward(sd[0], O.gray2rgb_batch(frame[:, 0:1]), O.VGG_OUT, preprocess=True)
        nA = [O.feature_normalize(t) for t in fA[1:]]
        theta = O.corr_project(sd[1], "theta", O.warp_features(sd[1], *nA))
        y, sim, f = O.correlate(theta, phi, IB, T)
        top2 = torch.topk(f, 2, dim=-1)[0]
        out = dict(y_small=y, sim_small=sim, argmax=f.argmax(-1)[0], gap=(top2[0, :, 0] - top2[0, :, 1]))
        if keep_theta:
            out["theta"] = theta
        return out


def hip_front(vgg, warp, cc, frame_dev, T):
    """The HIP path's front end of one frame with its taps (deterministic: the same launches ClipColorizer.clip issues)."""
    from dvc_amd import ops
    from dvc_amd.frame import VGG_OUT
    fA = vgg(ops.gray2rgb(frame_dev[:, 0:1]), VGG_OUT)
    nA = ops.channel_l2norm_multi(fA[1:])
    y_up, sim_up, tp = warp(cc.IB_lab, *nA, *nA, temperature=T, exemplar_cache=cc.ex_cache, return_taps=True)
    return dict(argmax=tp["argmax"][0].cpu().long(), sim_small=tp["sim_small"].cpu(), y_small=tp["y_small"].cpu(), y_up=y_up)


def matched_oracle_chunk(sd, IB, frames, fronts, hip_fronts, phi=None):
    """The oracle's recurrence over one chunk (I_last = 0 at its first frame) with the HIP path's tie-breaks.
    A row is "flipped" when the HIP path's warped colour differs from the oracle's by more than 1e-4 — another exemplar position
    at a near-tie, or (flat image regions: exactly duplicated exemplar features) another split of the weight among tied
    positions; on those rows the HIP path's own colour is handed to the oracle's ColorVidNet.  With `phi` and fronts that kept
    theta the substitution is checked to be ADMISSIBLE: the HIP colour must lie inside the range of the pooled colours of the
    keys whose float64 affinity is within 1e-5 of the row maximum (`inadmissible` counts the rows where it does not).
    Returns (list of ab, list of per-frame dicts: flipped rows, their oracle gaps, similarity error, ...)."""
    blab = F.avg_pool2d(IB, 4).view(3, -1)
    h, w = IB.shape[2] // 4, IB.shape[3] // 4
    outs, stats = [], []
    last = torch.zeros_like(frames[0])
    with torch.no_grad():
        for fr, fo, fh in zip(frames, fronts, hip_fronts):
            y = fo["y_small"].clone().view(3, -1)
            yh = fh["y_small"].view(3, -1)
            flipped = (yh - y).abs().max(0)[0] > 1e-4
            rows = flipped.nonzero().flatten()
            inadmissible = 0
            if rows.numel() and phi is not None and "theta" in fo:
                f64 = fo["theta"][0].double()[:, rows].t() @ phi[0].double()                   # [n_flipped, P]
                near = f64 >= f64.max(-1, keepdim=True)[0] - 1e-5
                for c in range(3):
                    lo = torch.where(near, blab[c].double()[None], torch.tensor(float("inf"), dtype=torch.float64)).min(-1)[0]
                    hi = torch.where(near, blab[c].double()[None], torch.tensor(float("-inf"), dtype=torch.float64)).max(-1)[0]
                    bad = (yh[c, rows].double() < lo - 1e-4) | (yh[c, rows].double() > hi + 1e-4)
                    inadmissible = max(inadmissible, int(bad.sum()))
            y[:, rows] = yh[:, rows]
            y_up = F.interpolate(y.view(1, 3, h, w), scale_factor=4, mode="nearest")
            sim_up = F.interpolate(fo["sim_small"], scale_factor=4, mode="nearest")
            ab = O.colorvidnet_forward(sd[2], torch.cat((fr[:, 0:1], y_up[:, 1:3], sim_up, last), dim=1))
            last = torch.cat((fr[:, 0:1], ab), dim=1)
            outs.append(ab)
            agree = ~flipped
            stats.append(dict(flipped=int(flipped.sum()), gaps=fo["gap"][flipped].tolist(), min_gap=fo["gap"].min().item(),
                              near_ties=int((fo["gap"] < 1e-5).sum()), exact_ties=int((fo["gap"] == 0).sum()),
                              argmax_differs=int((fh["argmax"] != fo["argmax"]).sum()), inadmissible=inadmissible,
                              sim_err=(fh["sim_small"] - fo["sim_small"]).abs().max().item(),
                              # warped colours on the agreeing rows (one-hot rows: pooled means of 16 values ~100)
                              y_err=(yh - fo["y_small"].view(3, -1))[:, agree].abs().max().item() if agree.any() else 0.0))
    return outs, stats
